// Device code of the iiwa environment's rigid-body mode (row N4) and its stand-alone dynamics primitives, float32 and
// float64 -- a translation unit of its own so that it compiles in parallel with the default kernels.
// Mappings: one environment per lane (double, and float beyond 16384 environments) or per DPP quad (float).
#include "atacom_ops_impl.h"
namespace atacom {
template <typename T>
struct Dyn {
    using E = Iiwa;
    template <int LANES, bool HOLD>
    static void launch_step(const atacom_config& c, void* f, int* ip, const void* act, void* obs, void* rew, uint8_t* ab,
                            uint8_t* last, hipStream_t s) {
        hipLaunchKernelGGL((k_step<T, E, LANES, HOLD, true>), dim3(nblk(c.batch * LANES, BLOCK<LANES>)), dim3(BLOCK<LANES>),
                           0, s, make_params<T>(c), (T*)f, ip, (const T*)act, (T*)obs, (T*)rew, ab, last);
    }
    static void step(const atacom_config& c, int lanes, void* f, int* ip, const void* act, void* obs, void* rew,
                     uint8_t* ab, uint8_t* last, hipStream_t s) {
        if (lanes >= 4) {
            if (c.hold_q) launch_step<4, true>(c, f, ip, act, obs, rew, ab, last, s);
            else launch_step<4, false>(c, f, ip, act, obs, rew, ab, last, s);
        } else {
            if (c.hold_q) launch_step<1, true>(c, f, ip, act, obs, rew, ab, last, s);
            else launch_step<1, false>(c, f, ip, act, obs, rew, ab, last, s);
        }
    }
    template <int LANES, bool HOLD>
    static void launch_rollout(const atacom_config& c, int n_steps, void* f, int* ip, const void* acts, void* obs,
                               void* nobs, void* rew, uint8_t* ab, uint8_t* last, void* rec, int rec_ld, hipStream_t s) {
        hipLaunchKernelGGL((k_rollout<T, E, LANES, HOLD, true>), dim3(nblk(c.batch * LANES, BLOCK<LANES>)),
                           dim3(BLOCK<LANES>), 0, s, make_params<T>(c), n_steps, (T*)f, ip, (const T*)acts, (T*)obs,
                           (T*)nobs, (T*)rew, ab, last, (T*)rec, rec_ld);
    }
    static void rollout(const atacom_config& c, int lanes, int n_steps, void* f, int* ip, const void* acts, void* obs,
                        void* nobs, void* rew, uint8_t* ab, uint8_t* last, void* rec, int rec_ld, hipStream_t s) {
        if (lanes >= 4) {
            if (c.hold_q) launch_rollout<4, true>(c, n_steps, f, ip, acts, obs, nobs, rew, ab, last, rec, rec_ld, s);
            else launch_rollout<4, false>(c, n_steps, f, ip, acts, obs, nobs, rew, ab, last, rec, rec_ld, s);
        } else {
            if (c.hold_q) launch_rollout<1, true>(c, n_steps, f, ip, acts, obs, nobs, rew, ab, last, rec, rec_ld, s);
            else launch_rollout<1, false>(c, n_steps, f, ip, acts, obs, nobs, rew, ab, last, rec, rec_ld, s);
        }
    }
    static void get_aux(const atacom_config& c, const void* f, void* out, hipStream_t s) {
        hipLaunchKernelGGL((k_get_aux<T, E>), dim3(nblk(c.batch, 256)), dim3(256), 0, s, c.batch, (const T*)f, (T*)out);
    }
    static void set_aux(const atacom_config& c, void* f, const void* in, hipStream_t s) {
        hipLaunchKernelGGL((k_set_aux<T, E>), dim3(nblk(c.batch, 256)), dim3(256), 0, s, c.batch, (T*)f, (const T*)in);
    }
    static void inverse_dynamics(int n, const void* q, const void* dq, const void* ddq, void* tau, void* M, hipStream_t s) {
        hipLaunchKernelGGL((k_inverse_dynamics<T>), dim3(nblk(n, WAVE)), dim3(WAVE), 0, s, n, (const T*)q, (const T*)dq,
                           (const T*)ddq, (T*)tau, (T*)M);
    }
    static void forward_dynamics(int n, const void* q, const void* dq, const void* tau6, const void* ddq_aux,
                                 int use_damping, void* ddq6, hipStream_t s) {
        hipLaunchKernelGGL((k_forward_dynamics<T>), dim3(nblk(n, WAVE)), dim3(WAVE), 0, s, n, (const T*)q, (const T*)dq,
                           (const T*)tau6, (const T*)ddq_aux, use_damping, (T*)ddq6);
    }
    static const DynOps* table() {
        static const DynOps ops = {&step, &rollout, &get_aux, &set_aux, &inverse_dynamics, &forward_dynamics};
        return &ops;
    }
};
const DynOps* ops_iiwa_dyn(int dtype) { return dtype == ATACOM_F64 ? Dyn<double>::table() : Dyn<float>::table(); }
}  // namespace atacom
