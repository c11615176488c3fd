// Device code of the iiwa environment's rigid-body mode (row N4), float32, the reference's chart -- plus the servo-joint
// state access and the stand-alone dynamics primitives in both precisions.  A translation unit of its own so that it
// compiles in parallel with the default kernels (float64: atacom_iiwa_dyn_f64.hip; canonical chart: atacom_iiwa_dyn_chart.hip).
// Mappings: one environment per lane or per DPP quad (Variant: the dynamics are computed redundantly by a group's lanes).
#include "atacom_ops_impl.h"
namespace atacom {
template <typename T>
struct Dyn {
    using E = Iiwa;
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
        static const DynOps ops = {&get_aux, &set_aux, &inverse_dynamics, &forward_dynamics};
        return &ops;
    }
};
const DynOps* ops_iiwa_dyn(int dtype) { return dtype == ATACOM_F64 ? Dyn<double>::table() : Dyn<float>::table(); }

const VariantOps* ops_iiwa_dyn_f64();
const VariantOps* ops_iiwa_dyn_chart(int dtype);
const VariantOps* ops_iiwa_dyn_variant(int dtype, int chart_mode) {
    if (chart_mode == 1) return ops_iiwa_dyn_chart(dtype);
    return dtype == ATACOM_F64 ? ops_iiwa_dyn_f64() : Variant<float, Iiwa, true, 0>::table();
}
}  // namespace atacom
