// Type-erased launch table: one per (environment, scalar type); defined in the per-environment device
// translation units, consumed by the C-ABI host code (atacom_capi.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/atacom_hip.h"

namespace atacom {

struct EnvOps {
    int n_planes, n_iplanes;     // per-environment allocation: values in the float buffer (hot + cold groups of four), ints
    int state_dim, init_dim, obs_dim, nq, nf, ng, nk;
    size_t elem;
    // lanes = 1, 2, 4 or 8 (lanes per environment)
    // mask (nullable): environments with a zero byte sit the step out
    void (*step)(const atacom_config&, int lanes, void* f, int* ip, const void* act, void* obs, void* rew,
                 uint8_t* ab, uint8_t* last, const uint8_t* mask, hipStream_t s);
    void (*rollout)(const atacom_config&, int lanes, int n_steps, void* f, int* ip, const void* acts, void* obs,
                    void* nobs, void* rew, uint8_t* ab, uint8_t* last, void* rec, int rec_ld, hipStream_t s);
    // returns 0, or -3 if the (env, hidden size) combination is not compiled in
    int (*rollout_mlp)(const atacom_config&, int lanes, int n_steps, const atacom_mlp& net, void* f, int* ip,
                       const void* noise, void* obs, void* nobs, void* acts, void* rew, uint8_t* ab, uint8_t* last,
                       void* rec, int rec_ld, hipStream_t s);
    void (*reset)(const atacom_config&, void* f, int* ip, const uint8_t* mask, const void* init, void* obs,
                  hipStream_t s);
    void (*fill_init)(const atacom_config&, void* f, int* ip, const void* row, hipStream_t s);
    void (*clear_stats)(const atacom_config&, void* f, int* ip, hipStream_t s);
    void (*stats)(const atacom_config&, const void* f, const int* ip, double* partial, int nblocks, hipStream_t s);
    void (*get_state)(const atacom_config&, const void* f, const int* ip, void* out, hipStream_t s);
    void (*set_state)(const atacom_config&, void* f, int* ip, const void* in, hipStream_t s);
    void (*nullspace)(int lanes, int n, const void* Jc, const void* rhs, double tol, void* x, void* nullb, void* rref,
                      hipStream_t s);
    void (*terms)(const atacom_config&, int n, const void* q, const void* dq, void* fun, void* J, void* b,
                  hipStream_t s);
    // obs_delay: the low-pass state [batch, 3 + nq] (planar / iiwa; no-op otherwise); set != 0 writes it
    void (*filter_io)(const atacom_config&, void* f, void* buf, int set, hipStream_t s);
    // the mapping a request for `lanes` really runs on (kind: 0 step, 1 T-step, 2 policy kernel; atacom_ops_impl.h: has_mapping)
    int (*lanes_run)(int kind, int lanes);
};

// The three stepping entry points of a kernel variant other than the default one (atacom_ops_impl.h: Variant)
struct VariantOps {
    void (*step)(const atacom_config&, int lanes, void* f, int* ip, const void* act, void* obs, void* rew,
                 uint8_t* ab, uint8_t* last, const uint8_t* mask, hipStream_t s);
    void (*rollout)(const atacom_config&, int lanes, int n_steps, void* f, int* ip, const void* acts, void* obs,
                    void* nobs, void* rew, uint8_t* ab, uint8_t* last, void* rec, int rec_ld, hipStream_t s);
    int (*rollout_mlp)(const atacom_config&, int lanes, int n_steps, const atacom_mlp& net, void* f, int* ip,
                       const void* noise, void* obs, void* nobs, void* acts, void* rew, uint8_t* ab, uint8_t* last,
                       void* rec, int rec_ld, hipStream_t s);
    // the canonical chart as a primitive: A [n, c, q], s [n, g], y [n, c], alpha [n, k] -> mu [n, q + g]
    void (*chart_mu)(int n, const void* A, const void* sl, const void* y, const void* alpha, double tol, void* mu,
                     hipStream_t s);
    int (*lanes_run)(int kind, int lanes);
};
// canonical-chart kernels (cfg.chart_mode = 1) of circle / planar / iiwa: atacom_chart.hip, atacom_chart_iiwa.hip
const VariantOps* ops_chart(int env_id, int dtype);
// the stepping kernels with the domain-randomisation options (cfg.obs_noise / obs_delay / env_noise) compiled in: planar and
// iiwa, kinematic mode, either chart (atacom_noise_planar.hip, atacom_noise_iiwa.hip, atacom_noise_iiwa_f64.hip)
const VariantOps* ops_noise(int env_id, int dtype, int chart_mode);
// Row N4: the rigid-body kernels of the iiwa environment (cfg.dynamics_mode = 1) with either chart: atacom_iiwa_dyn.hip
const VariantOps* ops_iiwa_dyn_variant(int dtype, int chart_mode);

// servo-joint state access and the stand-alone dynamics primitives (atacom_iiwa_dyn.hip)
struct DynOps {
    void (*get_aux)(const atacom_config&, const void* f, void* out, hipStream_t s);
    void (*set_aux)(const atacom_config&, void* f, const void* in, hipStream_t s);
    void (*inverse_dynamics)(int n, const void* q, const void* dq, const void* ddq, void* tau, void* M, hipStream_t s);
    void (*forward_dynamics)(int n, const void* q, const void* dq, const void* tau6, const void* ddq_aux, int use_damping,
                             void* ddq6, hipStream_t s);
};
const DynOps* ops_iiwa_dyn(int dtype);

const EnvOps* ops_circle(int dtype);
const EnvOps* ops_circle_ec(int dtype);
const EnvOps* ops_circle_t(int dtype);
const EnvOps* ops_planar(int dtype);
const EnvOps* ops_iiwa(int dtype);

}  // namespace atacom
