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
    void (*step)(const atacom_config&, int lanes, void* f, int* ip, const void* act, void* obs, void* rew,
                 uint8_t* ab, uint8_t* last, hipStream_t s);
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
};

// Row N4: the rigid-body kernels of the iiwa environment live in their own translation unit (atacom_iiwa_dyn.hip)
struct DynOps {
    void (*step)(const atacom_config&, int lanes, void* f, int* ip, const void* act, void* obs, void* rew, uint8_t* ab,
                 uint8_t* last, hipStream_t s);
    void (*rollout)(const atacom_config&, int lanes, int n_steps, void* f, int* ip, const void* acts, void* obs,
                    void* nobs, void* rew, uint8_t* ab, uint8_t* last, void* rec, int rec_ld, hipStream_t s);
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
