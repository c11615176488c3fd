// The lane-group kernels (4 and 8 lanes per environment) of the iiwa environment, float32, reference chart, kinematic mode --
// the headline kernels -- as a list, so that atacom_iiwa.hip can DECLARE them (extern template) and atacom_iiwa_group.hip
// DEFINE them in a translation unit of its own.  Why two units (round 5): the compiler's
// iterative-ilp scheduler makes these kernels 1.0 - 1.4 % faster (23.98 -> 23.74 us per step at 8192 environments,
// profiles/r05_ab_sched.log; no scratch in any of the twelve) and the one-environment-per-lane rollout kernel of the same
// source 29 % slower (116 instead of 12 bytes of scratch, profiles/r04_ab_sched_iterative_ilp.log) -- the flag is per unit.
#pragma once
#include "atacom_ops_impl.h"
#define ATACOM_IIWA_GROUP_KERNELS_OF(X, L, H)                                                                                  \
    X __global__ void k_step<float, Iiwa, L, H, false, 0, false>(const Params<float>, float*, int*, const float*, float*,     \
                                                                 float*, uint8_t*, uint8_t*, const uint8_t*);                  \
    X __global__ void k_rollout<float, Iiwa, L, H, false, 0, false>(const Params<float>, int, float*, int*, const float*,     \
                                                                    float*, float*, float*, uint8_t*, uint8_t*, float*, int);  \
    X __global__ void k_rollout_mlp<float, Iiwa, L, H, 64, false, 0, false>(                                                   \
        const Params<float>, const MlpArgs<float>, int, float*, int*, const float*, float*, float*, float*, float*, uint8_t*,  \
        uint8_t*, float*, int);
#define ATACOM_IIWA_GROUP_KERNELS(X)                                                                                           \
    ATACOM_IIWA_GROUP_KERNELS_OF(X, 8, true) ATACOM_IIWA_GROUP_KERNELS_OF(X, 8, false)                                         \
    ATACOM_IIWA_GROUP_KERNELS_OF(X, 4, true) ATACOM_IIWA_GROUP_KERNELS_OF(X, 4, false)
