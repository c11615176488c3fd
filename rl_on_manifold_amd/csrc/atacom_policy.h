// Row N2: the policy network evaluated INSIDE the rollout kernel, so a whole on-policy collection phase
// (observation -> MLP -> Gaussian exploration noise -> ATACOM env step, T times) is one launch with the
// per-env state in registers -- no per-step Python / launch round trip.
//
// Network = the actor architecture every reference training script builds (examples/network.py:8-36 PPONetwork,
// :39-68 TRPONetwork, :266-293 SACActorNetwork): Linear(n_in, 64) -> ReLU -> Linear(64, 64) -> ReLU ->
// Linear(64, n_out); observations first go through an affine normalisation (MinMaxPreprocessor,
// examples/iiwa_air_hockey_exp.py:32-34), the action is  mu(obs) + std * eps  (GaussianTorchPolicy, std_0 = 0.5,
// examples/iiwa_air_hockey_exp.py:138-146) with eps supplied by the caller, so the kernel draws no random numbers.
//
// Mapping: weights are staged once per workgroup into LDS (26 KB in f32, rows padded so that the four lanes of
// a quad hit different banks).  With 4 lanes per env the hidden units are interleaved over the quad (unit j ->
// lane j % 4): layer 1 and 2 cost 1/4 each per lane, the hidden vector is re-assembled with DPP broadcasts, the
// output layer is a partial dot product + quad sum.  This is ~1.8 k instructions per env step (+8 %).
// MFMA was considered and rejected here: per wave the GEMM is only 16 x 64 x 64 and the A operand would have to
// be re-laid-out from the quad-replicated observation through LDS on every step; the VALU form is <10 % of the
// step and needs no layout change.
#pragma once
#include "atacom_quad.h"

namespace atacom {

template <typename T>
struct MlpArgs {
    const T *W1, *b1, *W2, *b2, *W3, *b3;   // torch.nn.Linear layout: W[out][in]
    const T *obs_shift, *obs_scale;          // x = (obs - shift) * scale      (nullable: identity)
    const T *std;                            // exploration std per action dim (nullable: 0)
    // optional second network of the same shape producing log(sigma) per action dim (SAC: actor_sigma_params,
    // examples/iiwa_air_hockey_exp.py:310-314); when present it replaces `std`
    const T *sW1, *sb1, *sW2, *sb2, *sW3, *sb3;
    T log_std_min, log_std_max;              // clamp of the sigma network's output (MushroomRL SACPolicy: -20, 2)
    int n_in, n_out, activation;             // activation: 0 ReLU, 1 tanh
    int squash;                              // 1: action = tanh(mean + sigma * eps)   (SAC's squashed Gaussian)
};

template <int D, int H, int NK>
struct MlpLds {
    static constexpr int S1 = ((D + 3) / 4) * 4 + 4;   // padded row strides (floats)
    static constexpr int S2 = H + 4;
    static constexpr int S3 = 8;                        // W3 stored transposed: [unit][out], NK <= 8
    static constexpr int W1 = 0, W2 = W1 + H * S1, W3T = W2 + H * S2, B1 = W3T + H * S3, B2 = B1 + H,
                         B3 = B2 + H, SHIFT = B3 + 8, SCALE = SHIFT + ((D + 3) / 4) * 4,
                         STD = SCALE + ((D + 3) / 4) * 4, TOTAL = STD + 8;
};

// cooperative staging by the whole workgroup (call before any early return)
template <typename T, int D, int H, int NK>
__device__ __forceinline__ void mlp_stage_weights(T* lds, const T* W1, const T* b1, const T* W2, const T* b2,
                                                  const T* W3, const T* b3, int tid, int nthreads) {
    using L = MlpLds<D, H, NK>;
    for (int i = tid; i < H * D; i += nthreads) lds[L::W1 + (i / D) * L::S1 + (i % D)] = W1[i];
    for (int i = tid; i < H * H; i += nthreads) lds[L::W2 + (i / H) * L::S2 + (i % H)] = W2[i];
    for (int i = tid; i < NK * H; i += nthreads) lds[L::W3T + (i % H) * L::S3 + (i / H)] = W3[i];
    for (int i = tid; i < H; i += nthreads) { lds[L::B1 + i] = b1[i]; lds[L::B2 + i] = b2[i]; }
    for (int i = tid; i < NK; i += nthreads) lds[L::B3 + i] = b3[i];
}

// LDS holds one MlpLds block for the mean network and, if present, a second one for the sigma network
template <typename T, int D, int H, int NK>
__device__ __forceinline__ void mlp_stage(const MlpArgs<T>& net, T* lds, int tid, int nthreads) {
    using L = MlpLds<D, H, NK>;
    const int total = net.sW1 ? 2 * L::TOTAL : L::TOTAL;
    for (int i = tid; i < total; i += nthreads) lds[i] = T(0);
    __syncthreads();
    mlp_stage_weights<T, D, H, NK>(lds, net.W1, net.b1, net.W2, net.b2, net.W3, net.b3, tid, nthreads);
    if (net.sW1)
        mlp_stage_weights<T, D, H, NK>(lds + L::TOTAL, net.sW1, net.sb1, net.sW2, net.sb2, net.sW3, net.sb3, tid, nthreads);
    for (int i = tid; i < NK; i += nthreads) lds[L::STD + i] = net.std ? net.std[i] : T(0);
    for (int i = tid; i < D; i += nthreads) {
        lds[L::SHIFT + i] = net.obs_shift ? net.obs_shift[i] : T(0);
        lds[L::SCALE + i] = net.obs_scale ? net.obs_scale[i] : T(1);
    }
    __syncthreads();
}

template <typename T>
__device__ __forceinline__ T mlp_act(T v, int activation) {
    return activation == 0 ? num<T>::max(v, T(0)) : tanh(v);
}

// mean action of the policy for one env: x = normalised observation (replicated in the quad)
template <typename T, int D, int H, int NK, int LANES>
__device__ __forceinline__ void mlp_forward(const T* __restrict__ lds, const T* __restrict__ lds_norm,
                                            const T (&obs)[D], int activation, int lq, T (&mean)[NK]) {
    using L = MlpLds<D, H, NK>;
    constexpr int U = H / LANES;
    T x[D];
#pragma unroll
    for (int i = 0; i < D; ++i) x[i] = (obs[i] - lds_norm[L::SHIFT + i]) * lds_norm[L::SCALE + i];
    T h1[U];
#pragma unroll
    for (int m = 0; m < U; ++m) {
        const int j = LANES * m + lq;
        const T* row = lds + L::W1 + j * L::S1;
        T acc = lds[L::B1 + j];
#pragma unroll
        for (int i = 0; i < D; ++i) acc = num<T>::fma(row[i], x[i], acc);
        h1[m] = mlp_act(acc, activation);
        __builtin_amdgcn_sched_barrier(0);       // keep at most one unit's weights in flight (register pressure)
    }
    T f1[H];
#pragma unroll
    for (int m = 0; m < U; ++m) {
        if (LANES == 4) {
            f1[4 * m + 0] = qbcast<0>(h1[m]); f1[4 * m + 1] = qbcast<1>(h1[m]);
            f1[4 * m + 2] = qbcast<2>(h1[m]); f1[4 * m + 3] = qbcast<3>(h1[m]);
        } else {
            f1[m] = h1[m];
        }
    }
    T h2[U];
#pragma unroll
    for (int m = 0; m < U; ++m) {
        const int j = LANES * m + lq;
        const T* row = lds + L::W2 + j * L::S2;
        T acc = lds[L::B2 + j];
#pragma unroll
        for (int i = 0; i < H; ++i) {
            acc = num<T>::fma(row[i], f1[i], acc);
            if ((i & 15) == 15) __builtin_amdgcn_sched_barrier(0);
        }
        h2[m] = mlp_act(acc, activation);
    }
#pragma unroll
    for (int o = 0; o < NK; ++o) {
        T part = T(0);
#pragma unroll
        for (int m = 0; m < U; ++m) part = num<T>::fma(lds[L::W3T + (LANES * m + lq) * L::S3 + o], h2[m], part);
        mean[o] = lds[L::B3 + o] + (LANES == 4 ? qsum(part) : part);
    }
}

}  // namespace atacom
