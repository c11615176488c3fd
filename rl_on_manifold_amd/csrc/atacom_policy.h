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
// output layer is a partial dot product + quad sum.  Weights leave LDS as ds_read_b128 and meet the (paired)
// activations as v_pk_fma_f32: ~0.75 k packed FMAs + 0.37 k LDS reads per network and env step.
// MFMA was considered and rejected here: per wave the GEMM is only 16 x 64 x 64 and the A operand would have to
// be re-laid-out from the quad-replicated observation through LDS on every step; the VALU form is ~10 % of the
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
    return activation == 0 ? num<T>::max(v, T(0)) : num<T>::tanh(v);
}

template <typename T> using vec4 = T __attribute__((ext_vector_type(4)));

// one dense layer for this lane's units: out[m] = act(b[j] + W[j][:] . in),  j = LANES * m + lq.
// The input is replicated in the quad and held as pairs; a weight row comes out of LDS four floats per ds_read_b128
// (rows are zero-padded to a multiple of 4) and meets the input as two packed FMAs -- 2 MACs per issue slot, the
// two halves of the accumulator double as the two summation chains.
template <typename T, int NIN4, int U, int LANES>
__device__ __forceinline__ void mlp_dense(const T* __restrict__ w, int stride, const T* __restrict__ bias,
                                          const vec2<T> (&in2)[2 * NIN4], int activation, int lq, T (&out)[U]) {
    using V2 = vec2<T>;
    using V4 = vec4<T>;
#pragma unroll
    for (int m = 0; m < U; ++m) {
        const int j = LANES * m + lq;
        const V4* row = reinterpret_cast<const V4*>(w + j * stride);
        V2 acc0 = V2{bias[j], T(0)}, acc1 = splat2(T(0));      // two independent packed chains = four partial sums
#pragma unroll
        for (int q = 0; q < NIN4; ++q) {
            const V4 r = row[q];
            acc0 = fma2(V2{r.x, r.y}, in2[2 * q], acc0);
            acc1 = fma2(V2{r.z, r.w}, in2[2 * q + 1], acc1);
        }
        const V2 acc = acc0 + acc1;
        out[m] = mlp_act(acc.x + acc.y, activation);
        // one unit's weight row in flight at a time: left alone the scheduler hoists every ds_read of the layer to
        // the top and spills the rows to scratch
        __builtin_amdgcn_sched_barrier(0);
    }
}

// this lane's hidden units -> the full hidden vector, replicated in the quad, as pairs
template <typename T, int H, int LANES>
__device__ __forceinline__ void mlp_gather(const T (&h)[H / LANES], vec2<T> (&full)[H / 2]) {
    using V2 = vec2<T>;
    if constexpr (LANES == 4) {
#pragma unroll
        for (int m = 0; m < H / 4; ++m) {
            full[2 * m] = V2{qbcast<0>(h[m]), qbcast<1>(h[m])};
            full[2 * m + 1] = V2{qbcast<2>(h[m]), qbcast<3>(h[m])};
        }
    } else {
#pragma unroll
        for (int m = 0; m < H / 2; ++m) full[m] = V2{h[2 * m], h[2 * m + 1]};
    }
}

// mean action of the policy for one env: x = normalised observation (replicated in the quad)
template <typename T, int D, int H, int NK, int LANES>
__device__ __forceinline__ void mlp_forward(const T* __restrict__ lds, const T* __restrict__ lds_norm,
                                            const T (&obs)[D], int activation, int lq, T (&mean)[NK]) {
    using L = MlpLds<D, H, NK>;
    using V2 = vec2<T>;
    using V4 = vec4<T>;
    static_assert(H % 4 == 0 && H % LANES == 0 && NK <= 8, "");
    constexpr int U = H / LANES, D4 = (D + 3) / 4;
    V2 x2[2 * D4];
#pragma unroll
    for (int i = 0; i < 2 * D4; ++i) {
        T h[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int c = 2 * i + t;
            h[t] = (c < D) ? (obs[c < D ? c : 0] - lds_norm[L::SHIFT + (c < D ? c : 0)]) * lds_norm[L::SCALE + (c < D ? c : 0)]
                           : T(0);
        }
        x2[i] = V2{h[0], h[1]};
    }
    T h1[U], h2[U];
    V2 f1[H / 2], f2[H / 2];
    mlp_dense<T, D4, U, LANES>(lds + L::W1, L::S1, lds + L::B1, x2, activation, lq, h1);
    mlp_gather<T, H, LANES>(h1, f1);
    mlp_dense<T, H / 4, U, LANES>(lds + L::W2, L::S2, lds + L::B2, f1, activation, lq, h2);
    // output layer: W3 is stored transposed, [unit][8]; this lane's units contribute partial sums over the quad
    V2 part[4] = {splat2(T(0)), splat2(T(0)), splat2(T(0)), splat2(T(0))};
#pragma unroll
    for (int m = 0; m < U; ++m) {
        const V4* row = reinterpret_cast<const V4*>(lds + L::W3T + (LANES * m + lq) * L::S3);
        const V4 r0 = row[0], r1 = row[1];
        const V2 hm = splat2(h2[m]);
        part[0] = fma2(V2{r0.x, r0.y}, hm, part[0]);
        if (NK > 2) part[1] = fma2(V2{r0.z, r0.w}, hm, part[1]);
        if (NK > 4) part[2] = fma2(V2{r1.x, r1.y}, hm, part[2]);
        if (NK > 6) part[3] = fma2(V2{r1.z, r1.w}, hm, part[3]);
        if ((m & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int o = 0; o < NK; ++o) {
        const T p = part[o / 2][o % 2];
        mean[o] = lds[L::B3 + o] + (LANES == 4 ? qsum(p) : p);
    }
    (void)f2;
}

}  // namespace atacom
