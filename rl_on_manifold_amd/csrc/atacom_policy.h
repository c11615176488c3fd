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
// Two implementations:
//   * float + quad mapping (production): the three layers run on the matrix cores -- mlp_forward_mfma at the bottom of
//     this file, one wavefront = 16 environments = one GEMM column block;
//   * double (parity build) and the lane mapping: VALU form, mlp_forward.  Weights are staged once per workgroup into
//     LDS (rows padded so that the four lanes of a quad hit different banks); with 4 lanes per env the hidden units are
//     interleaved over the quad (unit j -> lane j % 4), the hidden vector is re-assembled with DPP broadcasts, the
//     output layer is a partial dot product + quad sum; weights leave LDS as ds_read_b128 and meet the (paired)
//     activations as v_pk_fma_f32.
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
    } else if constexpr (LANES == 2) {
#pragma unroll
        for (int m = 0; m < H / 2; ++m) full[m] = V2{qbcast<0, 2>(h[m]), qbcast<1, 2>(h[m])};
    } else if constexpr (LANES == 8) {
        // unit 8 m + l lives in lane l: pair p = 4 m + i holds units (8 m + 2 i, 8 m + 2 i + 1)
        static_for<0, H / 8>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            full[4 * m + 0] = V2{qbcast<0, 8>(h[m]), qbcast<1, 8>(h[m])};
            full[4 * m + 1] = V2{qbcast<2, 8>(h[m]), qbcast<3, 8>(h[m])};
            full[4 * m + 2] = V2{qbcast<4, 8>(h[m]), qbcast<5, 8>(h[m])};
            full[4 * m + 3] = V2{qbcast<6, 8>(h[m]), qbcast<7, 8>(h[m])};
        });
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
    V2 f1[H / 2];
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
        mean[o] = lds[L::B3 + o] + (LANES > 1 ? qsum<(LANES > 1 ? LANES : 4)>(p) : p);
    }
}

// ------------------------------------------------------------------------------------------------------------
// MFMA form of the same network (float, quad mapping): one wavefront = 16 environments, and the three layers are
// GEMMs of exactly the shape of v_mfma_f32_16x16x4_f32.  Computed TRANSPOSED,  H^T = W . X^T  (M = units,
// N = the wave's 16 environments, K = inputs), because then the accumulator layout of one layer (lane l, register v
// = unit 4 (l / 16) + v of environment l % 16) IS the B-operand layout of the next -- with the K index permuted,
// which only changes where the A operand (a weight) is fetched from: W[row l % 16][16 t + 4 (l / 16) + v], four
// consecutive floats per lane = one ds_read_b128 per four MFMAs.  So no re-layout between layers; the only shuffles
// are the observation going in (quad-replicated -> [env][element], via 2 KB of per-wave LDS) and the NK action
// means coming out.  100 MFMAs + ~50 LDS accesses per network and step, against ~0.75 k packed FMAs + 0.37 k LDS
// reads in the VALU form above.  The fp32 MFMA rate equals the packed fp32 VALU rate on this chip; what is gained
// is issue slots (one instruction per 1024 MACs) and LDS traffic, which is what a lone wave per SIMD runs out of.
typedef float mfma_v4f __attribute__((ext_vector_type(4)));

template <int D, int H, int NK>
struct MlpLdsM {
    static_assert(H == 64 && NK <= 8 && D <= 32, "MFMA policy path: 64 hidden units, <= 8 actions, <= 32 observations");
    static constexpr int CH = (D + 3) / 4;       // observation elements per lane group = K-steps of layer 1
    static constexpr int S1 = 32;                // W1 row: [lane group g][8] holding elements CH g + s
    static constexpr int S2 = H + 4;             // W2 / W3 rows (k contiguous), padded against bank conflicts
    static constexpr int W1 = 0, W2 = W1 + H * S1, W3 = W2 + H * S2, B1 = W3 + 16 * S2, B2 = B1 + H, B3 = B2 + H,
                         SHIFT = B3 + 16, SCALE = SHIFT + 32, STD = SCALE + 32, NET = STD + 16;
    // per-wave staging (floats) for NB blocks of 16 environments: observations [env][32], action means [env][8]
    static constexpr int XT = 0;
    static constexpr int act_offset(int nb) { return 16 * nb * 32; }
    static constexpr int wave_stage(int nb) { return 16 * nb * (32 + 8); }
};

template <int D, int H, int NK>
__device__ __forceinline__ void mlp_stage_weights_mfma(float* lds, const float* W1, const float* b1, const float* W2,
                                                       const float* b2, const float* W3, const float* b3, int tid,
                                                       int nthreads) {
    using L = MlpLdsM<D, H, NK>;
    for (int i = tid; i < H * D; i += nthreads) {
        const int u = i / D, e = i % D;
        lds[L::W1 + u * L::S1 + (e / L::CH) * 8 + (e % L::CH)] = W1[i];
    }
    for (int i = tid; i < H * H; i += nthreads) lds[L::W2 + (i / H) * L::S2 + (i % H)] = W2[i];
    for (int i = tid; i < NK * H; i += nthreads) lds[L::W3 + (i / H) * L::S2 + (i % H)] = W3[i];
    for (int i = tid; i < H; i += nthreads) { lds[L::B1 + i] = b1[i]; lds[L::B2 + i] = b2[i]; }
    for (int i = tid; i < NK; i += nthreads) lds[L::B3 + i] = b3[i];
}

// LDS: [net 0][net 1 (sigma network, optional)][per-wave staging x nwaves]
template <int D, int H, int NK>
__device__ __forceinline__ void mlp_stage_mfma(const MlpArgs<float>& net, float* lds, int tid, int nthreads,
                                               int stage_floats) {
    using L = MlpLdsM<D, H, NK>;
    const int total = 2 * L::NET + stage_floats;
    for (int i = tid; i < total; i += nthreads) lds[i] = 0.0f;
    __syncthreads();
    mlp_stage_weights_mfma<D, H, NK>(lds, net.W1, net.b1, net.W2, net.b2, net.W3, net.b3, tid, nthreads);
    if (net.sW1)
        mlp_stage_weights_mfma<D, H, NK>(lds + L::NET, net.sW1, net.sb1, net.sW2, net.sb2, net.sW3, net.sb3, tid, nthreads);
    for (int i = tid; i < NK; i += nthreads) lds[L::STD + i] = net.std ? net.std[i] : 0.0f;
    for (int i = tid; i < 32; i += nthreads) {
        // [g][8] like W1; padding: shift 0, scale 1 (the padded observation elements are 0)
        const int g = i / 8, s = i % 8, e = L::CH * g + s;
        const bool real = s < L::CH && e < D;
        lds[L::SHIFT + i] = (real && net.obs_shift) ? net.obs_shift[e] : 0.0f;
        lds[L::SCALE + i] = (real && net.obs_scale) ? net.obs_scale[e] : 1.0f;
    }
    __syncthreads();
}

// wave-level ordering of LDS traffic between lanes of ONE wavefront: its DS instructions execute in program order,
// so all that is needed is that the compiler keeps that order
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// activation of one lane's 16 accumulator entries; ONE wave-uniform branch per layer (per element it costs a branch
// pair each)
__device__ __forceinline__ void mlp_act16(mfma_v4f (&h)[4], int activation) {
    if (activation == 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) h[t][v] = num<float>::max(h[t][v], 0.0f);
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) h[t][v] = num<float>::tanh(h[t][v]);
    }
}

// The wavefront's environments are processed as NB blocks of 16 (quad mapping: NB = 1, lane mapping: NB = 4); the A
// operands (weights) are fetched once per layer tile and reused for every block.
// xin[blk][s] = normalised observation element CH g + s of environment 16 blk + lane % 16   (g = lane / 16): the B
// operand of layer 1, produced by mlp_obs_to_operand once per step and shared by both networks.
// `erow` = this lane's own environment within the wavefront (quad mapping: lane / 4, lane mapping: lane).
template <int D, int H, int NK, int NB>
__device__ __forceinline__ void mlp_forward_mfma(const float* __restrict__ net, float* __restrict__ stage,
                                                 const float (&xin)[NB][MlpLdsM<D, H, NK>::CH], int activation,
                                                 int lane, int erow, float (&mean)[NK]) {
    using L = MlpLdsM<D, H, NK>;
    using V4 = mfma_v4f;
    const int n16 = lane & 15, g = lane >> 4;
    // ---- layer 1: h1[blk][t][v] = unit 16 t + 4 g + v of environment 16 blk + n16
    V4 h1[NB][4];
    float w1[4][8];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const V4 bias = *reinterpret_cast<const V4*>(net + L::B1 + 16 * t + 4 * g);
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) h1[blk][t] = bias;
        const float* row = net + L::W1 + (16 * t + n16) * L::S1 + 8 * g;
        const V4 lo = *reinterpret_cast<const V4*>(row);
        const V4 hi = *reinterpret_cast<const V4*>(row + 4);
#pragma unroll
        for (int s = 0; s < 4; ++s) { w1[t][s] = lo[s]; w1[t][4 + s] = hi[s]; }
    }
#pragma unroll
    for (int s = 0; s < L::CH; ++s)
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                h1[blk][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[t][s], xin[blk][s], h1[blk][t], 0, 0, 0);
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) mlp_act16(h1[blk], activation);
    // ---- layer 2: K-step (t, v) carries unit 16 t + 4 g + v, i.e. register v of h1[.][t]
    V4 h2[NB][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const V4 bias = *reinterpret_cast<const V4*>(net + L::B2 + 16 * u + 4 * g);
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) h2[blk][u] = bias;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        V4 w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = *reinterpret_cast<const V4*>(net + L::W2 + (16 * u + n16) * L::S2 + 16 * t + 4 * g);
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int blk = 0; blk < NB; ++blk)
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    h2[blk][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u][v], h1[blk][t][v], h2[blk][u], 0, 0, 0);
    }
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) mlp_act16(h2[blk], activation);
    // ---- output layer: rows >= NK of W3 are zero; with one block, two accumulators (even / odd k-tiles) halve the
    // dependent chain
    V4 oa[NB], ob[NB];
    {
        const V4 bias = *reinterpret_cast<const V4*>(net + L::B3 + 4 * g);
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) { oa[blk] = bias; ob[blk] = V4{0.0f, 0.0f, 0.0f, 0.0f}; }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const V4 w = *reinterpret_cast<const V4*>(net + L::W3 + n16 * L::S2 + 16 * t + 4 * g);
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                if ((t & 1) == 0 || NB > 1) oa[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[v], h2[blk][t][v], oa[blk], 0, 0, 0);
                else ob[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[v], h2[blk][t][v], ob[blk], 0, 0, 0);
            }
    }
    // ---- back to the owners: lane (n16, g) holds outputs 4 g .. 4 g + 3 of environment 16 blk + n16 -> [env][8]
    float* act = stage + L::act_offset(NB);
    wave_lds_fence();
    if (g < 2) {
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) *reinterpret_cast<V4*>(act + (16 * blk + n16) * 8 + 4 * g) = oa[blk] + ob[blk];
    }
    wave_lds_fence();
    const float* mine = act + erow * 8;
    const V4 m0 = *reinterpret_cast<const V4*>(mine), m1 = *reinterpret_cast<const V4*>(mine + 4);
#pragma unroll
    for (int o = 0; o < NK; ++o) mean[o] = o < 4 ? m0[o < 4 ? o : 0] : m1[o >= 4 ? o - 4 : 0];
    wave_lds_fence();
}

// observation of this lane's environment -> the wave's staging area -> B operands of layer 1 (normalised).  In the
// quad mapping the four lanes of a quad store identical values to the same row.
template <int D, int H, int NK, int NB>
__device__ __forceinline__ void mlp_obs_to_operand(const float* __restrict__ net0, float* __restrict__ stage,
                                                   const float (&obs)[D], int lane, int erow,
                                                   float (&xin)[NB][MlpLdsM<D, H, NK>::CH]) {
    using L = MlpLdsM<D, H, NK>;
    using V4 = mfma_v4f;
    const int n16 = lane & 15, g = lane >> 4;
    float* row = stage + L::XT + erow * 32;
#pragma unroll
    for (int q = 0; q < L::CH; ++q) {
        V4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (4 * q + i < D) ? obs[4 * q + i < D ? 4 * q + i : 0] : 0.0f;
        *reinterpret_cast<V4*>(row + 4 * q) = v;
    }
    wave_lds_fence();
#pragma unroll
    for (int s = 0; s < L::CH; ++s) {
        const float sh = net0[L::SHIFT + 8 * g + s], sc = net0[L::SCALE + 8 * g + s];
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
            xin[blk][s] = (stage[L::XT + (16 * blk + n16) * 32 + L::CH * g + s] - sh) * sc;
    }
    wave_lds_fence();
}

}  // namespace atacom
