// Lane-group variant of the null-space solver: EIGHT, FOUR (a DPP quad) or TWO (a lane pair) LANES PER ENVIRONMENT.
// (Written and described for the quad; the group size is the template parameter LN, see qbcast / qsum below.)
//
// Why (measured in profiles/): a wave64 vector instruction occupies its SIMD for 4 clocks and a
// lone wave already saturates it, so the step time is (vector instructions per wave) x ~5 clk.  With one environment
// per lane the headline batch of 8192 environments is 128 wavefronts on 1024 SIMDs, each running the whole 23 k-
// instruction step (51 us).  Splitting every environment over the 4 lanes of a DPP quad puts 512 waves to work at
// ~0.5x the instructions per wave (26 us); beyond 16384 environments (1024 waves) the narrower mappings win again
// because their total instruction count is lower.
//
// Data distribution inside a quad (lq = lane & 3) -- "column 0 replicated", see split_slots below:
//   * matrices with N columns (J_c: M x N, null basis: N x K): column 0 is held by every lane, column c >= 1 lives in
//     lane (c - 1) % 4, "slot" (c - 1) / 4  (S = ceil((N - 1) / 4) slots per lane; slots past N hold zeros);
//   * vectors over the M rows (rhs y, the bidiagonal d / e, left reflectors u) are REPLICATED in the four
//     lanes and computed redundantly -- their values stay bitwise identical across the quad because every
//     cross-lane sum uses the same commutative butterfly;
//   * cross-lane traffic is DPP only (quad_perm; row_half_mirror for 8 lanes): a broadcast is one v_mov_dpp, a quad
//     sum two v_add_f32_dpp; no LDS, no ds_bpermute, no barriers;
//   * "my column of a replicated array" is picked by a one-hot FMA blend, never by a select chain on lq (the
//     optimiser turns those into a divergent 4-way switch).
// The arithmetic is the same Householder bidiagonalisation / rref chart as atacom_linalg.h (the one-lane-per-env
// implementation); only the summation order inside dot products differs.
#pragma once
#include "atacom_linalg.h"

namespace atacom {

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}

// Everything below is written for a GROUP of LN lanes per environment: LN = 4 (a DPP quad), LN = 2 (a lane pair, two
// environments per quad) or LN = 8 (two adjacent quads = half a DPP row).  The fewer lanes, the less redundant work in
// total but the more instructions per wave; the mapping of choice is the widest one whose waves still find a SIMD each
// (atacom_capi.cpp: pick_lanes).  LN = 8 crosses the quad boundary with row_half_mirror (lane i <-> 7 - i inside each
// group of 8): after the two quad_perm butterfly levels every lane of a quad holds the quad's sum, so mirroring pairs
// each quad with the other one.
constexpr int DPP_ROW_HALF_MIRROR = 0x141;
template <int CTRL, int BANK_MASK>
__device__ __forceinline__ float dpp_mov_keep(float old, float v) {      // lanes outside BANK_MASK keep `old`
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                                 CTRL, 0xF, BANK_MASK, false));
}
template <int CTRL, int BANK_MASK>
__device__ __forceinline__ int dpp_mov_keep(int old, int v) {
    return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xF, BANK_MASK, false);
}
template <int CTRL, int BANK_MASK>
__device__ __forceinline__ double dpp_mov_keep(double old, double v) {
    const long long b = __builtin_bit_cast(long long, v), o = __builtin_bit_cast(long long, old);
    const int lo = __builtin_amdgcn_update_dpp((int)(o & 0xffffffffll), (int)(b & 0xffffffffll), CTRL, 0xF, BANK_MASK, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(o >> 32), (int)(b >> 32), CTRL, 0xF, BANK_MASK, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
//
// value of lane O (0..LN-1) of the group, in all its lanes
template <int O, int LN = 4, typename V>
__device__ __forceinline__ V qbcast(V v) {
    static_assert(LN == 8 || LN == 4 || LN == 2, "");
    if constexpr (LN == 8) {
        // quad_perm [o,o,o,o] makes every quad uniform; the quad that does not own lane O then takes the other quad's
        // value through row_half_mirror, written only to its banks (bank = 4 lanes; a row of 16 lanes = 4 banks)
        const V t = dpp_mov<(O % 4) * 0x55>(v);
        return dpp_mov_keep<DPP_ROW_HALF_MIRROR, (O / 4 == 0) ? 0xA : 0x5>(t, t);
    } else {
        // quad_perm: LN = 4 -> [O,O,O,O];  LN = 2 -> [O,O,2+O,2+O]
        return dpp_mov<(LN == 4) ? O * 0x55 : (O * 0x05 + (2 + O) * 0x50)>(v);
    }
}
// sum over the group, identical bits in all its lanes
template <int LN = 4, typename T>
__device__ __forceinline__ T qsum(T v) {
    const T s1 = v + dpp_mov<0xB1>(v);     // quad_perm [1,0,3,2]
    if constexpr (LN == 2) return s1;
    else {
        const T s2 = s1 + dpp_mov<0x4E>(s1);    // quad_perm [2,3,0,1]
        if constexpr (LN == 4) return s2;
        else return s2 + dpp_mov<DPP_ROW_HALF_MIRROR>(s2);
    }
}
// element LN*slot + lq of a replicated compile-time-indexed array (0 past its end), as a one-hot blend over the
// group (exact for finite inputs; a select chain on lq tends to be lowered to a divergent switch)
template <typename T, int LEN, int LN = 4>
__device__ __forceinline__ T pick4(const T (&z)[LEN], int slot4, int lq) {
    T v = T(0);
#pragma unroll
    for (int l = 0; l < LN; ++l)
        if (slot4 + l < LEN) v = num<T>::fma((lq == l) ? T(1) : T(0), z[slot4 + l < LEN ? slot4 + l : 0], v);
    return v;
}

// ---- quad reductions / broadcasts of two-wide vectors (vec2, splat2, fma2: atacom_linalg.h)
// float: the two butterfly levels of 2 / 4 / 6 independent quad sums written out as v_add_f32_dpp (DPP operand
// folded into the add).  Left to itself the compiler pairs the halves into v_pk_add_f32, which cannot take a DPP
// operand, so every level became 2 x v_mov_dpp + v_pk_add + an s_nop for the DPP read-after-write hazard (2
// wait states, which the assembler does not insert inside asm: the leading s_nop covers a producer issued right
// before, the interleaving of >= 4 sums covers the second level).  Results are bitwise those of qsum().
#define ATACOM_DPP_X1 " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define ATACOM_DPP_X2 " quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
__device__ __forceinline__ void qsum_n(float& a, float& b) {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0" ATACOM_DPP_X1 "v_add_f32_dpp %1, %1, %1" ATACOM_DPP_X1 "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0" ATACOM_DPP_X2 "v_add_f32_dpp %1, %1, %1" ATACOM_DPP_X2
        : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void qsum_n(float& a, float& b, float& c, float& d) {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0" ATACOM_DPP_X1 "v_add_f32_dpp %1, %1, %1" ATACOM_DPP_X1
        "v_add_f32_dpp %2, %2, %2" ATACOM_DPP_X1 "v_add_f32_dpp %3, %3, %3" ATACOM_DPP_X1
        "v_add_f32_dpp %0, %0, %0" ATACOM_DPP_X2 "v_add_f32_dpp %1, %1, %1" ATACOM_DPP_X2
        "v_add_f32_dpp %2, %2, %2" ATACOM_DPP_X2 "v_add_f32_dpp %3, %3, %3" ATACOM_DPP_X2
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void qsum_n(float& a, float& b, float& c, float& d, float& e, float& f) {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0" ATACOM_DPP_X1 "v_add_f32_dpp %1, %1, %1" ATACOM_DPP_X1
        "v_add_f32_dpp %2, %2, %2" ATACOM_DPP_X1 "v_add_f32_dpp %3, %3, %3" ATACOM_DPP_X1
        "v_add_f32_dpp %4, %4, %4" ATACOM_DPP_X1 "v_add_f32_dpp %5, %5, %5" ATACOM_DPP_X1
        "v_add_f32_dpp %0, %0, %0" ATACOM_DPP_X2 "v_add_f32_dpp %1, %1, %1" ATACOM_DPP_X2
        "v_add_f32_dpp %2, %2, %2" ATACOM_DPP_X2 "v_add_f32_dpp %3, %3, %3" ATACOM_DPP_X2
        "v_add_f32_dpp %4, %4, %4" ATACOM_DPP_X2 "v_add_f32_dpp %5, %5, %5" ATACOM_DPP_X2
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
}
// half rows (8 lanes): a third level through row_half_mirror
#define ATACOM_DPP_X4 " row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
__device__ __forceinline__ void osum_n(float& a, float& b) {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0" ATACOM_DPP_X1 "v_add_f32_dpp %1, %1, %1" ATACOM_DPP_X1 "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0" ATACOM_DPP_X2 "v_add_f32_dpp %1, %1, %1" ATACOM_DPP_X2 "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0" ATACOM_DPP_X4 "v_add_f32_dpp %1, %1, %1" ATACOM_DPP_X4
        : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void osum_n(float& a, float& b, float& c, float& d) {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0" ATACOM_DPP_X1 "v_add_f32_dpp %1, %1, %1" ATACOM_DPP_X1
        "v_add_f32_dpp %2, %2, %2" ATACOM_DPP_X1 "v_add_f32_dpp %3, %3, %3" ATACOM_DPP_X1
        "v_add_f32_dpp %0, %0, %0" ATACOM_DPP_X2 "v_add_f32_dpp %1, %1, %1" ATACOM_DPP_X2
        "v_add_f32_dpp %2, %2, %2" ATACOM_DPP_X2 "v_add_f32_dpp %3, %3, %3" ATACOM_DPP_X2
        "v_add_f32_dpp %0, %0, %0" ATACOM_DPP_X4 "v_add_f32_dpp %1, %1, %1" ATACOM_DPP_X4
        "v_add_f32_dpp %2, %2, %2" ATACOM_DPP_X4 "v_add_f32_dpp %3, %3, %3" ATACOM_DPP_X4
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void osum_n(float& a, float& b, float& c, float& d, float& e, float& f) {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0" ATACOM_DPP_X1 "v_add_f32_dpp %1, %1, %1" ATACOM_DPP_X1
        "v_add_f32_dpp %2, %2, %2" ATACOM_DPP_X1 "v_add_f32_dpp %3, %3, %3" ATACOM_DPP_X1
        "v_add_f32_dpp %4, %4, %4" ATACOM_DPP_X1 "v_add_f32_dpp %5, %5, %5" ATACOM_DPP_X1
        "v_add_f32_dpp %0, %0, %0" ATACOM_DPP_X2 "v_add_f32_dpp %1, %1, %1" ATACOM_DPP_X2
        "v_add_f32_dpp %2, %2, %2" ATACOM_DPP_X2 "v_add_f32_dpp %3, %3, %3" ATACOM_DPP_X2
        "v_add_f32_dpp %4, %4, %4" ATACOM_DPP_X2 "v_add_f32_dpp %5, %5, %5" ATACOM_DPP_X2
        "v_add_f32_dpp %0, %0, %0" ATACOM_DPP_X4 "v_add_f32_dpp %1, %1, %1" ATACOM_DPP_X4
        "v_add_f32_dpp %2, %2, %2" ATACOM_DPP_X4 "v_add_f32_dpp %3, %3, %3" ATACOM_DPP_X4
        "v_add_f32_dpp %4, %4, %4" ATACOM_DPP_X4 "v_add_f32_dpp %5, %5, %5" ATACOM_DPP_X4
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
}
__device__ __forceinline__ void osum_n(double& a, double& b) { a = qsum<8>(a); b = qsum<8>(b); }
__device__ __forceinline__ void osum_n(double& a, double& b, double& c, double& d) {
    a = qsum<8>(a); b = qsum<8>(b); c = qsum<8>(c); d = qsum<8>(d);
}
__device__ __forceinline__ void osum_n(double& a, double& b, double& c, double& d, double& e, double& f) {
    a = qsum<8>(a); b = qsum<8>(b); c = qsum<8>(c); d = qsum<8>(d); e = qsum<8>(e); f = qsum<8>(f);
}
// lane pairs: one butterfly level
__device__ __forceinline__ void psum_n(float& a, float& b) {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0" ATACOM_DPP_X1 "v_add_f32_dpp %1, %1, %1" ATACOM_DPP_X1
        : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void psum_n(float& a, float& b, float& c, float& d) {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0" ATACOM_DPP_X1 "v_add_f32_dpp %1, %1, %1" ATACOM_DPP_X1
        "v_add_f32_dpp %2, %2, %2" ATACOM_DPP_X1 "v_add_f32_dpp %3, %3, %3" ATACOM_DPP_X1
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void psum_n(float& a, float& b, float& c, float& d, float& e, float& f) {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0" ATACOM_DPP_X1 "v_add_f32_dpp %1, %1, %1" ATACOM_DPP_X1
        "v_add_f32_dpp %2, %2, %2" ATACOM_DPP_X1 "v_add_f32_dpp %3, %3, %3" ATACOM_DPP_X1
        "v_add_f32_dpp %4, %4, %4" ATACOM_DPP_X1 "v_add_f32_dpp %5, %5, %5" ATACOM_DPP_X1
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
}
__device__ __forceinline__ void psum_n(double& a, double& b) { a = qsum<2>(a); b = qsum<2>(b); }
__device__ __forceinline__ void psum_n(double& a, double& b, double& c, double& d) {
    a = qsum<2>(a); b = qsum<2>(b); c = qsum<2>(c); d = qsum<2>(d);
}
__device__ __forceinline__ void psum_n(double& a, double& b, double& c, double& d, double& e, double& f) {
    a = qsum<2>(a); b = qsum<2>(b); c = qsum<2>(c); d = qsum<2>(d); e = qsum<2>(e); f = qsum<2>(f);
}
__device__ __forceinline__ void qsum_n(double& a, double& b) { a = qsum(a); b = qsum(b); }
__device__ __forceinline__ void qsum_n(double& a, double& b, double& c, double& d) {
    a = qsum(a); b = qsum(b); c = qsum(c); d = qsum(d);
}
__device__ __forceinline__ void qsum_n(double& a, double& b, double& c, double& d, double& e, double& f) {
    a = qsum(a); b = qsum(b); c = qsum(c); d = qsum(d); e = qsum(e); f = qsum(f);
}
// group sums of the halves of CNT (1..3) two-wide vectors, in place
template <int CNT, typename T, int LN = 4>
__device__ __forceinline__ void qsum_pairs(vec2<T> (&w)[CNT]) {
    static_assert(CNT >= 1 && CNT <= 3, "");
    T h[2 * CNT];
#pragma unroll
    for (int j = 0; j < CNT; ++j) { h[2 * j] = w[j].x; h[2 * j + 1] = w[j].y; }
    if constexpr (LN == 2) {
        if constexpr (CNT == 1) psum_n(h[0], h[1]);
        else if constexpr (CNT == 2) psum_n(h[0], h[1], h[2], h[3]);
        else psum_n(h[0], h[1], h[2], h[3], h[4], h[5]);
    } else if constexpr (LN == 8) {
        if constexpr (CNT == 1) osum_n(h[0], h[1]);
        else if constexpr (CNT == 2) osum_n(h[0], h[1], h[2], h[3]);
        else osum_n(h[0], h[1], h[2], h[3], h[4], h[5]);
    } else if constexpr (CNT == 1) qsum_n(h[0], h[1]);
    else if constexpr (CNT == 2) qsum_n(h[0], h[1], h[2], h[3]);
    else qsum_n(h[0], h[1], h[2], h[3], h[4], h[5]);
#pragma unroll
    for (int j = 0; j < CNT; ++j) w[j] = vec2<T>{h[2 * j], h[2 * j + 1]};
}
// ---- "split" group sums for 8 lanes (round 5): 2 H values of which THIS lane's half of the group reduces only H.
// The P-application sums the K + 1 vectors [nb_0 .. nb_{K-1}, x] over the group.  The vectors are interchangeable, so the
// upper half of the group (lanes 4..7) keeps them in ROTATED register order: its physical slot q holds vector (q + H) mod 2H.
// With X = slots 0..H-1 and Y = slots H..2H-1 one level across the halves, X_k += mirror(Y_k), then hands every half the
// partial sums of "its" H vectors over both halves' columns (lower half: vectors 0..H-1, upper half: H..2H-1 -- the partner's
// Y_k is the same vector as my X_k); two quad levels finish them, and Y_k = mirror(X_k) returns the other H totals:
// 4 H cross-lane instructions instead of the butterfly's 6 H, the same bits in all eight lanes (a vector's total is formed
// in one half and copied).  h[0..2H-1] in, totals out (in the lane's own physical order).
#define ATACOM_DPP_HM " row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
__device__ __forceinline__ void split_sum(float (&h)[6]) {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %3, %0" ATACOM_DPP_HM "v_add_f32_dpp %1, %4, %1" ATACOM_DPP_HM "v_add_f32_dpp %2, %5, %2" ATACOM_DPP_HM
        "v_add_f32_dpp %0, %0, %0" ATACOM_DPP_X1 "v_add_f32_dpp %1, %1, %1" ATACOM_DPP_X1 "v_add_f32_dpp %2, %2, %2" ATACOM_DPP_X1
        "v_add_f32_dpp %0, %0, %0" ATACOM_DPP_X2 "v_add_f32_dpp %1, %1, %1" ATACOM_DPP_X2 "v_add_f32_dpp %2, %2, %2" ATACOM_DPP_X2
        "v_mov_b32_dpp %3, %0" ATACOM_DPP_HM "v_mov_b32_dpp %4, %1" ATACOM_DPP_HM "v_mov_b32_dpp %5, %2" ATACOM_DPP_HM
        : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "+v"(h[4]), "+v"(h[5]));
}
__device__ __forceinline__ void split_sum(float (&h)[4]) {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %2, %0" ATACOM_DPP_HM "v_add_f32_dpp %1, %3, %1" ATACOM_DPP_HM "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0" ATACOM_DPP_X1 "v_add_f32_dpp %1, %1, %1" ATACOM_DPP_X1 "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0" ATACOM_DPP_X2 "v_add_f32_dpp %1, %1, %1" ATACOM_DPP_X2 "s_nop 0\n\t"
        "v_mov_b32_dpp %2, %0" ATACOM_DPP_HM "v_mov_b32_dpp %3, %1" ATACOM_DPP_HM
        : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]));
}
template <int H2>
__device__ __forceinline__ void split_sum(double (&h)[H2]) {
    constexpr int H = H2 / 2;
#pragma unroll
    for (int k = 0; k < H; ++k) {
        double x = h[k] + dpp_mov<DPP_ROW_HALF_MIRROR>(h[H + k]);
        x = x + dpp_mov<0xB1>(x);
        x = x + dpp_mov<0x4E>(x);
        h[k] = x;
        h[H + k] = dpp_mov<DPP_ROW_HALF_MIRROR>(x);
    }
}
#ifndef ATACOM_P_RELABEL
#define ATACOM_P_RELABEL 1          // -DATACOM_P_RELABEL=0: the A/B build with the butterfly sums in the P-application
#endif

template <int O, int LN = 4, typename T> __device__ __forceinline__ vec2<T> qbcast2(vec2<T> v) {
    return vec2<T>{qbcast<O, LN>(v.x), qbcast<O, LN>(v.y)};
}
// value held by lane L (compile time) of the group
template <int L, int LN = 4, typename V> __device__ __forceinline__ V qfrom(V v) { return qbcast<L, LN>(v); }

// ---------------------------------------------------------------------------------------------------------------
// Column layout of the lane-group solver ("column 0 replicated"):
//   column 0 of an N-column matrix (and coordinate 0 of a length-N vector) is held by EVERY lane of the group;
//   column c >= 1 lives in lane (c - 1) % LN, slot (c - 1) / LN;  S = ceil((N - 1) / LN) slots per lane.
// Why: the shapes of this workload are N = 17 (iiwa) and N = 9 (planar) -- one more than a multiple of 4 and 8.  With
// all N columns split, 4 lanes need 5 slots for 17 columns (3 of 20 wasted) and 8 lanes need 3 (7 of 24 wasted), and
// every slot-proportional loop of the solver pays for the padding.  Keeping one column replicated leaves 16 (8)
// columns that split without remainder: 4 (2) slots for 4 lanes, 2 (1) for 8 lanes.  The replicated column costs one
// extra (redundant, bitwise identical) row-vector of work, and it is the natural choice: column 0 is consumed by the
// very first reflector pair, after which no step touches it again, and reading "my pivot column" needs no broadcast.
__host__ __device__ constexpr int split_slots(int n, int ln) { return (n - 1 + ln - 1) / ln; }

// a: M x N in the layout above (aget(row, slot) = A[row][LN*slot + lq + 1], zeros past N; a0get(row) = A[row][0]);
// y replicated.  On return  x0, x[slot]  and  nb0[k], nb[slot][k]  are  A^+ y  and the orthonormal null basis in the
// same layout (see bidiag_solve_null in atacom_linalg.h for the algorithm and its provenance).
// The matrix and right-hand side are handed over as generators called with compile-time indices
// (std::integral_constant), so the operands are born in their register pairs: an intermediate T a[M][S] array here
// made the optimiser merge neighbouring stores and then fail to dissolve the array, which put it in scratch / LDS
// (+10 us per step, measured).
// PRE0: the first right reflector G(0) has been generated and applied by the caller (it is the same for the four
// physics sub-steps of a step whenever row 0 carries no slack entry: env_step, g0_precompute) -- row 0 of the matrix
// handed over already holds the reflector vector, the rows below are already updated, pre_d0 / pre_tau0 are its
// beta / tau.  Saves the reflector generation and its application to M - 1 rows in every sub-step.
template <typename T, int M, int N, int LN, bool PRE0 = false, typename AF, typename A0F, typename YF>
__device__ __forceinline__ void bidiag_solve_null_quad_inl(AF&& aget, A0F&& a0get, YF&& yget, T& x0,
                                                           T (&x)[split_slots(N, LN)], T (&nb0)[N - M],
                                                           T (&nb)[split_slots(N, LN)][N - M], const int lq,
                                                           const T pre_d0 = T(0), const T pre_tau0 = T(0)) {
    constexpr int S = split_slots(N, LN), K = N - M;
    constexpr int MP = (M + 1) / 2;          // row pairs (a zero row pads an odd M: it is a fixed point of every step)
    constexpr int KP = (K + 2) / 2;          // pairs over the K null vectors + x
    using V2 = vec2<T>;
    V2 a2[S][MP], c0[MP], y2[MP];
    static_for<0, MP>([&](auto pc) {
        constexpr int p = decltype(pc)::value;
        constexpr int r0 = 2 * p, r1 = (2 * p + 1 < M) ? 2 * p + 1 : 2 * p;
        constexpr bool two = 2 * p + 1 < M;
        const T ya = yget(std::integral_constant<int, r0>{});
        const T yb = two ? yget(std::integral_constant<int, r1>{}) : T(0);
        y2[p] = V2{ya, yb};
        const T ca = a0get(std::integral_constant<int, r0>{});
        const T cb = two ? a0get(std::integral_constant<int, r1>{}) : T(0);
        c0[p] = V2{ca, cb};
        static_for<0, S>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            const T va = aget(std::integral_constant<int, r0>{}, sc);
            const T vb = two ? aget(std::integral_constant<int, r1>{}, sc) : T(0);
            a2[s][p] = V2{va, vb};
        });
    });
    T d[M], e[M], taup[M];
    static_for<0, M>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr bool first = (i == 0);                              // the pivot column is the replicated one
        constexpr int si = first ? 0 : (i - 1) / LN, li = first ? 0 : (i - 1) % LN;   // home of column i (i >= 1)
        constexpr int s0 = first ? 0 : si;                            // first slot holding a column >= i
        constexpr int pi = i / 2, hi = i % 2;      // row i is half hi of row pair pi
        // ---- right reflector G(i) from row i, columns > i
        ATACOM_MARK("G_larfg");
        T vrow[S];
#pragma unroll
        for (int s = 0; s < S; ++s) vrow[s] = a2[s][pi][hi];
        constexpr bool pre = first && PRE0;                           // G(0) came with the matrix
        if constexpr (pre) { d[0] = pre_d0; taup[0] = pre_tau0; }
        T tp = pre_tau0;
        if constexpr (!pre) {
        T part;
        if constexpr (first) {
            part = vrow[0] * vrow[0];
#pragma unroll
            for (int s = 1; s < S; ++s) part = num<T>::fma(vrow[s], vrow[s], part);
        } else {
            part = (lq > li) ? vrow[si] * vrow[si] : T(0);
#pragma unroll
            for (int s = si + 1; s < S; ++s) part = num<T>::fma(vrow[s], vrow[s], part);
        }
        const T ss = qsum<LN>(part);
        T alpha;
        if constexpr (first) alpha = c0[0].x;
        else alpha = qfrom<li, LN>(vrow[si]);
        T beta;
        const T sc = larfg_scale(alpha, ss, beta, tp);
        d[i] = beta;
        taup[i] = tp;
        // row i becomes the FULL reflector vector over the split columns: 0 for c < i, 1 at c == i, v for c > i
        // (for i == 0 the unit entry sits in the replicated column and stays implicit)
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if constexpr (first) vrow[s] *= sc;
            else if (s < si) vrow[s] = T(0);
            else if (s == si) vrow[s] = (lq > li) ? vrow[s] * sc : ((lq == li) ? T(1) : T(0));
            else vrow[s] *= sc;
            a2[s][pi][hi] = vrow[s];
        }
        }   // !pre
        if constexpr (i < M - 1) {
            constexpr int p0 = (i + 1) / 2;        // first row pair holding a row > i
            ATACOM_MARK("G_apply");
            if constexpr (!pre) {
            // rows > i, processed in groups of (up to) 3 row pairs so that their quad sums share one DPP sequence
            static_for<0, (MP - p0 + 2) / 3>([&](auto gc) {
                constexpr int pa = p0 + 3 * decltype(gc)::value;
                constexpr int CNT = (MP - pa) < 3 ? (MP - pa) : 3;
                V2 w[CNT];
#pragma unroll
                for (int j = 0; j < CNT; ++j) {
                    w[j] = a2[s0][pa + j] * splat2(vrow[s0]);
#pragma unroll
                    for (int s = s0 + 1; s < S; ++s) w[j] = fma2(a2[s][pa + j], splat2(vrow[s]), w[j]);
                }
                qsum_pairs<CNT, T, LN>(w);
#pragma unroll
                for (int j = 0; j < CNT; ++j) {
                    if constexpr (first) w[j] += c0[pa + j];            // the implicit unit entry of v
                    w[j] *= splat2(tp);
                    if (2 * (pa + j) <= i) w[j].x = T(0);   // the pair's first row is row i itself: leave it alone
                    if constexpr (first) c0[pa + j] -= w[j];
#pragma unroll
                    for (int s = s0; s < S; ++s) a2[s][pa + j] = fma2(-w[j], splat2(vrow[s]), a2[s][pa + j]);
                }
            });
            }   // !pre
            // ---- left reflector H(i) from column i, rows i+1..M-1: the replicated column for i == 0 (no broadcast),
            // slot si of lane li otherwise.  u over the pairs p0..: 0 for rows <= i, 1 at row i+1, scaled entries below
            ATACOM_MARK("H_larfg");
            if constexpr (i == M - 2) {
                // a single row is left below row i: nothing to annihilate, H(i) = I (dlarfg with an empty x), the
                // sub-diagonal entry is the element itself
                if constexpr (first) e[i] = (hi == 0) ? c0[p0].y : c0[p0].x;
                else e[i] = qfrom<li, LN>(hi == 0 ? a2[si][p0].y : a2[si][p0].x);
            } else {
            V2 colp[MP];
#pragma unroll
            for (int p = p0; p < MP; ++p) colp[p] = first ? c0[p] : a2[si][p];
            V2 sq = splat2(T(0));
#pragma unroll
            for (int p = p0 + 1; p < MP; ++p) sq = fma2(colp[p], colp[p], sq);
            T sup = sq.x + sq.y;
            if constexpr (hi == 1) sup = num<T>::fma(colp[p0].y, colp[p0].y, sup);   // row i+2 shares i+1's pair
            T su, alq;
            if constexpr (first) { su = sup; alq = colp[p0].y; }
            else { su = qfrom<li, LN>(sup); alq = qfrom<li, LN>(hi == 0 ? colp[p0].y : colp[p0].x); }
            T betaq, tq;
            const T scq = larfg_scale(alq, su, betaq, tq);
            e[i] = betaq;
            V2 u2[MP];
#pragma unroll
            for (int p = p0; p < MP; ++p) {
                if constexpr (first) u2[p] = colp[p] * splat2(scq);
                else u2[p] = qbcast2<li, LN>(colp[p]) * splat2(scq);
            }
            if constexpr (hi == 0) u2[p0] = V2{T(0), T(1)};
            else u2[p0].x = T(1);
            ATACOM_MARK("H_apply");
#pragma unroll
            for (int s = s0; s < S; ++s) {
                V2 acc = u2[p0] * a2[s][p0];
#pragma unroll
                for (int p = p0 + 1; p < MP; ++p) acc = fma2(u2[p], a2[s][p], acc);
                T w = (acc.x + acc.y) * tq;
                if constexpr (!first) {
                    if (s == si) w = (lq > li) ? w : T(0);      // columns <= i are not touched
                }
#pragma unroll
                for (int p = p0; p < MP; ++p) a2[s][p] = fma2(splat2(-w), u2[p], a2[s][p]);
            }
            {
                V2 acc = u2[p0] * y2[p0];
#pragma unroll
                for (int p = p0 + 1; p < MP; ++p) acc = fma2(u2[p], y2[p], acc);
                const T w = (acc.x + acc.y) * tq;
#pragma unroll
                for (int p = p0; p < MP; ++p) y2[p] = fma2(splat2(-w), u2[p], y2[p]);
            }
            }   // i < M - 2
        }
    });
    // ---- z = B^{-1} Q^T y (replicated), then split by column
    ATACOM_MARK("Z_solve");
    T z[M];
    bidiag_forward_solve<T, M>(d, e, [&](int i) { return y2[i / 2][i % 2]; }, z);
    // [nb_0 .. nb_{K-1}, x] as pairs over the vector index; coordinate 0 replicated (nx0), coordinates >= 1 split
    // RELABEL (8 lanes, K + 1 even): the upper half of the group holds vector (q + H) mod 2H in physical slot q (split_sum)
    // (K + 1 = 6, the iiwa shape, only: with four vectors the split form is 11 instructions with their wait states against the
    // butterfly's 12 and a longer dependent chain -- the planar T-step kernel measured 8.8 instead of 8.3 us per step)
    constexpr bool RELABEL = ATACOM_P_RELABEL && LN == 8 && K + 1 == 6;
    constexpr int HV = (K + 1) / 2;
    const bool hf = RELABEL && (lq >= LN / 2);
    V2 nx[S][KP], nx0[KP];
    if constexpr (!RELABEL) {
#pragma unroll
    for (int j = 0; j < KP; ++j) nx0[j] = V2{(2 * j == K) ? z[0] : T(0), (2 * j + 1 == K) ? z[0] : T(0)};
#pragma unroll
    for (int s = 0; s < S; ++s) {
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            T h[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int k = 2 * j + t;
                h[t] = (k < K) ? ((LN * s + lq + 1 == M + k) ? T(1) : T(0))
                               : ((k == K) ? pick4<T, M, LN>(z, LN * s + 1, lq) : T(0));
            }
            nx[s][j] = V2{h[0], h[1]};
        }
    }
    } else {
        // physical slot q: vector q in lanes 0..3, vector (q + HV) mod (K + 1) in lanes 4..7.  The unit entry of null vector k
        // sits at coordinate M + k, i.e. in ONE known lane: the test below is a compile-time constant per (slot, lane half)
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            T h[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int q = 2 * j + t;
                h[t] = (q == K) ? (hf ? T(0) : z[0]) : (((q + HV) % (K + 1) == K) ? (hf ? z[0] : T(0)) : T(0));
            }
            nx0[j] = V2{h[0], h[1]};
        }
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const T zv = pick4<T, M, LN>(z, LN * s + 1, lq);
#pragma unroll
            for (int j = 0; j < KP; ++j) {
                T h[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int q = 2 * j + t;
                    T v = T(0);
#pragma unroll
                    for (int up = 0; up < 2; ++up) {                   // the vector this slot holds in the lower / upper half
                        const int k = up ? (q + HV) % (K + 1) : q;
                        if (k < K) {
                            const int c = M + k, own = (c - 1) % LN;       // coordinate of the unit entry, its lane
                            if ((c - 1) / LN == s && (own >= LN / 2) == (up == 1)) v = (lq == own) ? T(1) : v;
                        } else {
                            v = (hf == (up == 1)) ? zv : v;
                        }
                    }
                    h[t] = v;
                }
                nx[s][j] = V2{h[0], h[1]};
            }
        }
    }
    // ---- [nb | x] <- G(1) ... G(M) [nb | x]
    ATACOM_MARK("P_apply");
    static_for<0, M>([&](auto kc) {
        constexpr int i = M - 1 - decltype(kc)::value;
        constexpr bool first = (i == 0);
        constexpr int s0 = first ? 0 : (i - 1) / LN;
        T vrow[S];
#pragma unroll
        for (int s = s0; s < S; ++s) vrow[s] = a2[s][i / 2][i % 2];
        static_for<0, (KP + 2) / 3>([&](auto gc) {
            constexpr int ja = 3 * decltype(gc)::value;
            constexpr int CNT = (KP - ja) < 3 ? (KP - ja) : 3;
            V2 w[CNT];
#pragma unroll
            for (int j = 0; j < CNT; ++j) {
                w[j] = splat2(vrow[s0]) * nx[s0][ja + j];
#pragma unroll
                for (int s = s0 + 1; s < S; ++s) w[j] = fma2(splat2(vrow[s]), nx[s][ja + j], w[j]);
            }
            if constexpr (RELABEL) {
                static_assert(!RELABEL || CNT == KP, "all K + 1 vectors in one reduction");
                T h[2 * CNT];
#pragma unroll
                for (int j = 0; j < CNT; ++j) { h[2 * j] = w[j].x; h[2 * j + 1] = w[j].y; }
                split_sum(h);
#pragma unroll
                for (int j = 0; j < CNT; ++j) w[j] = V2{h[2 * j], h[2 * j + 1]};
            } else {
                qsum_pairs<CNT, T, LN>(w);
            }
#pragma unroll
            for (int j = 0; j < CNT; ++j) {
                if constexpr (first) w[j] += nx0[ja + j];
                w[j] *= splat2(taup[i]);
                if constexpr (first) nx0[ja + j] -= w[j];
#pragma unroll
                for (int s = s0; s < S; ++s) nx[s][ja + j] = fma2(-w[j], splat2(vrow[s]), nx[s][ja + j]);
            }
        });
    });
    ATACOM_MARK("P_done");
    if constexpr (!RELABEL) {
    x0 = nx0[K / 2][K % 2];
#pragma unroll
    for (int k = 0; k < K; ++k) nb0[k] = nx0[k / 2][k % 2];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        x[s] = nx[s][K / 2][K % 2];
#pragma unroll
        for (int k = 0; k < K; ++k) nb[s][k] = nx[s][k / 2][k % 2];
    }
    } else {
        // back to the vectors' own order: vector k sits in slot k (lanes 0..3) or (k + HV) mod (K + 1) (lanes 4..7)
        auto at = [&](const V2 (&v)[KP], int k) -> T {
            const int qu = (k + HV) % (K + 1);
            const T lo = v[k / 2][k % 2], up = v[qu / 2][qu % 2];
            return hf ? up : lo;
        };
        x0 = at(nx0, K);
#pragma unroll
        for (int k = 0; k < K; ++k) nb0[k] = at(nx0, k);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            x[s] = at(nx[s], K);
#pragma unroll
            for (int k = 0; k < K; ++k) nb[s][k] = at(nx[s], k);
        }
    }
}

// Chart (rref + Nc @ alpha) on the null basis in the layout above (coordinate 0 replicated: nb0, the rest split).
// Semantics: null_space_coordinate.py:40-79 with row_vectors=False (scan the columns left to right; a column
// whose largest entry over the not-yet-pivot rows is <= tol is skipped and those entries are zeroed; otherwise
// the arg-max row becomes the next pivot row, is scaled to 1 and eliminated from every other row; stop after K
// pivots).
//
// Formulated as K PIVOT ROUNDS instead of N column steps.  Between two pivots the matrix does not change
// (a skip only zeroes entries of the skipped column), so "the next pivot column" is simply the first column
// >= j0 whose masked column-max exceeds tol -- all columns are tested at once (each lane its own slots, one
// group-min), and the cost is a fixed K rounds with no data-dependent trip count: the column-by-column form
// spent most of its time in the ~11% of environments that walk almost all N columns looking for their last
// pivot, and a wavefront runs as long as its slowest quad.
//   * the skip-zeroing is never materialised: eliminations are gated to columns >= the pivot column, so a
//     skipped column is frozen from the moment it is passed, and the final contraction only takes, for
//     column c, the rows that were already pivot rows when c was passed (jrow[r] <= c);
//   * rows ARE swapped like the reference's (pivot row <-> row t in round t), so "not yet a pivot row" is simply
//     r >= t, a compile-time range: no used[] masks in the column scan or the arg-max, alpha[r] pairs with row r,
//     and np.argmax's first-maximum tie-break is reproduced in the reference's own row order.
// out0 = (Nc @ alpha)[0] (replicated), out[slot] = (Nc @ alpha)[LN*slot + lq + 1].
template <typename T, int N, int K, int LN>
__device__ __forceinline__ void rref_apply_quad_inl(T (&nb0)[K], T (&nb)[split_slots(N, LN)][K], const T (&alpha)[K],
                                                    T tol, T& out0, T (&out)[split_slots(N, LN)], const int lq) {
    constexpr int S = split_slots(N, LN);
    constexpr int BIG = 1 << 20;
    int col[S];
#pragma unroll
    for (int s = 0; s < S; ++s) col[s] = LN * s + lq + 1;
    int jrow[K];                       // column at which row r became a pivot row (BIG: never)
    int j0 = 0;
    static_for<0, K>([&](auto tc) {
        constexpr int t = decltype(tc)::value;          // round t: rows < t are pivot rows, rows >= t are not
        // ---- next pivot column: first column >= j0 with max |entry| over the rows >= t above tol
        ATACOM_MARK("R_scan");
        int cand = BIG;
#pragma unroll
        for (int s = S - 1; s >= 0; --s) {
            T cm = num<T>::abs(nb[s][t]);
#pragma unroll
            for (int r = t + 1; r < K; ++r) cm = num<T>::max(cm, num<T>::abs(nb[s][r]));
            const bool e = (cm > tol) && (col[s] >= j0);
            cand = e ? col[s] : cand;
        }
        if constexpr (t == 0) {
            // the replicated column 0 can only be the pivot column of round 0: afterwards it has been passed (j0 >= 1),
            // or -- if round 0 found no column at all -- nothing changes any more and it stays below the tolerance
            T cm = num<T>::abs(nb0[t]);
#pragma unroll
            for (int r = t + 1; r < K; ++r) cm = num<T>::max(cm, num<T>::abs(nb0[r]));
            cand = (cm > tol) ? 0 : cand;
        }
        int jmin = min(cand, dpp_mov<0xB1>(cand));
        if constexpr (LN >= 4) jmin = min(jmin, dpp_mov<0x4E>(jmin));
        if constexpr (LN == 8) jmin = min(jmin, dpp_mov<DPP_ROW_HALF_MIRROR>(jmin));
        const bool found = jmin < BIG;
        // ---- that column's K entries, replicated over the group (zeros when no column was found)
        ATACOM_MARK("R_gather");
        T w[S];
#pragma unroll
        for (int s = 0; s < S; ++s) w[s] = (col[s] == jmin) ? T(1) : T(0);
        const T w0 = (jmin == 0) ? T(1) : T(0);
        T f[K];
#pragma unroll
        for (int r = 0; r < K; ++r) {
            T v = w[0] * nb[0][r];
#pragma unroll
            for (int s = 1; s < S; ++s) v = num<T>::fma(w[s], nb[s][r], v);
            if constexpr (t == 0) f[r] = num<T>::fma(w0, nb0[r], qsum<LN>(v));
            else f[r] = qsum<LN>(v);                      // jmin >= 1 from round 1 on
        }
        // ---- pivot row: first arg-max of |f| over rows t..K-1 (np.argmax semantics in the reference's row order)
        ATACOM_MARK("R_argmax");
        T p = T(-1);
        int kk = t;
#pragma unroll
        for (int r = t; r < K; ++r) {
            const T av = num<T>::abs(f[r]);
            const bool gt = av > p;                              // strict: first maximum, like np.argmax
            p = gt ? av : p;
            kk = gt ? r : kk;
        }
        bool isp[K];
        T pj = T(1);
#pragma unroll
        for (int r = t; r < K; ++r) {
            isp[r] = found && (kk == r);
            pj = isp[r] ? f[r] : pj;
        }
        const T inv = num<T>::rcp(pj);                       // 1 when nothing was found
        // ---- swap rows (pivot row <-> row t), as the reference does: picks the pivot row by a one-hot blend
        // (row t itself when nothing was found), drops the old row t where the pivot row was
        T oh[K];
#pragma unroll
        for (int r = t; r < K; ++r) oh[r] = (isp[r] || (r == t && !found)) ? T(1) : T(0);
        const T ft = f[t];
#pragma unroll
        for (int r = t + 1; r < K; ++r) f[r] = isp[r] ? ft : f[r];
        ATACOM_MARK("R_update");
        auto update = [&](T (&colv)[K], const bool right_of_pivot) {
            T rowp = oh[t] * colv[t];
#pragma unroll
            for (int r = t + 1; r < K; ++r) rowp = num<T>::fma(oh[r], colv[r], rowp);
            const T oldt = colv[t];
#pragma unroll
            for (int r = t + 1; r < K; ++r) colv[r] = isp[r] ? oldt : colv[r];
            // scaled pivot row: zero left of the pivot column (those columns are frozen, see the header comment)
            const T pr = (right_of_pivot || !found) ? rowp * inv : T(0);
            colv[t] = pr;
#pragma unroll
            for (int r = 0; r < K; ++r)
                if (r != t) colv[r] = num<T>::fma(found ? -f[r] : T(0), pr, colv[r]);
        };
#pragma unroll
        for (int s = 0; s < S; ++s) update(nb[s], col[s] >= jmin);
        // column 0 lies left of every later pivot column, and the contraction below reads it only in a row whose pivot
        // column it is -- row 0 of round 0, which no later row swap (rows >= t >= 1) touches: rounds >= 1 leave it alone
        if constexpr (t == 0) update(nb0, jmin == 0);
        jrow[t] = jmin;
        j0 = found ? jmin + 1 : j0;
    });
    ATACOM_MARK("R_contract");
#pragma unroll
    for (int s = 0; s < S; ++s) {
        T v = T(0);
#pragma unroll
        for (int r = 0; r < K; ++r) v = num<T>::fma((jrow[r] <= col[s]) ? alpha[r] : T(0), nb[s][r], v);
        out[s] = v;
    }
    {
        T v = T(0);
#pragma unroll
        for (int r = 0; r < K; ++r) v = num<T>::fma((jrow[r] <= 0) ? alpha[r] : T(0), nb0[r], v);
        out0 = v;
    }
}

// ---- entry points: everything inlined into the one straight-line kernel, float and double alike.
// History: rounds 1 - 5 kept the two big pieces as real (noinline) functions in the double build, after hipcc 7.2 had been seen
// to miscompile the fully inlined k_step<double, Iiwa, 4, false> in round 1 (found by tests/test_gpu_parity.py::
// test_refresh_and_exact_bias_variants_against_oracle).  The calls cost the double kernels their registers: arrays handed over
// by reference live in scratch (1.6 - 2.6 KB per lane; 183 of 331 double kernels), the callee-save convention pins 512
// registers -- 112.7 us per step at 8192 iiwa environments on 8 lanes.  Round 6 inlined them again: the same test and the
// other 124 float64 tests pass at 1e-8 on every mapping, the 8-lane step kernel holds 256 + 228 registers and no scratch and
// runs 50.2 us (profiles/r06_f64_inline.md).
template <typename T, int M, int N, int LN = 4, bool PRE0 = false, typename AF, typename A0F, typename YF>
__device__ __forceinline__ void bidiag_solve_null_quad(AF&& aget, A0F&& a0get, YF&& yget, T& x0,
                                                       T (&x)[split_slots(N, LN)], T (&nb0)[N - M],
                                                       T (&nb)[split_slots(N, LN)][N - M], const int lq,
                                                       const T pre_d0 = T(0), const T pre_tau0 = T(0)) {
    bidiag_solve_null_quad_inl<T, M, N, LN, PRE0>(aget, a0get, yget, x0, x, nb0, nb, lq, pre_d0, pre_tau0);
}
template <typename T, int N, int K, int LN = 4>
__device__ __forceinline__ void rref_apply_quad(T (&nb0)[K], T (&nb)[split_slots(N, LN)][K], const T (&alpha)[K], T tol,
                                                T& out0, T (&out)[split_slots(N, LN)], const int lq) {
    rref_apply_quad_inl<T, N, K, LN>(nb0, nb, alpha, tol, out0, out, lq);
}

}  // namespace atacom
