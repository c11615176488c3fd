// Lane-group variant of the null-space solver: FOUR (a DPP quad) or TWO (a lane pair) LANES PER ENVIRONMENT.
// (Written and described for the quad; the group size is the template parameter LN, see qbcast / qsum below.)
//
// Why (measured in profiles/): a wave64 vector instruction occupies its SIMD for 4 clocks and a
// lone wave already saturates it, so the step time is (vector instructions per wave) x ~5 clk.  With one environment
// per lane the headline batch of 8192 environments is 128 wavefronts on 1024 SIMDs, each running the whole 24 k-
// instruction step (52 us).  Splitting every environment over the 4 lanes of a DPP quad puts 512 waves to work at
// ~0.6x the instructions per wave (30 us); beyond 16384 environments (1024 waves) the lane mapping wins again
// because its total instruction count is lower.
//
// Data distribution inside a quad (lq = lane & 3):
//   * matrices with N columns (J_c: M x N, null basis: N x K) are split BY COLUMN: column c lives in lane
//     c % 4, "slot" c / 4  (S = ceil(N / 4) slots per lane; slots past N hold zeros);
//   * vectors over the M rows (rhs y, the bidiagonal d / e, left reflectors u) are REPLICATED in the four
//     lanes and computed redundantly -- their values stay bitwise identical across the quad because every
//     cross-lane sum uses the same commutative butterfly;
//   * cross-lane traffic is DPP only (quad_perm): a broadcast is one v_mov_dpp, a quad sum two v_add_f32_dpp;
//     no LDS, no ds_bpermute, no barriers;
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

// Everything below is written for a GROUP of LN lanes per environment: LN = 4 (a DPP quad) or LN = 2 (a lane pair, two
// environments per quad).  The pair form has half the redundancy and one butterfly level instead of two; it is the
// mapping of choice when there are enough environments to fill the chip with 32-env waves but not with 64-env waves.
//
// value of lane O (0..LN-1) of the group, in all its lanes
template <int O, int LN = 4, typename V>
__device__ __forceinline__ V qbcast(V v) {
    static_assert(LN == 4 || LN == 2, "");
    // quad_perm: LN = 4 -> [O,O,O,O];  LN = 2 -> [O,O,2+O,2+O]
    return dpp_mov<(LN == 4) ? O * 0x55 : (O * 0x05 + (2 + O) * 0x50)>(v);
}
// sum over the group, identical bits in all its lanes
template <int LN = 4, typename T>
__device__ __forceinline__ T qsum(T v) {
    const T s1 = v + dpp_mov<0xB1>(v);     // quad_perm [1,0,3,2]
    if constexpr (LN == 2) return s1;
    else return s1 + dpp_mov<0x4E>(s1);    // quad_perm [2,3,0,1]
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
    } else if constexpr (CNT == 1) qsum_n(h[0], h[1]);
    else if constexpr (CNT == 2) qsum_n(h[0], h[1], h[2], h[3]);
    else qsum_n(h[0], h[1], h[2], h[3], h[4], h[5]);
#pragma unroll
    for (int j = 0; j < CNT; ++j) w[j] = vec2<T>{h[2 * j], h[2 * j + 1]};
}
template <int O, int LN = 4, typename T> __device__ __forceinline__ vec2<T> qbcast2(vec2<T> v) {
    return vec2<T>{qbcast<O, LN>(v.x), qbcast<O, LN>(v.y)};
}
// value held by lane L (compile time) of the group
template <int L, int LN = 4, typename V> __device__ __forceinline__ V qfrom(V v) { return qbcast<L, LN>(v); }

// a: M x N split by column over the quad (a[r][slot] = A[r][4*slot+lq], zeros past N); y replicated.
// On return x[slot] and nb[slot][k] are the column-split  A^+ y  and orthonormal null basis (see
// bidiag_solve_null in atacom_linalg.h for the algorithm and its provenance).
// The matrix and right-hand side are handed over as generators  aget(row, slot) / yget(row)  called with
// compile-time indices (std::integral_constant), so the operands are born in their register pairs: an
// intermediate T a[M][S] array here made the optimiser merge neighbouring stores and then fail to dissolve the
// array, which put it in scratch / LDS (+10 us per step, measured).
template <typename T, int M, int N, int LN, typename AF, typename YF>
__device__ __forceinline__ void bidiag_solve_null_quad_inl(AF&& aget, YF&& yget, T (&x)[(N + LN - 1) / LN],
                                                       T (&nb)[(N + LN - 1) / LN][N - M], const int lq) {
    constexpr int S = (N + LN - 1) / LN, K = N - M;
    constexpr int MP = (M + 1) / 2;          // row pairs (a zero row pads an odd M: it is a fixed point of every step)
    constexpr int KP = (K + 2) / 2;          // pairs over the K null vectors + x
    using V2 = vec2<T>;
    V2 a2[S][MP], y2[MP];
    static_for<0, MP>([&](auto pc) {
        constexpr int p = decltype(pc)::value;
        constexpr int r0 = 2 * p, r1 = (2 * p + 1 < M) ? 2 * p + 1 : 2 * p;
        constexpr bool two = 2 * p + 1 < M;
        const T ya = yget(std::integral_constant<int, r0>{});
        const T yb = two ? yget(std::integral_constant<int, r1>{}) : T(0);
        y2[p] = V2{ya, yb};
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
        constexpr int si = i / LN, li = i % LN;    // column i lives in slot si of lane li
        constexpr int pi = i / 2, hi = i % 2;      // row i is half hi of row pair pi
        // ---- right reflector G(i) from row i, columns > i
        T vrow[S];
#pragma unroll
        for (int s = 0; s < S; ++s) vrow[s] = a2[s][pi][hi];
        T part = (lq > li) ? vrow[si] * vrow[si] : T(0);
#pragma unroll
        for (int s = si + 1; s < S; ++s) part = num<T>::fma(vrow[s], vrow[s], part);
        const T ss = qsum<LN>(part);
        const T alpha = qfrom<li, LN>(vrow[si]);
        T beta, tp;
        const T sc = larfg_scale(alpha, ss, beta, tp);
        d[i] = beta;
        taup[i] = tp;
        // row i becomes the FULL reflector vector: 0 for c < i, 1 at c == i, v for c > i
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (s < si) vrow[s] = T(0);
            else if (s == si) vrow[s] = (lq > li) ? vrow[s] * sc : ((lq == li) ? T(1) : T(0));
            else vrow[s] *= sc;
            a2[s][pi][hi] = vrow[s];
        }
        if constexpr (i < M - 1) {
            constexpr int p0 = (i + 1) / 2;        // first row pair holding a row > i
            // rows > i, processed in groups of (up to) 3 row pairs so that their quad sums share one DPP sequence
            static_for<0, (MP - p0 + 2) / 3>([&](auto gc) {
                constexpr int pa = p0 + 3 * decltype(gc)::value;
                constexpr int CNT = (MP - pa) < 3 ? (MP - pa) : 3;
                V2 w[CNT];
#pragma unroll
                for (int j = 0; j < CNT; ++j) {
                    w[j] = a2[si][pa + j] * splat2(vrow[si]);
#pragma unroll
                    for (int s = si + 1; s < S; ++s) w[j] = fma2(a2[s][pa + j], splat2(vrow[s]), w[j]);
                }
                qsum_pairs<CNT, T, LN>(w);
#pragma unroll
                for (int j = 0; j < CNT; ++j) {
                    w[j] *= splat2(tp);
                    if (2 * (pa + j) <= i) w[j].x = T(0);   // the pair's first row is row i itself: leave it alone
#pragma unroll
                    for (int s = si; s < S; ++s) a2[s][pa + j] = fma2(-w[j], splat2(vrow[s]), a2[s][pa + j]);
                }
            });
            // ---- left reflector H(i) from column i (slot si of lane li), rows i+1..M-1
            // u over the pairs p0..: 0 for rows <= i, 1 at row i+1, column entries * scale below
            V2 sq = splat2(T(0));
#pragma unroll
            for (int p = p0 + 1; p < MP; ++p) sq = fma2(a2[si][p], a2[si][p], sq);
            T sup = sq.x + sq.y;
            if constexpr (hi == 1) sup = num<T>::fma(a2[si][p0].y, a2[si][p0].y, sup);   // row i+2 shares i+1's pair
            const T su = qfrom<li, LN>(sup);
            const T alq = qfrom<li, LN>(hi == 0 ? a2[si][p0].y : a2[si][p0].x);
            T betaq, tq;
            const T scq = larfg_scale(alq, su, betaq, tq);
            e[i] = betaq;
            V2 u2[MP];
#pragma unroll
            for (int p = p0; p < MP; ++p) u2[p] = qbcast2<li, LN>(a2[si][p]) * splat2(scq);
            if constexpr (hi == 0) u2[p0] = V2{T(0), T(1)};
            else u2[p0].x = T(1);
#pragma unroll
            for (int s = si; s < S; ++s) {
                V2 acc = u2[p0] * a2[s][p0];
#pragma unroll
                for (int p = p0 + 1; p < MP; ++p) acc = fma2(u2[p], a2[s][p], acc);
                T w = (acc.x + acc.y) * tq;
                if (s == si) w = (lq > li) ? w : T(0);          // columns <= i are not touched
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
        }
    });
    // ---- z = B^{-1} Q^T y (replicated), then split by column
    T z[M];
    z[0] = num<T>::div(y2[0].x, d[0]);
#pragma unroll
    for (int i = 1; i < M; ++i) z[i] = num<T>::div(num<T>::fma(-e[i - 1], z[i - 1], y2[i / 2][i % 2]), d[i]);
    // [nb_0 .. nb_{K-1}, x] as pairs over the vector index
    V2 nx[S][KP];
#pragma unroll
    for (int s = 0; s < S; ++s) {
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            T h[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int k = 2 * j + t;
                h[t] = (k < K) ? ((LN * s + lq == M + k) ? T(1) : T(0)) : ((k == K) ? pick4<T, M, LN>(z, LN * s, lq) : T(0));
            }
            nx[s][j] = V2{h[0], h[1]};
        }
    }
    // ---- [nb | x] <- G(1) ... G(M) [nb | x]
    static_for<0, M>([&](auto kc) {
        constexpr int i = M - 1 - decltype(kc)::value;
        constexpr int si = i / LN;
        T vrow[S];
#pragma unroll
        for (int s = si; s < S; ++s) vrow[s] = a2[s][i / 2][i % 2];
        static_for<0, (KP + 2) / 3>([&](auto gc) {
            constexpr int ja = 3 * decltype(gc)::value;
            constexpr int CNT = (KP - ja) < 3 ? (KP - ja) : 3;
            V2 w[CNT];
#pragma unroll
            for (int j = 0; j < CNT; ++j) {
                w[j] = splat2(vrow[si]) * nx[si][ja + j];
#pragma unroll
                for (int s = si + 1; s < S; ++s) w[j] = fma2(splat2(vrow[s]), nx[s][ja + j], w[j]);
            }
            qsum_pairs<CNT, T, LN>(w);
#pragma unroll
            for (int j = 0; j < CNT; ++j) {
                w[j] *= splat2(taup[i]);
#pragma unroll
                for (int s = si; s < S; ++s) nx[s][ja + j] = fma2(-w[j], splat2(vrow[s]), nx[s][ja + j]);
            }
        });
    });
#pragma unroll
    for (int s = 0; s < S; ++s) {
        x[s] = nx[s][K / 2][K % 2];
#pragma unroll
        for (int k = 0; k < K; ++k) nb[s][k] = nx[s][k / 2][k % 2];
    }
}

// Chart (rref + Nc @ alpha) on the column-split null basis.
// Semantics: null_space_coordinate.py:40-79 with row_vectors=False (scan the columns left to right; a column
// whose largest entry over the not-yet-pivot rows is <= tol is skipped and those entries are zeroed; otherwise
// the arg-max row becomes the next pivot row, is scaled to 1 and eliminated from every other row; stop after K
// pivots).
//
// Formulated as K PIVOT ROUNDS instead of N column steps.  Between two pivots the matrix does not change
// (a skip only zeroes entries of the skipped column), so "the next pivot column" is simply the first column
// >= j0 whose masked column-max exceeds tol -- all columns are tested at once (each lane its own slots, one
// quad-min), and the cost is a fixed K rounds with no data-dependent trip count: the column-by-column form
// spent most of its time in the ~11% of environments that walk almost all N columns looking for their last
// pivot, and a wavefront runs as long as its slowest quad.
//   * the skip-zeroing is never materialised: eliminations are gated to columns >= the pivot column, so a
//     skipped column is frozen from the moment it is passed, and the final contraction only takes, for
//     column c, the rows that were already pivot rows when c was passed (jrow[r] <= c);
//   * rows ARE swapped like the reference's (pivot row <-> row t in round t), so "not yet a pivot row" is simply
//     r >= t, a compile-time range: no used[] masks in the column scan or the arg-max, alpha[r] pairs with row r,
//     and np.argmax's first-maximum tie-break is reproduced in the reference's own row order.
// out[slot] = (Nc @ alpha)[4*slot + lq].
template <typename T, int N, int K, int LN>
__device__ __forceinline__ void rref_apply_quad_inl(T (&nb)[(N + LN - 1) / LN][K], const T (&alpha)[K], T tol,
                                                T (&out)[(N + LN - 1) / LN], const int lq) {
    constexpr int S = (N + LN - 1) / LN;
    constexpr int BIG = 1 << 20;
    int col[S];
#pragma unroll
    for (int s = 0; s < S; ++s) col[s] = LN * s + lq;
    int jrow[K];                       // column at which row r became a pivot row (BIG: never)
    int j0 = 0;
    static_for<0, K>([&](auto tc) {
        constexpr int t = decltype(tc)::value;          // round t: rows < t are pivot rows, rows >= t are not
        // ---- next pivot column: first column >= j0 with max |entry| over the rows >= t above tol
        int cand = BIG;
#pragma unroll
        for (int s = S - 1; s >= 0; --s) {
            T cm = num<T>::abs(nb[s][t]);
#pragma unroll
            for (int r = t + 1; r < K; ++r) cm = num<T>::max(cm, num<T>::abs(nb[s][r]));
            const bool e = (cm > tol) && (col[s] >= j0);
            cand = e ? col[s] : cand;
        }
        int jmin = min(cand, dpp_mov<0xB1>(cand));
        if constexpr (LN == 4) jmin = min(jmin, dpp_mov<0x4E>(jmin));
        const bool found = jmin < BIG;
        // ---- that column's K entries, replicated over the quad (zeros when no column was found)
        T w[S];
#pragma unroll
        for (int s = 0; s < S; ++s) w[s] = (col[s] == jmin) ? T(1) : T(0);
        T f[K];
#pragma unroll
        for (int r = 0; r < K; ++r) {
            T v = w[0] * nb[0][r];
#pragma unroll
            for (int s = 1; s < S; ++s) v = num<T>::fma(w[s], nb[s][r], v);
            f[r] = qsum<LN>(v);
        }
        // ---- pivot row: first arg-max of |f| over rows t..K-1 (np.argmax semantics in the reference's row order)
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
#pragma unroll
        for (int s = 0; s < S; ++s) {
            T rowp = oh[t] * nb[s][t];
#pragma unroll
            for (int r = t + 1; r < K; ++r) rowp = num<T>::fma(oh[r], nb[s][r], rowp);
            const T oldt = nb[s][t];
#pragma unroll
            for (int r = t + 1; r < K; ++r) nb[s][r] = isp[r] ? oldt : nb[s][r];
            // scaled pivot row: zero left of the pivot column (those columns are frozen, see the header comment)
            const T pr = ((col[s] >= jmin) || !found) ? rowp * inv : T(0);
            nb[s][t] = pr;
#pragma unroll
            for (int r = 0; r < K; ++r)
                if (r != t) nb[s][r] = num<T>::fma(found ? -f[r] : T(0), pr, nb[s][r]);
        }
        jrow[t] = jmin;
        j0 = found ? jmin + 1 : j0;
    });
#pragma unroll
    for (int s = 0; s < S; ++s) {
        T v = T(0);
#pragma unroll
        for (int r = 0; r < K; ++r) v = num<T>::fma((jrow[r] <= col[s]) ? alpha[r] : T(0), nb[s][r], v);
        out[s] = v;
    }
}

// ---- entry points.  float (production): everything inlined into the one straight-line kernel.  double (parity
// build): the two big pieces are kept as real functions.  Reason: hipcc 7.2 miscompiles the fully inlined
// k_step<double, Iiwa, 4, false> (512 VGPRs + 0.9 KB scratch: garbage in mu at -O3 and -O1 alike, while the same source
// is exact as float, as double with HOLD = true, and as double with either piece outlined) -- found by
// tests/test_gpu_parity.py::test_refresh_and_exact_bias_variants_against_oracle.  Outlining keeps the double kernels
// far from the register ceiling; their speed is irrelevant.
template <typename T, int M, int N, int LN, typename AF, typename YF>
__device__ __attribute__((noinline)) void bidiag_solve_null_quad_out(AF& aget, YF& yget, T (&x)[(N + LN - 1) / LN],
                                                                     T (&nb)[(N + LN - 1) / LN][N - M], const int lq) {
    bidiag_solve_null_quad_inl<T, M, N, LN>(aget, yget, x, nb, lq);
}
template <typename T, int M, int N, int LN = 4, typename AF, typename YF>
__device__ __forceinline__ void bidiag_solve_null_quad(AF&& aget, YF&& yget, T (&x)[(N + LN - 1) / LN],
                                                       T (&nb)[(N + LN - 1) / LN][N - M], const int lq) {
    if constexpr (std::is_same<T, double>::value) bidiag_solve_null_quad_out<T, M, N, LN>(aget, yget, x, nb, lq);
    else bidiag_solve_null_quad_inl<T, M, N, LN>(aget, yget, x, nb, lq);
}
template <typename T, int N, int K, int LN>
__device__ __attribute__((noinline)) void rref_apply_quad_out(T (&nb)[(N + LN - 1) / LN][K], const T (&alpha)[K], T tol,
                                                              T (&out)[(N + LN - 1) / LN], const int lq) {
    rref_apply_quad_inl<T, N, K, LN>(nb, alpha, tol, out, lq);
}
template <typename T, int N, int K, int LN = 4>
__device__ __forceinline__ void rref_apply_quad(T (&nb)[(N + LN - 1) / LN][K], const T (&alpha)[K], T tol,
                                                T (&out)[(N + LN - 1) / LN], const int lq) {
    if constexpr (std::is_same<T, double>::value) rref_apply_quad_out<T, N, K, LN>(nb, alpha, tol, out, lq);
    else rref_apply_quad_inl<T, N, K, LN>(nb, alpha, tol, out, lq);
}

}  // namespace atacom
