// Quad-cooperative variant of the null-space solver: FOUR LANES PER ENVIRONMENT.
//
// Why (DESIGN.md "Kernel design", measured in profiles/): with one environment per lane the headline
// batch of 8192 environments is only 128 wavefronts -- 1/8 of the chip's 1024 SIMDs, each running one
// latency-bound dependent chain (~90 us per env step regardless of batch up to 65536).  Splitting every
// environment over the 4 lanes of a DPP quad quarters the chain length, uses 4x the SIMDs at the same
// batch, and costs only ~1.25x the total lane-work, so it is also competitive at large batch.
//
// Data distribution inside a quad (lq = lane & 3):
//   * matrices with N columns (J_c: M x N, null basis: N x K) are split BY COLUMN: column c lives in lane
//     c % 4, "slot" c / 4  (S = ceil(N / 4) slots per lane; slots past N hold zeros);
//   * vectors over the M rows (rhs y, the bidiagonal d / e, left reflectors u) are REPLICATED in the four
//     lanes and computed redundantly -- their values stay bitwise identical across the quad because every
//     cross-lane sum uses the same commutative butterfly;
//   * cross-lane traffic is DPP only (quad_perm): a broadcast is one v_mov_dpp, a quad sum two v_add_dpp;
//     no LDS, no ds_bpermute, no barriers.
// The arithmetic is the same Householder bidiagonalisation / rref chart as atacom_linalg.h (which remains
// the one-lane-per-env reference implementation); only the summation order inside dot products differs.
#pragma once
#include <type_traits>
#include "atacom_linalg.h"

namespace atacom {

// compile-time loop: the body is instantiated once per index, so every array index below is a constant
// regardless of what the loop unroller decides (the DPP intrinsics are `convergent`, which makes LLVM
// reluctant to fully unroll the big outer loops on its own).
template <int I, int END, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < END) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, END>(f);
    }
}

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

// value of lane O (0..3) of the quad, in all four lanes
template <int O, typename V>
__device__ __forceinline__ V qbcast(V v) { return dpp_mov<O * 0x55>(v); }
// sum over the quad, identical bits in all four lanes
template <typename T>
__device__ __forceinline__ T qsum(T v) {
    const T s1 = v + dpp_mov<0xB1>(v);     // quad_perm [1,0,3,2]
    return s1 + dpp_mov<0x4E>(s1);         // quad_perm [2,3,0,1]
}
// element 4*slot + lq of a replicated compile-time-indexed array (0 past its end)
template <typename T, int LEN>
__device__ __forceinline__ T pick4(const T (&z)[LEN], int slot4, int lq) {
    const T z0 = slot4 + 0 < LEN ? z[slot4 + 0 < LEN ? slot4 + 0 : 0] : T(0);
    const T z1 = slot4 + 1 < LEN ? z[slot4 + 1 < LEN ? slot4 + 1 : 0] : T(0);
    const T z2 = slot4 + 2 < LEN ? z[slot4 + 2 < LEN ? slot4 + 2 : 0] : T(0);
    const T z3 = slot4 + 3 < LEN ? z[slot4 + 3 < LEN ? slot4 + 3 : 0] : T(0);
    return lq == 0 ? z0 : (lq == 1 ? z1 : (lq == 2 ? z2 : z3));
}

// a: M x N split by column over the quad (a[r][slot] = A[r][4*slot+lq], zeros past N); y replicated.
// On return x[slot] and nb[slot][k] are the column-split  A^+ y  and orthonormal null basis (see
// bidiag_solve_null in atacom_linalg.h for the algorithm and its provenance).
template <typename T, int M, int N>
__device__ __forceinline__ void bidiag_solve_null_quad(T (&a)[M][(N + 3) / 4], T (&y)[M], T (&x)[(N + 3) / 4],
                                                       T (&nb)[(N + 3) / 4][N - M], const int lq) {
    constexpr int S = (N + 3) / 4, K = N - M;
    T d[M], e[M], taup[M];
    static_for<0, M>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int si = i / 4, li = i % 4;      // column i lives in slot si of lane li
        // ---- right reflector G(i) from row i, columns > i
        T part = (lq > li) ? a[i][si] * a[i][si] : T(0);
#pragma unroll
        for (int s = si + 1; s < S; ++s) part = num<T>::fma(a[i][s], a[i][s], part);
        const T ss = qsum(part);
        const T alpha = (li == 0) ? qbcast<0>(a[i][si]) : (li == 1) ? qbcast<1>(a[i][si])
                      : (li == 2) ? qbcast<2>(a[i][si]) : qbcast<3>(a[i][si]);
        T beta, tp;
        const T sc = larfg_scale(alpha, ss, beta, tp);
        d[i] = beta;
        taup[i] = tp;
        // store the FULL reflector vector in row i: 0 for c < i, 1 at c == i, v for c > i
        a[i][si] = (lq > li) ? a[i][si] * sc : ((lq == li) ? T(1) : T(0));
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (s > si) a[i][s] *= sc;
            if (s < si) a[i][s] = T(0);
        }
        if constexpr (i < M - 1) {
#pragma unroll
            for (int r = i + 1; r < M; ++r) {
                T wp = a[r][si] * a[i][si];
#pragma unroll
                for (int s = si + 1; s < S; ++s) wp = num<T>::fma(a[r][s], a[i][s], wp);
                const T w = qsum(wp) * tp;
#pragma unroll
                for (int s = si; s < S; ++s) a[r][s] = num<T>::fma(-w, a[i][s], a[r][s]);
            }
            // ---- left reflector H(i) from column i (slot si of lane li), rows i+1..M-1
            T sup0 = T(0), sup1 = T(0);                  // two accumulators: the chain is on the critical path
#pragma unroll
            for (int r = i + 2; r < M; ++r) {
                if ((r - i) & 1) sup1 = num<T>::fma(a[r][si], a[r][si], sup1);
                else sup0 = num<T>::fma(a[r][si], a[r][si], sup0);
            }
            const T sup = sup0 + sup1;
            T su, alq;
            T u[M];
            if (li == 0) { su = qbcast<0>(sup); alq = qbcast<0>(a[i + 1][si]); }
            else if (li == 1) { su = qbcast<1>(sup); alq = qbcast<1>(a[i + 1][si]); }
            else if (li == 2) { su = qbcast<2>(sup); alq = qbcast<2>(a[i + 1][si]); }
            else { su = qbcast<3>(sup); alq = qbcast<3>(a[i + 1][si]); }
            T betaq, tq;
            const T scq = larfg_scale(alq, su, betaq, tq);
            e[i] = betaq;
#pragma unroll
            for (int r = i + 2; r < M; ++r) {
                const T col = (li == 0) ? qbcast<0>(a[r][si]) : (li == 1) ? qbcast<1>(a[r][si])
                            : (li == 2) ? qbcast<2>(a[r][si]) : qbcast<3>(a[r][si]);
                u[r] = col * scq;
            }
#pragma unroll
            for (int s = si; s < S; ++s) {
                T w = a[i + 1][s], w1 = T(0);
#pragma unroll
                for (int r = i + 2; r < M; ++r) {
                    if ((r - i) & 1) w1 = num<T>::fma(u[r], a[r][s], w1);
                    else w = num<T>::fma(u[r], a[r][s], w);
                }
                w = (w + w1) * tq;
                if (s == si) w = (lq > li) ? w : T(0);          // columns <= i are not touched
                a[i + 1][s] -= w;
#pragma unroll
                for (int r = i + 2; r < M; ++r) a[r][s] = num<T>::fma(-w, u[r], a[r][s]);
            }
            {
                T w = y[i + 1];
#pragma unroll
                for (int r = i + 2; r < M; ++r) w = num<T>::fma(u[r], y[r], w);
                w *= tq;
                y[i + 1] -= w;
#pragma unroll
                for (int r = i + 2; r < M; ++r) y[r] = num<T>::fma(-w, u[r], y[r]);
            }
        }
    });
    // ---- z = B^{-1} Q^T y (replicated), then split by column
    T z[M];
    z[0] = num<T>::div(y[0], d[0]);
#pragma unroll
    for (int i = 1; i < M; ++i) z[i] = num<T>::div(num<T>::fma(-e[i - 1], z[i - 1], y[i]), d[i]);
#pragma unroll
    for (int s = 0; s < S; ++s) {
        x[s] = pick4<T, M>(z, 4 * s, lq);
#pragma unroll
        for (int k = 0; k < K; ++k) nb[s][k] = (4 * s + lq == M + k) ? T(1) : T(0);
    }
    // ---- [x | nb] <- G(1) ... G(M) [x | nb]
    static_for<0, M>([&](auto kc) {
        constexpr int i = M - 1 - decltype(kc)::value;
        constexpr int si = i / 4;
        {
            T wp = a[i][si] * x[si];
#pragma unroll
            for (int s = si + 1; s < S; ++s) wp = num<T>::fma(a[i][s], x[s], wp);
            const T w = qsum(wp) * taup[i];
#pragma unroll
            for (int s = si; s < S; ++s) x[s] = num<T>::fma(-w, a[i][s], x[s]);
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            T wp = a[i][si] * nb[si][k];
#pragma unroll
            for (int s = si + 1; s < S; ++s) wp = num<T>::fma(a[i][s], nb[s][k], wp);
            const T w = qsum(wp) * taup[i];
#pragma unroll
            for (int s = si; s < S; ++s) nb[s][k] = num<T>::fma(-w, a[i][s], nb[s][k]);
        }
    });
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
//   * rows are not physically swapped: row r remembers the column jrow[r] at which it became a pivot row and
//     the action component ar[r] = alpha[round] that the reference's row order pairs it with;
//   * the pivot-row scaling and the elimination are one fused update  row_r -= g_r * (row_p / pivot)  with
//     g_p = pivot - 1, which leaves no per-element select in the K x S inner loop.
// out[slot] = (Nc @ alpha)[4*slot + lq].
template <typename T, int N, int K>
__device__ __forceinline__ void rref_apply_quad(T (&nb)[(N + 3) / 4][K], const T (&alpha)[K], T tol,
                                                T (&out)[(N + 3) / 4], const int lq) {
    constexpr int S = (N + 3) / 4;
    constexpr int BIG = 1 << 20;
    int col[S];
#pragma unroll
    for (int s = 0; s < S; ++s) col[s] = 4 * s + lq;
    bool used[K];
    int jrow[K];
    T ar[K];
#pragma unroll
    for (int r = 0; r < K; ++r) { used[r] = false; jrow[r] = BIG; ar[r] = T(0); }
    int j0 = 0;
    static_for<0, K>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        // ---- next pivot column: first column >= j0 with max |entry| over the unused rows > tol
        int cand = BIG;
#pragma unroll
        for (int s = S - 1; s >= 0; --s) {
            T cm = T(0);
#pragma unroll
            for (int r = 0; r < K; ++r) cm = num<T>::max(cm, used[r] ? T(0) : num<T>::abs(nb[s][r]));
            const bool e = (cm > tol) && (col[s] >= j0);
            cand = e ? col[s] : cand;
        }
        int jmin = min(cand, dpp_mov<0xB1>(cand));
        jmin = min(jmin, dpp_mov<0x4E>(jmin));
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
            f[r] = qsum(v);
        }
        // ---- pivot row: first arg-max of |f| over the unused rows
        T p = T(-1);
#pragma unroll
        for (int r = 0; r < K; ++r) p = num<T>::max(p, used[r] ? T(-1) : num<T>::abs(f[r]));
        bool isp[K];
        bool taken = !found;
        T pj = T(1);
#pragma unroll
        for (int r = 0; r < K; ++r) {
            const bool m = !used[r] && (num<T>::abs(f[r]) == p);
            isp[r] = m && !taken;
            taken = taken || m;
            pj = isp[r] ? f[r] : pj;
        }
        const T inv = num<T>::rcp(pj);
        T oh[K], g[K];
#pragma unroll
        for (int r = 0; r < K; ++r) {
            oh[r] = isp[r] ? inv : T(0);
            g[r] = isp[r] ? f[r] - T(1) : f[r];
        }
        // ---- scaled pivot row (zero left of the pivot column and when nothing was found), fused update
#pragma unroll
        for (int s = 0; s < S; ++s) {
            T pr = oh[0] * nb[s][0];
#pragma unroll
            for (int r = 1; r < K; ++r) pr = num<T>::fma(oh[r], nb[s][r], pr);
            pr = (col[s] >= jmin) ? pr : T(0);
#pragma unroll
            for (int r = 0; r < K; ++r) nb[s][r] = num<T>::fma(-g[r], pr, nb[s][r]);
        }
#pragma unroll
        for (int r = 0; r < K; ++r) {
            used[r] = used[r] || isp[r];
            jrow[r] = isp[r] ? jmin : jrow[r];
            ar[r] = isp[r] ? alpha[t] : ar[r];
        }
        j0 = jmin + 1;
    });
#pragma unroll
    for (int s = 0; s < S; ++s) {
        T v = T(0);
#pragma unroll
        for (int r = 0; r < K; ++r) v = num<T>::fma((jrow[r] <= col[s]) ? ar[r] : T(0), nb[s][r], v);
        out[s] = v;
    }
}

}  // namespace atacom
