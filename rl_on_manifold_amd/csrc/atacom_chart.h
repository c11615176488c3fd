// The opt-in canonical chart (cfg.chart_mode = 1; SURVEY.md section 7.3 H1 "null_mode = exact"): one sub-step's
//     mu = -Jc^+ (psi + Kc c) + N alpha
// through the slack structure of  Jc = [[a, 0], [A, diag(s)]]  instead of a factorisation of Jc.  Specification, derivation
// and the relation to the reference's rref(tol) chart (atacom/atacom.py:123-133, null_space_coordinate.py:8-79):
// oracle/canonical_chart.py -- this file is that recursion, decision by decision, in square-root form (see canonical_mu),
// for one environment per lane.
//
//   M = I + sum_soft A_g^T A_g / s_g^2,  Gamma = M^-1   (dim_q x dim_q; the Gram matrix of the coordinate functionals on
//                                                        the null space once every row has been imposed)
//   rows that cannot go into M -- the equality row (s = 0) and stiff rows |s_g| < theta max|A_g| -- by one exact rank-one
//   conditioning step each (the first stiff row keeps its slack velocity as a coordinate of the state: exact for every
//   s >= 0);  x = minimum-norm solution
//   chart: Cholesky of Gamma in joint order that skips a joint whose current diagonal ||P_S e_j||^2 <= tol^2; missing
//   free coordinates go to the first slack columns that pass; N alpha falls out of the same recursion
//   w_g = -(y_g + A_g u) / s_g with the true slack.
// Cost for the iiwa task (12 x 17): 15.9 kFLOP per env-step against 52.2 k for the LAPACK-basis chart (bench.py:
// algorithmic_flops_canonical); measured 8.6 k vector instructions per wave and step against 11.8 k.
// The plain path is straight-line code on compile-time indices with selects for the data-dependent choices; the rare
// parts sit behind wave-uniform ballots -- the stiff rows and the slack stages -- and work on PER-LANE ROWS (below).
#pragma once
#include "atacom_envs.h"
#include "atacom_linalg.h"
#include "atacom_quad.h"

namespace atacom {

// tuning build only (profiles/tools/gpu_phase_probe.py): how often a wavefront went through the data-dependent parts of the chart --
// [0] trips of the stiff-row loop, [1] trips of slack stage A, [2] slack stage B
#ifdef ATACOM_TIMESTAMPS
#define ATACOM_DBG_PARAM , int (&dbg)[3]
#define ATACOM_DBG_ARG(d) , d
#define ATACOM_DBG_COUNT(i) dbg[i] += 1
#else
#define ATACOM_DBG_PARAM
#define ATACOM_DBG_ARG(d)
#define ATACOM_DBG_COUNT(i)
#endif

// Group kernels: the constraint ROWS owned by the lanes of a group in slack stage B and in the assembly (1), or every lane
// working through all of them (0: the first form of round 3, kept for A/B builds: -DATACOM_CHART_ROWDIST=0).
#ifndef ATACOM_CHART_ROWDIST
#define ATACOM_CHART_ROWDIST 1
#endif
// Group kernels, slack stage A (two or more free coordinates missing): the first-fit scan over the slack columns in static row
// order with a wave-uniform early-out (STATIC_A; compile-time indices, structural zeros, no row gather: 80 instead of 190
// instructions per candidate), or per lane -- trip n = every environment's n-th untaken column, gathered by one-hot blends.
// Measured (iiwa, 8192 constraint-active environments, one box, profiles/r03_ab_stage_a.log): static order shortens the
// SLOWEST wavefronts -- atacom_step 35.7 -> 32.2 us (4 lanes), 35.8 -> 31.0 (8 lanes) -- but lengthens the average one (its
// eleven unrolled row bodies cost the plain path registers): T-step kernels 19.6 -> 20.4 us per step, the 120-step packed
// collection 2.14 -> 2.36 ms.  So k_step takes the static form and the T-step kernels keep the per-lane one (0 here: per lane
// everywhere, for A/B builds).
#ifndef ATACOM_CHART_STAGEA_STATIC
#define ATACOM_CHART_STAGEA_STATIC 1
#endif
// Group kernels: 3 = the third form (atacom_chart_group.h, round 4: own columns, metric by columns, no inverse factor, row
// slots always, all candidate rows of slack stage A at once); 2 = the second form below (canonical_mu_group), kept for A/B
// builds (-DATACOM_CHART_FORM=2).
#ifndef ATACOM_CHART_FORM
#define ATACOM_CHART_FORM 3
#endif

template <typename T> struct chart_const {
    static constexpr T THETA = T(3e-2);     // stiff-row threshold (oracle/canonical_chart.py: THETA)
    static constexpr T TINY = T(1e-6);      // below this (relative) slack a row is an equality: w_g -> 0
    static constexpr T FLOOR = std::is_same<T, float>::value ? T(1e-30) : T(1e-300);
    static constexpr T REL = std::is_same<T, float>::value ? T(64 * 1.1920928955078125e-07) : T(64 * 2.220446049250313e-16);
};

// symmetric NQ x NQ matrix, lower triangle stored: sym(i, j) = sym(j, i)
template <typename T, int N>
struct SymMat {
    T v[N * (N + 1) / 2];
    __device__ __forceinline__ T& operator()(int i, int j) { return (i >= j) ? v[i * (i + 1) / 2 + j] : v[j * (j + 1) / 2 + i]; }
    __device__ __forceinline__ const T& operator()(int i, int j) const {
        return (i >= j) ? v[i * (i + 1) / 2 + j] : v[j * (j + 1) / 2 + i];
    }
};

// L^-1 of the Cholesky factor of M (M = L L^T, lower triangle of M given): Gamma = M^-1 = Li^T Li.
template <typename T, int N>
__device__ __forceinline__ void chol_inverse_factor(const SymMat<T, N>& M, T (&Li)[N][N]) {
    T L[N][N], inv[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        T d = M(j, j);
#pragma unroll
        for (int k = 0; k < j; ++k) d = num<T>::fma(-L[j][k], L[j][k], d);
        d = num<T>::max(d, chart_const<T>::FLOOR);
        const T r = num<T>::rcp(num<T>::sqrt(d));
        inv[j] = r;
        L[j][j] = d * r;
#pragma unroll
        for (int i = j + 1; i < N; ++i) {
            T a = M(i, j);
#pragma unroll
            for (int k = 0; k < j; ++k) a = num<T>::fma(-L[i][k], L[j][k], a);
            L[i][j] = a * r;
        }
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
        Li[j][j] = inv[j];
#pragma unroll
        for (int i = j + 1; i < N; ++i) {
            T a = T(0);
#pragma unroll
            for (int k = j; k < i; ++k) a = num<T>::fma(L[i][k], Li[k][j], a);
            Li[i][j] = -a * inv[i];
        }
    }
}

// A row of A as a conditioning step sees it: row R known at compile time (its structural zeros cost nothing) ...
template <typename T, typename E, int R>
struct StaticRow {
    const T (&A)[E::NC][E::NQ];
    static constexpr bool zero(int i) { return E::jac_zero(R, i); }
    __device__ __forceinline__ T operator()(int i) const { return A[R][i]; }
};
// ... or the lane's OWN row: every lane of the wavefront conditions on a different row in the same instructions
template <typename T, typename E>
struct LaneRow {
    T v[E::NQ];
    static constexpr bool zero(int) { return false; }
    __device__ __forceinline__ T operator()(int i) const { return v[i]; }
};
// Row (NF + g) of A for the g whose bit is `low` (at most one bit; none: the zero row), and entry g of a per-row family:
// one-hot blends -- exact (1 * a + 0 * ...), and never a switch (see the note on alpha in canonical_mu).
template <typename T, typename E>
__device__ __forceinline__ void lane_row(const T (&A)[E::NC][E::NQ], const unsigned low, LaneRow<T, E>& row) {
#pragma unroll
    for (int i = 0; i < E::NQ; ++i) row.v[i] = T(0);
#pragma unroll
    for (int g = 0; g < E::NG; ++g) {
        const T h = (low == (1u << g)) ? T(1) : T(0);
#pragma unroll
        for (int i = 0; i < E::NQ; ++i)
            if (!E::jac_zero(E::NF + g, i)) row.v[i] = num<T>::fma(h, A[E::NF + g][i], row.v[i]);
    }
}
template <typename T, int NG, typename Z>
__device__ __forceinline__ T lane_pick(Z&& z, const unsigned low) {
    T v = T(0);
#pragma unroll
    for (int g = 0; g < NG; ++g) v = num<T>::fma((low == (1u << g)) ? T(1) : T(0), z(g), v);
    return v;
}
// PER-LANE ROWS.  The data-dependent parts of the chart -- conditioning on the stiff rows, the scan of the slack columns
// for a missing free coordinate -- touch ONE row per environment, a different one in each.  Written per row (static_for
// over the rows, each under a wave-uniform ballot) a wavefront runs one step per DISTINCT row among its environments; per
// lane (trip n: every lane works on ITS n-th row, gathered by lane_row) it is one trip, rarely two.  Measured on
// constraint-active iiwa states (profiles/tools/gpu_phase_probe.py, path counters of the tuning build; 8192 environments, 4 lanes):
// a trip of the per-row slack scan cost 2.6 us (all 11 columns evaluated for every lane) and the slowest wavefronts of a
// launch -- four trips -- set its duration, 33 us against 15 us on quiet states.
template <typename T, typename E, typename F>
__device__ __forceinline__ int for_each_stiff_row(const T (&A)[E::NC][E::NQ], const T (&s)[E::NG], const T (&y)[E::NC],
                                                  const bool (&soft)[E::NG], F&& cond) {
    constexpr int NF = E::NF, NG = E::NG;
    unsigned todo = 0u;
#pragma unroll
    for (int g = 0; g < NG; ++g) todo |= soft[g] ? 0u : (1u << g);
    bool first = true;
    int trips = 0;
#pragma unroll 1
    while (__builtin_amdgcn_ballot_w64(todo != 0u) != 0ull) {
        ++trips;
        const bool on = todo != 0u;
        const unsigned low = todo & (0u - todo);                    // the lane's next stiff row (0: it has none left)
        todo ^= low;
        LaneRow<T, E> row;
        lane_row<T, E>(A, low, row);
        const T sg = lane_pick<T, NG>([&](int g) { return s[g]; }, low);
        const T yg = lane_pick<T, NG>([&](int g) { return y[NF + g]; }, low);
        const bool prim = first && on;                              // the first stiff row: its slack is a coordinate
        first = false;
        cond(row, prim ? sg : T(0), prim ? T(0) : sg * sg, yg, on);
    }
    return trips;
}

// A (NC x NQ) = K J with the equality row (if any) first; arow[g] = max_c |A[NF + g][c]|; s: slacks; y = psi + Kc c;
// alpha: the NK null coordinates.  mu (NN) = [joint accelerations | slack velocities].
//
// State of the recursion: x = (u, w_p), N1 = NQ + 1 -- w_p is the slack velocity of the FIRST stiff row p of the
// environment, carried as a coordinate of its own (exact for every s_p >= 0, no division by s_p); inert (zero prior
// variance) when the environment has no stiff row.
//
// SQUARE-ROOT FORM.  oracle/canonical_chart.py carries Gamma itself and takes the chart's pivots off its diagonal after
// rank-one downdates; in float32 that loses them: a pivot near the tolerance, tol^2 = 2.5e-3, is what is left of O(1)
// entries, so its relative error is eps / tol^2 and the coordinates of N alpha inherit it (measured on the iiwa task: 1 %
// of the sub-steps off by > 3e-3 of max|mu|).  Here Gamma = R R^T is never formed: the recursion keeps the vectors
// v_i = R^T e_i (one per coordinate of x; Gamma_ij = v_i . v_j), a functional p x becomes the vector w = sum_i p_i v_i, a
// pivot is a SQUARED NORM |v_j|^2 (relative error eps / tol), and conditioning is the projection v_i -= w (w . v_i) / S
// (Potter's square-root update when the row carries noise).  Same recursion, same decisions, in exact arithmetic the same
// numbers; float32 vs the float64 specification: median 1.5e-7, 99.9 % below 1e-5 (profiles/r03_chart_float32.md).
// Layout VT[k][i] = component k of v_i: both inner loops (g_i = v_i . w over k; v_i -= c g_i w) run over i with k fixed,
// i.e. they are axpy-shaped and pack into v_pk_fma_f32 pairs.
template <typename T, typename E>
__device__ __forceinline__ void canonical_mu(const T (&A)[E::NC][E::NQ], const T (&arow)[E::NG], const T (&s)[E::NG],
                                             const T (&y)[E::NC], const T (&alpha)[E::NK], const T tol, T (&mu)[E::NN] ATACOM_DBG_PARAM) {
    constexpr int NQ = E::NQ, NF = E::NF, NG = E::NG, NK = NQ - NF, N1 = NQ + 1;
    static_assert(NF <= 1, "at most one equality row");
    using CC = chart_const<T>;
    const T tol2 = tol * tol;
    // ---- metric of the soft rows
    bool soft[NG], isp[NG];
    bool has_stiff = false;
    T VT[N1][N1];                       // VT[k][i]: component k of v_i (i < NQ: joint i; i = NQ: the coordinate slack)
    T x[N1];
    {
        SymMat<T, NQ> M;
        T b[NQ];
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            b[i] = T(0);
#pragma unroll
            for (int j = 0; j <= i; ++j) M(i, j) = (i == j) ? T(1) : T(0);
        }
        bool has_p = false;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int r = NF + g;
            soft[g] = num<T>::abs(s[g]) >= CC::THETA * arow[g];
            isp[g] = !soft[g] && !has_p;                            // the first stiff row: its slack is a coordinate
            has_p = has_p || !soft[g];
            const T om = soft[g] ? num<T>::rcp(s[g] * s[g]) : T(0);
            T wa[NQ];
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                if (E::jac_zero(r, i)) continue;
                wa[i] = om * A[r][i];
                b[i] = num<T>::fma(wa[i], y[r], b[i]);
#pragma unroll
                for (int j = 0; j <= i; ++j)
                    if (!E::jac_zero(r, j)) M(i, j) = num<T>::fma(wa[i], A[r][j], M(i, j));
            }
        }
        has_stiff = has_p;
        T Li[NQ][NQ];
        chol_inverse_factor<T, NQ>(M, Li);
        // Gamma = Li^T Li: v_i = column i of Li (zeros above the diagonal);  x = -Gamma b = -Li^T (Li b)
        T z[NQ];
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            T a = T(0);
#pragma unroll
            for (int j = 0; j <= k; ++j) a = num<T>::fma(Li[k][j], b[j], a);
            z[k] = a;
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            T a = T(0);
#pragma unroll
            for (int k = i; k < NQ; ++k) a = num<T>::fma(Li[k][i], z[k], a);
            x[i] = -a;
        }
        x[NQ] = T(0);
#pragma unroll
        for (int k = 0; k < N1; ++k)
#pragma unroll
            for (int i = 0; i < N1; ++i)
                VT[k][i] = (k < NQ && i < NQ) ? ((k >= i) ? Li[k < NQ ? k : 0][i < NQ ? i : 0] : T(0))
                                               : ((k == NQ && i == NQ && has_p) ? T(1) : T(0));
    }
    // the vector of a functional  f(x) = sum_i p_i x_i + cw x_w  and its products with every v_i
    auto project = [&](const T (&w)[N1], const T cproj, T (&g)[N1]) {
        // g_i = v_i . w;  v_i -= cproj g_i w
#pragma unroll
        for (int i = 0; i < N1; ++i) g[i] = T(0);
#pragma unroll
        for (int k = 0; k < N1; ++k)
#pragma unroll
            for (int i = 0; i < N1; ++i) g[i] = num<T>::fma(VT[k][i], w[k], g[i]);
#pragma unroll
        for (int k = 0; k < N1; ++k) {
            const T wk = w[k] * cproj;
#pragma unroll
            for (int i = 0; i < N1; ++i) VT[k][i] = num<T>::fma(-g[i], wk, VT[k][i]);
        }
    };
    // exact rank-one conditioning of the state on  (A_r, cw) . x + (noise of variance s2) = -yr
    auto condition = [&](const auto& row, const T cw, const T s2, const T yr, const bool on) {
        using R = std::decay_t<decltype(row)>;
        T w[N1], g[N1];
        T ww = T(0), e = -yr, nrm = num<T>::fma(cw, cw, s2);
#pragma unroll
        for (int k = 0; k < N1; ++k) {
            T a = VT[k][NQ] * cw;
#pragma unroll
            for (int i = 0; i < NQ; ++i)
                if (!R::zero(i)) a = num<T>::fma(VT[k][i], row(i), a);
            w[k] = a;
            ww = num<T>::fma(a, a, ww);
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            if (R::zero(i)) continue;
            e = num<T>::fma(-row(i), x[i], e);
            nrm = num<T>::fma(row(i), row(i), nrm);
        }
        e = num<T>::fma(-cw, x[NQ], e);
        const T S = s2 + ww;
        // Gamma <= I: a row that has nothing left to say (vanishing, or dependent on rows imposed before) is dropped
        const bool ok = on && (S > CC::REL * nrm);
        const T iS = ok ? num<T>::rcp(S) : T(0);
        // Potter: (I - c w w^T)^2 = I - w w^T / S  for  c = (1 / S) / (1 + sqrt(s2 / S))   (s2 = 0: the projection)
        const T c = iS * num<T>::rcp(T(1) + num<T>::sqrt(s2 * iS));
        project(w, c, g);
        const T ce = e * iS;
#pragma unroll
        for (int i = 0; i < N1; ++i) x[i] = num<T>::fma(g[i], ce, x[i]);
    };
    if constexpr (NF == 1) condition(StaticRow<T, E, 0>{A}, T(0), T(0), y[0], true);
    if (__builtin_amdgcn_ballot_w64(has_stiff) != 0ull) {
        [[maybe_unused]] const int trips = for_each_stiff_row<T, E>(A, s, y, soft, condition);
#ifdef ATACOM_TIMESTAMPS
        dbg[0] += trips;
#endif
    }
    // ---- the chart: conditioning recursion over the joints with the skip rule
    T U[N1];
#pragma unroll
    for (int i = 0; i < N1; ++i) U[i] = T(0);
    int n_acc = 0;
    static_for<0, NQ>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        // (joints NK ... NQ - 1 are looked at only while a free coordinate is missing: wave-uniform skip)
        if (j >= NK && __builtin_amdgcn_ballot_w64(n_acc < NK) == 0ull) return;
        T w[N1], g[N1];
        T dj = T(0);
#pragma unroll
        for (int k = 0; k < N1; ++k) { w[k] = VT[k][j]; dj = num<T>::fma(w[k], w[k], dj); }
        const bool acc = (n_acc < NK) && (dj > tol2);
        // alpha[n_acc] as a one-hot blend: written as a select chain the optimiser turns it into a dynamically indexed
        // private array, promotes that to LDS, and -- to address it -- reads the workgroup size from the AQL dispatch packet
        // in HOST memory: 2 us per wave alone, up to 13 us with a full launch queueing for it (profiles/tools/gpu_phase_probe.py)
        T tv = T(0);
#pragma unroll
        for (int i = 0; i < NK; ++i) tv = num<T>::fma((n_acc == i) ? T(1) : T(0), alpha[i], tv);
        const T inv = acc ? num<T>::rcp(dj) : T(0);
        const T coef = (tv - U[j]) * inv;
        project(w, inv, g);
#pragma unroll
        for (int i = 0; i < N1; ++i) U[i] = num<T>::fma(g[i], coef, U[i]);
        n_acc += acc ? 1 : 0;
    });
    // ---- free coordinates still missing after the joints: slack columns, in column order, the first one that passes.
    // The functional of slack column g on the extended state: f_g(x) = A_g u (then w_g = -f_g / s_g and
    // ||P_S e_col||^2 = |vector of f_g|^2 / s_g^2), except for the coordinate slack p: f_p(x) = w_p itself.
    // (flags are recomputed where they are cheap: every bool array costs 2 SGPRs per row, and the masks spill)
    bool sel[NG];
    T wtgt[NG];
    auto tiny = [&](int g) -> bool { return !isp[g] && (num<T>::abs(s[g]) < CC::TINY * arow[g]); };
#pragma unroll
    for (int g = 0; g < NG; ++g) { sel[g] = false; wtgt[g] = T(0); }
    bool done = false;
    // (A) more than one missing (0.1 % of the iiwa sub-steps): the general step, any rank
    if constexpr (NK >= 2) {
#pragma unroll 1
        for (int it = 0; it < NK - 1; ++it) {
            const bool want = (n_acc < NK - 1) && !done;
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(want) == 0ull, 1)) break;
            ATACOM_DBG_COUNT(1);
            T tv = T(0);                            // one-hot blend, see the joint recursion
#pragma unroll
            for (int i = 0; i < NK; ++i) tv = num<T>::fma((n_acc == i) ? T(1) : T(0), alpha[i], tv);
            T wsel[N1], vsel = T(0), rsel = T(0);
#pragma unroll
            for (int k = 0; k < N1; ++k) wsel[k] = T(0);
            bool any = false, tnsel = false;
            // (per row, not per lane as in canonical_mu_group: with 64 environments per wavefront the first-fit scan ran
            // as many trips as its slowest lane -- measured slower, 41.8 against 36.3 us per step at 8192 environments)
            static_for<0, NG>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                constexpr int r = NF + g;
                T wg[N1], v = T(0), fu = T(0);
#pragma unroll
                for (int k = 0; k < N1; ++k) {
                    T a = T(0);
#pragma unroll
                    for (int i = 0; i < NQ; ++i)
                        if (!E::jac_zero(r, i)) a = num<T>::fma(VT[k][i], A[r][i], a);
                    wg[k] = isp[g] ? VT[k][NQ] : a;
                    v = num<T>::fma(wg[k], wg[k], v);
                }
#pragma unroll
                for (int i = 0; i < NQ; ++i)
                    if (!E::jac_zero(r, i)) fu = num<T>::fma(A[r][i], U[i], fu);
                fu = isp[g] ? U[NQ] : fu;
                const T thr = tol2 * (isp[g] ? T(1) : s[g] * s[g]);
                const bool pass = want && !sel[g] && (tiny(g) || (v > thr));
                const bool take = pass && !any;
                any = any || pass;
#pragma unroll
                for (int k = 0; k < N1; ++k) wsel[k] = take ? wg[k] : wsel[k];
                vsel = take ? v : vsel;
                // slack g: f_g(x) = -s_g target;  coordinate slack p: f_p(x) = +target
                rsel = take ? (isp[g] ? fu - tv : num<T>::fma(s[g], tv, fu)) : rsel;
                tnsel = take ? tiny(g) : tnsel;
                wtgt[g] = take ? tv : wtgt[g];
                sel[g] = sel[g] || take;
            });
            done = done || (want && !any);
            const bool live = any && (vsel > T(0)) && !tnsel;
            const T iv = live ? num<T>::rcp(vsel) : T(0);
            T g[N1];
            project(wsel, iv, g);
            const T coef = rsel * iv;
#pragma unroll
            for (int i = 0; i < N1; ++i) U[i] = num<T>::fma(-g[i], coef, U[i]);
            n_acc += any ? 1 : 0;
        }
    }
    // (B) exactly one missing: S is one-dimensional, every v_i = beta_i dhat -- a scalar test per row
    const bool need1 = (n_acc == NK - 1) && !done;
    const T tv_last = alpha[NK - 1];
    if (__builtin_amdgcn_ballot_w64(need1) != 0ull) {
        ATACOM_DBG_COUNT(2);
        T nrm2[N1];
#pragma unroll
        for (int i = 0; i < N1; ++i) nrm2[i] = T(0);
#pragma unroll
        for (int k = 0; k < N1; ++k)
#pragma unroll
            for (int i = 0; i < N1; ++i) nrm2[i] = num<T>::fma(VT[k][i], VT[k][i], nrm2[i]);
        T dh[N1], sig = nrm2[0];                                    // the longest v_i: the best-conditioned direction
#pragma unroll
        for (int k = 0; k < N1; ++k) dh[k] = VT[k][0];
#pragma unroll
        for (int j = 1; j < N1; ++j) {
            const bool better = nrm2[j] > sig;                      // first maximum, like np.argmax
            sig = better ? nrm2[j] : sig;
#pragma unroll
            for (int k = 0; k < N1; ++k) dh[k] = better ? VT[k][j] : dh[k];
        }
        const T isd = (sig > T(0)) ? num<T>::rcp(num<T>::sqrt(sig)) : T(0);
        T beta[N1];                                                 // Gamma = beta beta^T
#pragma unroll
        for (int i = 0; i < N1; ++i) beta[i] = T(0);
#pragma unroll
        for (int k = 0; k < N1; ++k) {
            const T dk = dh[k] * isd;
#pragma unroll
            for (int i = 0; i < N1; ++i) beta[i] = num<T>::fma(VT[k][i], dk, beta[i]);
        }
        T fd[NG], res[NG], val[NG];
        bool pick[NG];
        bool any = false;
        T vbest = T(-1);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int r = NF + g;
            T a = T(0), fu = T(0);
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                if (E::jac_zero(r, i)) continue;
                a = num<T>::fma(A[r][i], beta[i], a);
                fu = num<T>::fma(A[r][i], U[i], fu);
            }
            fd[g] = isp[g] ? beta[NQ] : a;
            res[g] = isp[g] ? U[NQ] - tv_last : num<T>::fma(s[g], tv_last, fu);
            val[g] = fd[g] * fd[g];                                 // = f_g Gamma f_g^T
            const T thr = tol2 * (isp[g] ? T(1) : s[g] * s[g]);
            const bool pass = !sel[g] && (tiny(g) || (val[g] > thr));
            pick[g] = need1 && pass && !any;
            any = any || pass;
            vbest = sel[g] ? vbest : num<T>::max(vbest, val[g]);
        }
        // nothing passed (a numerically rank-deficient remainder): the untaken column with the largest projection
        bool taken = false;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const bool fb = need1 && !any && !taken && !sel[g] && (val[g] == vbest);
            pick[g] = pick[g] || fb;
            taken = taken || fb;
        }
        T fds = T(0), rs = T(0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            fds = pick[g] ? (tiny(g) ? T(0) : fd[g]) : fds; rs = pick[g] ? res[g] : rs;
            wtgt[g] = pick[g] ? tv_last : wtgt[g];
            sel[g] = sel[g] || pick[g];
        }
        // x -= Gamma f^T (f x - target value) / (f Gamma f^T)  =  beta (...) / (f beta)
        const T coef = (need1 && fds != T(0)) ? num<T>::div(rs, fds) : T(0);
#pragma unroll
        for (int i = 0; i < N1; ++i) U[i] = num<T>::fma(-beta[i], coef, U[i]);
    }
    // ---- assembly.  The equality row once more, exactly (rounding only): a u = -y_0, a U = 0
    if constexpr (NF == 1) {
        T aa = T(0), au = y[0], aU = T(0);
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            if (E::jac_zero(0, i)) continue;
            aa = num<T>::fma(A[0][i], A[0][i], aa);
            au = num<T>::fma(A[0][i], x[i], au);
            aU = num<T>::fma(A[0][i], U[i], aU);
        }
        const T iaa = (aa > T(0)) ? num<T>::rcp(aa) : T(0);
        au *= iaa; aU *= iaa;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            if (E::jac_zero(0, i)) continue;
            x[i] = num<T>::fma(-A[0][i], au, x[i]);
            U[i] = num<T>::fma(-A[0][i], aU, U[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) mu[i] = x[i] + U[i];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int r = NF + g;
        T wm = y[r], wa = T(0);
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            if (E::jac_zero(r, i)) continue;
            wm = num<T>::fma(A[r][i], x[i], wm);
            wa = num<T>::fma(A[r][i], U[i], wa);
        }
        const T inv_s = (num<T>::abs(s[g]) >= CC::TINY * arow[g]) ? num<T>::rcp(s[g]) : T(0);
        // a free slack coordinate takes its target itself; the coordinate slack is a component of the state
        const T w = sel[g] ? num<T>::fma(-wm, inv_s, wtgt[g]) : -(wm + wa) * inv_s;
        mu[NQ + g] = isp[g] ? x[NQ] + U[NQ] : w;
    }
}


// ------------------------------------------------------------------ the same recursion, LG lanes per environment
// The vectors are DISTRIBUTED over the lanes of the group: v_i lives in lane i % LG, slot i / LG (S = ceil(N1 / LG) slots;
// 8 lanes: one vector per lane for the iiwa task), together with "its" coordinates x_i, U_i.  A projection step is then
// local work on the own vector(s): g_i = v_i . w and v_i -= c g_i w -- 2 N1 multiply-adds per slot instead of 2 N1^2 --
// once w is known to every lane: a broadcast from the owner when w is one of the vectors (the joint steps), a group sum
// of the lanes' contributions when it is a combination (a constraint row).  The small dense prologue (metric, Cholesky
// factor) and the final assembly are replicated, like everything else outside the solver in the group kernels; every
// decision is taken on replicated or group-summed values, so the lanes of a group agree bit for bit.
// Cross-lane traffic: DPP broadcasts / butterfly sums (atacom_quad.h) and, once per slack stage, one ds_bpermute gather of
// the longest vector (its owner is data dependent).
// ROW SLOTS (second form of round 3): what the inequality rows contribute one by one -- the NG column tests of slack stage
// B, the NG slack velocities of the assembly -- is done by the lane that OWNS the row (row g: lane g % LG, slot g / LG),
// ceil(NG / LG) slots per lane instead of NG rows; the passing columns travel as ONE group-summed number (bit g = column g,
// disjoint powers of two: exact), first fit is its lowest set bit.  Same arithmetic on the same operands (the blended copies
// carry exact zeros where the row has structural ones), so the results are bit for bit those of the replicated form
// (-DATACOM_CHART_ROWDIST=0 builds it for A/B); measured on one box, 8192 environments, T-step kernels: iiwa 8 lanes 21.1 ->
// 19.8 us per step, 4 lanes 22.0 -> 21.5, 2 lanes 24.9 -> 23.7; planar 4 lanes 9.0 -> 8.1 (profiles/r03_ab_rowslots.log).
template <typename T>
__device__ __forceinline__ T lane_gather(T v, int src_lane) {
    if constexpr (sizeof(T) == 4) {
        return __builtin_bit_cast(T, __builtin_amdgcn_ds_bpermute(src_lane << 2, __builtin_bit_cast(int, v)));
    } else {
        const long long b = __builtin_bit_cast(long long, v);
        const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(b & 0xffffffffll));
        const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(b >> 32));
        return __builtin_bit_cast(T, ((long long)hi << 32) | (unsigned int)lo);
    }
}

// max over the group, in all its lanes (the butterfly of qsum, atacom_quad.h)
template <int LN, typename T>
__device__ __forceinline__ T qmax(T v) {
    const T s1 = num<T>::max(v, dpp_mov<0xB1>(v));
    if constexpr (LN == 2) return s1;
    else {
        const T s2 = num<T>::max(s1, dpp_mov<0x4E>(s1));
        if constexpr (LN == 4) return s2;
        else return num<T>::max(s2, dpp_mov<DPP_ROW_HALF_MIRROR>(s2));
    }
}

template <typename T, typename E, int LG, bool STATIC_A = false>
__device__ __forceinline__ void canonical_mu_group(const T (&A)[E::NC][E::NQ], const T (&arow)[E::NG], const T (&s)[E::NG],
                                                   const T (&y)[E::NC], const T (&alpha)[E::NK], const T tol, T (&mu)[E::NN],
                                                   const int lq ATACOM_DBG_PARAM) {
    constexpr int NQ = E::NQ, NF = E::NF, NG = E::NG, NK = NQ - NF, N1 = NQ + 1;
    constexpr int S = (N1 + LG - 1) / LG;
    static_assert(NF <= 1, "at most one equality row");
    using CC = chart_const<T>;
    const T tol2 = tol * tol;
    T oh[LG];                                     // one-hot of the lane's position in its group (blend, never a switch)
#pragma unroll
    for (int l = 0; l < LG; ++l) oh[l] = (lq == l) ? T(1) : T(0);
    // element (LG sl + lq) of a replicated, compile-time indexed family z(i), i < N1 (0 past the end)
    auto own = [&](auto&& z, int sl) -> T {
        T v = T(0);
#pragma unroll
        for (int l = 0; l < LG; ++l)
            if (LG * sl + l < N1) v = num<T>::fma(oh[l], z(LG * sl + l), v);
        return v;
    };
    // ROW SLOTS (ATACOM_CHART_ROWDIST).  Inequality row g belongs to lane g % LG, slot g / LG: what the rows contribute one
    // by one -- the eleven tests of slack stage B, the slack velocities of the assembly -- a lane does for ITS rows only
    // (RS = ceil(NG / LG) slots instead of NG rows), on copies of the row data gathered by one-hot blends; A is held over
    // the sub-steps of a step (hold_q), so its blend leaves the sub-step loop.  What the group needs to agree on travels
    // as group sums of values only one lane contributes to (exact), so the lanes still agree bit for bit.
    constexpr bool ROWDIST = (ATACOM_CHART_ROWDIST != 0) && (NG > LG);
    constexpr int RS = (NG + LG - 1) / LG;
    // entry (row slot t, joint i) is structurally zero for every lane of the group
    auto slot_zero = [](int t, int i) constexpr -> bool {
        bool z = true;
        for (int l = 0; l < LG; ++l)
            if (LG * t + l < NG) z = z && E::jac_zero(NF + LG * t + l, i);
        return z;
    };
    auto ownrow = [&](auto&& z, int t) -> T {               // z(g) of the lane's row in slot t (0: the lane has none)
        T v = T(0);
#pragma unroll
        for (int l = 0; l < LG; ++l)
            if (LG * t + l < NG) v = num<T>::fma(oh[l], z(LG * t + l), v);
        return v;
    };
    auto ownflag = [&](auto&& z, int t) -> bool {
        bool v = false;
#pragma unroll
        for (int l = 0; l < LG; ++l)
            if (LG * t + l < NG) v = v || ((lq == l) && z(LG * t + l));
        return v;
    };
    [[maybe_unused]] T Ao[RS][NQ], so[RS], ao[RS], yo[RS], bito[RS];
    if constexpr (ROWDIST && !STATIC_A) {            // T-step kernels: up front ("WHEN the row slots are built", below)
        T pw = T(0);                                         // 2^lq
#pragma unroll
        for (int l = 0; l < LG; ++l) pw = num<T>::fma(oh[l], T(1u << l), pw);
#pragma unroll
        for (int t = 0; t < RS; ++t) {
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                T v = T(0);
#pragma unroll
                for (int l = 0; l < LG; ++l)
                    if (LG * t + l < NG && !E::jac_zero(NF + LG * t + l, i)) v = num<T>::fma(oh[l], A[NF + LG * t + l][i], v);
                Ao[t][i] = v;
            }
            so[t] = ownrow([&](int g) { return s[g]; }, t);
            ao[t] = ownrow([&](int g) { return arow[g]; }, t);
            yo[t] = ownrow([&](int g) { return y[NF + g]; }, t);
            bito[t] = pw * T(1u << (LG * t));                // 2^g of the own row (a lane without a row in the slot never passes)
        }
    
    }
    ATACOM_MARK("CH_metric");
    // ---- replicated prologue: metric of the soft rows, its Cholesky factor
    bool soft[NG], isp[NG];
    bool has_stiff = false;
    T VL[S][N1], xl[S], Ul[S];
    {
        SymMat<T, NQ> M;
        T b[NQ];
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            b[i] = T(0);
#pragma unroll
            for (int j = 0; j <= i; ++j) M(i, j) = (i == j) ? T(1) : T(0);
        }
        bool has_p = false;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int r = NF + g;
            soft[g] = num<T>::abs(s[g]) >= CC::THETA * arow[g];
            isp[g] = !soft[g] && !has_p;
            has_p = has_p || !soft[g];
            const T om = soft[g] ? num<T>::rcp(s[g] * s[g]) : T(0);
            T wa[NQ];
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                if (E::jac_zero(r, i)) continue;
                wa[i] = om * A[r][i];
                b[i] = num<T>::fma(wa[i], y[r], b[i]);
#pragma unroll
                for (int j = 0; j <= i; ++j)
                    if (!E::jac_zero(r, j)) M(i, j) = num<T>::fma(wa[i], A[r][j], M(i, j));
            }
        }
        has_stiff = has_p;
    ATACOM_MARK("CH_chol");
        T Li[NQ][NQ];
        chol_inverse_factor<T, NQ>(M, Li);
        T z[NQ], x0[NQ];
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            T a = T(0);
#pragma unroll
            for (int j = 0; j <= k; ++j) a = num<T>::fma(Li[k][j], b[j], a);
            z[k] = a;
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            T a = T(0);
#pragma unroll
            for (int k = i; k < NQ; ++k) a = num<T>::fma(Li[k][i], z[k], a);
            x0[i] = -a;
        }
    ATACOM_MARK("CH_own");
        const T hp = has_p ? T(1) : T(0);
#pragma unroll
        for (int sl = 0; sl < S; ++sl) {
            xl[sl] = own([&](int i) { return i < NQ ? x0[i < NQ ? i : 0] : T(0); }, sl);
            Ul[sl] = T(0);
#pragma unroll
            for (int k = 0; k < N1; ++k)
                VL[sl][k] = own([&](int i) {
                    return (k < NQ && i < NQ) ? ((k >= i) ? Li[k < NQ ? k : 0][i < NQ ? i : 0] : T(0))
                                              : ((k == NQ && i == NQ) ? hp : T(0));
                }, sl);
        }
    }
    // g_own = v_own . w;  v_own -= cproj g_own w
    auto project = [&](const T (&w)[N1], const T cproj, T (&g)[S]) {
#pragma unroll
        for (int sl = 0; sl < S; ++sl) {
            T a = T(0);
#pragma unroll
            for (int k = 0; k < N1; ++k) a = num<T>::fma(VL[sl][k], w[k], a);
            g[sl] = a;
            const T gc = a * cproj;
#pragma unroll
            for (int k = 0; k < N1; ++k) VL[sl][k] = num<T>::fma(-gc, w[k], VL[sl][k]);
        }
    };
    // the vector of the functional (A_r, cw) and its value on a distributed coordinate vector: group sums
    auto functional = [&](const auto& row, const T cw, T (&w)[N1], const T (&vl)[S], T& fval) {
        using R = std::decay_t<decltype(row)>;
        T part = T(0);
#pragma unroll
        for (int k = 0; k < N1; ++k) w[k] = T(0);
#pragma unroll
        for (int sl = 0; sl < S; ++sl) {
            const T c = own([&](int i) { return i < NQ ? (R::zero(i < NQ ? i : 0) ? T(0) : row(i < NQ ? i : 0)) : cw; }, sl);
#pragma unroll
            for (int k = 0; k < N1; ++k) w[k] = num<T>::fma(c, VL[sl][k], w[k]);
            part = num<T>::fma(c, vl[sl], part);
        }
#pragma unroll
        for (int k = 0; k < N1; ++k) w[k] = qsum<LG>(w[k]);
        fval = qsum<LG>(part);
    };
    auto condition = [&](const auto& row, const T cw, const T s2, const T yr, const bool on) {
        using R = std::decay_t<decltype(row)>;
        T w[N1], g[S], fx;
        functional(row, cw, w, xl, fx);
        T ww = T(0), nrm = num<T>::fma(cw, cw, s2);
#pragma unroll
        for (int k = 0; k < N1; ++k) ww = num<T>::fma(w[k], w[k], ww);
#pragma unroll
        for (int i = 0; i < NQ; ++i)
            if (!R::zero(i)) nrm = num<T>::fma(row(i), row(i), nrm);
        const T e = -yr - fx;
        const T Sv = s2 + ww;
        const bool ok = on && (Sv > CC::REL * nrm);
        const T iS = ok ? num<T>::rcp(Sv) : T(0);
        const T c = iS * num<T>::rcp(T(1) + num<T>::sqrt(s2 * iS));
        project(w, c, g);
        const T ce = e * iS;
#pragma unroll
        for (int sl = 0; sl < S; ++sl) xl[sl] = num<T>::fma(g[sl], ce, xl[sl]);
    };
    ATACOM_MARK("CH_eq");
    if constexpr (NF == 1) condition(StaticRow<T, E, 0>{A}, T(0), T(0), y[0], true);
    if (__builtin_amdgcn_ballot_w64(has_stiff) != 0ull) {
        [[maybe_unused]] const int trips = for_each_stiff_row<T, E>(A, s, y, soft, condition);
#ifdef ATACOM_TIMESTAMPS
        dbg[0] += trips;
#endif
    }
    ATACOM_MARK("CH_joints");
    // ---- the chart: conditioning recursion over the joints with the skip rule
    int n_acc = 0;
    static_for<0, NQ>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if (j >= NK && __builtin_amdgcn_ballot_w64(n_acc < NK) == 0ull) return;
        T w[N1], g[S];
        T dj = T(0);
#pragma unroll
        for (int k = 0; k < N1; ++k) { w[k] = qbcast<j % LG, LG>(VL[j / LG][k]); dj = num<T>::fma(w[k], w[k], dj); }
        const bool acc = (n_acc < NK) && (dj > tol2);
        // alpha[n_acc] as a one-hot blend: written as a select chain the optimiser turns it into a dynamically indexed
        // private array, promotes that to LDS, and -- to address it -- reads the workgroup size from the AQL dispatch packet
        // in HOST memory: 2 us per wave alone, up to 13 us with a full launch queueing for it (profiles/tools/gpu_phase_probe.py)
        T tv = T(0);
#pragma unroll
        for (int i = 0; i < NK; ++i) tv = num<T>::fma((n_acc == i) ? T(1) : T(0), alpha[i], tv);
        const T inv = acc ? num<T>::rcp(dj) : T(0);
        const T coef = (tv - qbcast<j % LG, LG>(Ul[j / LG])) * inv;
        project(w, inv, g);
#pragma unroll
        for (int sl = 0; sl < S; ++sl) Ul[sl] = num<T>::fma(g[sl], coef, Ul[sl]);
        n_acc += acc ? 1 : 0;
    });
    ATACOM_MARK("CH_stageA");
    // ---- free coordinates still missing after the joints (see canonical_mu)
    // (flags are recomputed where they are cheap: every bool array costs 2 SGPRs per row, and the masks spill)
    bool sel[NG];
    T wtgt[NG];
    auto tiny = [&](int g) -> bool { return !isp[g] && (num<T>::abs(s[g]) < CC::TINY * arow[g]); };
#pragma unroll
    for (int g = 0; g < NG; ++g) { sel[g] = false; wtgt[g] = T(0); }
    bool done = false;
    if constexpr (NK >= 2) {
#pragma unroll 1
        for (int it = 0; it < NK - 1; ++it) {
            const bool want = (n_acc < NK - 1) && !done;
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(want) == 0ull, 1)) break;
            ATACOM_DBG_COUNT(1);
            T tv = T(0);                            // one-hot blend, see the joint recursion
#pragma unroll
            for (int i = 0; i < NK; ++i) tv = num<T>::fma((n_acc == i) ? T(1) : T(0), alpha[i], tv);
            T wsel[N1], vsel = T(0), rsel = T(0);
#pragma unroll
            for (int k = 0; k < N1; ++k) wsel[k] = T(0);
            bool any = false, tnsel = false;
            // the coordinate slack: f_p = w_p, i.e. the vector v_NQ and the coordinate U_NQ themselves
            T wp[N1];
#pragma unroll
            for (int k = 0; k < N1; ++k) wp[k] = qbcast<NQ % LG, LG>(VL[NQ / LG][k]);
            const T fp = qbcast<NQ % LG, LG>(Ul[NQ / LG]);
            if constexpr (STATIC_A && (ATACOM_CHART_STAGEA_STATIC != 0) && (NG > 6)) {     // (planar, 6 rows: the static form spills)
                // first fit in STATIC row order with a wave-uniform early-out: row g is looked at while some environment of the
                // wavefront still has it as a candidate and has not found its column -- as many rows as the slowest
                // environment scans, but each on compile-time indices (structural zeros, no row gather)
                static_for<0, NG>([&](auto gc) {
                    constexpr int g = decltype(gc)::value;
                    constexpr int r = NF + g;
                    const bool act = want && !sel[g] && !any;
                    if (__builtin_amdgcn_ballot_w64(act) == 0ull) return;
                    T wa[N1], fa, v = T(0);
                    functional(StaticRow<T, E, r>{A}, T(0), wa, Ul, fa);                          // f_g = A_g u
                    T wg[N1];
#pragma unroll
                    for (int k = 0; k < N1; ++k) { wg[k] = isp[g] ? wp[k] : wa[k]; v = num<T>::fma(wg[k], wg[k], v); }
                    const T fu = isp[g] ? fp : fa;
                    const T thr = tol2 * (isp[g] ? T(1) : s[g] * s[g]);
                    const bool tn = tiny(g);
                    const bool take = act && (tn || (v > thr));
#pragma unroll
                    for (int k = 0; k < N1; ++k) wsel[k] = take ? wg[k] : wsel[k];
                    vsel = take ? v : vsel;
                    rsel = take ? (isp[g] ? fu - tv : num<T>::fma(s[g], tv, fu)) : rsel;
                    tnsel = take ? tn : tnsel;
                    wtgt[g] = take ? tv : wtgt[g];
                    sel[g] = sel[g] || take;
                    any = any || take;
                });
            } else {
                // first fit, per lane group: trip n tests the environment's n-th untaken column (PER-LANE ROWS above)
                unsigned cand = 0u, pbit = 0u, gsel = 0u;
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    cand |= (want && !sel[g]) ? (1u << g) : 0u;
                    pbit |= isp[g] ? (1u << g) : 0u;
                }
#pragma unroll 1
                while (__builtin_amdgcn_ballot_w64((cand != 0u) && !any) != 0ull) {
                    const bool act = (cand != 0u) && !any;
                    const unsigned low = act ? (cand & (0u - cand)) : 0u;
                    cand ^= low;
                    LaneRow<T, E> row;
                    lane_row<T, E>(A, low, row);
                    const T sg = lane_pick<T, NG>([&](int g) { return s[g]; }, low);
                    const T ag = lane_pick<T, NG>([&](int g) { return arow[g]; }, low);
                    const bool ip = (low & pbit) != 0u;
                    T wa[N1], fa, v = T(0);
                    functional(row, T(0), wa, Ul, fa);                                             // f_g = A_g u
                    T wg[N1];
#pragma unroll
                    for (int k = 0; k < N1; ++k) { wg[k] = ip ? wp[k] : wa[k]; v = num<T>::fma(wg[k], wg[k], v); }
                    const T fu = ip ? fp : fa;
                    const T thr = tol2 * (ip ? T(1) : sg * sg);
                    const bool tn = !ip && (num<T>::abs(sg) < CC::TINY * ag);
                    const bool take = act && (tn || (v > thr));
#pragma unroll
                    for (int k = 0; k < N1; ++k) wsel[k] = take ? wg[k] : wsel[k];
                    vsel = take ? v : vsel;
                    rsel = take ? (ip ? fu - tv : num<T>::fma(sg, tv, fu)) : rsel;
                    tnsel = take ? tn : tnsel;
                    gsel = take ? low : gsel;
                    any = any || take;
                }
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const bool tk = gsel == (1u << g);
                    wtgt[g] = num<T>::fma(tk ? T(1) : T(0), tv, wtgt[g]);
                    sel[g] = sel[g] || tk;
                }
            }
            done = done || (want && !any);
            const bool live = any && (vsel > T(0)) && !tnsel;
            const T iv = live ? num<T>::rcp(vsel) : T(0);
            T g[S];
            project(wsel, iv, g);
            const T coef = rsel * iv;
#pragma unroll
            for (int sl = 0; sl < S; ++sl) Ul[sl] = num<T>::fma(-g[sl], coef, Ul[sl]);
            n_acc += any ? 1 : 0;
        }
    }
    ATACOM_MARK("CH_stageB");
    // (B) exactly one missing: every v_i = beta_i dhat
    const bool need1 = (n_acc == NK - 1) && !done;
    const T tv_last = alpha[NK - 1];
    // row slots: the flags of the lane's own rows (stage A, rare, works on the replicated ones)
    [[maybe_unused]] bool ispo[RS], selo[RS], hro[RS];
    [[maybe_unused]] T wto[RS];
    // WHEN the row slots are built.  The blends are not free: built unconditionally they cost a wavefront that stays on the
    // plain path 1.3 - 2.2 us per step (profiles/r03_phase_probe_3way.log, quiet states: 12.7 -> 15.0 us per wave).  The
    // single-step kernels (STATIC_A) therefore build them only where some environment of the wavefront needs slack stage B
    // (wave-uniform), and only then does the assembly take its row-slot form -- a wavefront on the plain path runs the
    // replicated assembly.  The T-step kernels build them up front: there the blend of A leaves the sub-step loop (A is held
    // over the sub-steps); built inside the branch it runs four times a step (120-step collection 2.14 vs 2.28 ms).
    // (Both placements are written out: routed through a shared lambda the arrays went to scratch.)
    if constexpr (ROWDIST && !STATIC_A) {
#pragma unroll
        for (int t = 0; t < RS; ++t) {
            ispo[t] = ownflag([&](int g) { return isp[g]; }, t);
            selo[t] = ownflag([&](int g) { return sel[g]; }, t);
            wto[t] = ownrow([&](int g) { return wtgt[g]; }, t);
            hro[t] = (LG * t + LG <= NG) ? true : (lq < NG - LG * t);         // the lane has a row in this slot
        }
    
    }
    const bool rowd = ROWDIST && (!STATIC_A || (__builtin_amdgcn_ballot_w64(need1) != 0ull));
    // the coordinates, replicated (needed by stage B and by the assembly)
    T xa[N1], Ua[N1];
    auto gather_all = [&]() {
        static_for<0, N1>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            xa[i] = qbcast<i % LG, LG>(xl[i / LG]);
            Ua[i] = qbcast<i % LG, LG>(Ul[i / LG]);
        });
    };
    if (__builtin_amdgcn_ballot_w64(need1) != 0ull) {
        ATACOM_DBG_COUNT(2);
        if constexpr (ROWDIST && STATIC_A) {
            T pw = T(0);                                         // 2^lq
#pragma unroll
            for (int l = 0; l < LG; ++l) pw = num<T>::fma(oh[l], T(1u << l), pw);
#pragma unroll
            for (int t = 0; t < RS; ++t) {
#pragma unroll
                for (int i = 0; i < NQ; ++i) {
                    T v = T(0);
#pragma unroll
                    for (int l = 0; l < LG; ++l)
                        if (LG * t + l < NG && !E::jac_zero(NF + LG * t + l, i)) v = num<T>::fma(oh[l], A[NF + LG * t + l][i], v);
                    Ao[t][i] = v;
                }
                so[t] = ownrow([&](int g) { return s[g]; }, t);
                ao[t] = ownrow([&](int g) { return arow[g]; }, t);
                yo[t] = ownrow([&](int g) { return y[NF + g]; }, t);
                bito[t] = pw * T(1u << (LG * t));                // 2^g of the own row (a lane without a row in the slot never passes)
            }
    
#pragma unroll
            for (int t = 0; t < RS; ++t) {
                ispo[t] = ownflag([&](int g) { return isp[g]; }, t);
                selo[t] = ownflag([&](int g) { return sel[g]; }, t);
                wto[t] = ownrow([&](int g) { return wtgt[g]; }, t);
                hro[t] = (LG * t + LG <= NG) ? true : (lq < NG - LG * t);         // the lane has a row in this slot
            }
    
        }
        // the longest vector (first maximum in coordinate order, like np.argmax) and who owns it
        T nrm2[N1];
        static_for<0, N1>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            T a = T(0);
#pragma unroll
            for (int k = 0; k < N1; ++k) a = num<T>::fma(VL[i / LG][k], VL[i / LG][k], a);
            nrm2[i] = qbcast<i % LG, LG>(a);
        });
        T sig = nrm2[0];
        int jm = 0;
#pragma unroll
        for (int j = 1; j < N1; ++j) {
            const bool better = nrm2[j] > sig;
            sig = better ? nrm2[j] : sig;
            jm = better ? j : jm;
        }
        const int lane = (int)(threadIdx.x & 63u);
        const int src = lane - lq + (jm % LG);
        const int jsl = jm / LG;
        T dh[N1];
#pragma unroll
        for (int k = 0; k < N1; ++k) {
            T v = VL[0][k];
#pragma unroll
            for (int sl = 1; sl < S; ++sl) v = (jsl == sl) ? VL[sl][k] : v;
            dh[k] = lane_gather(v, src);
        }
        const T isd = (sig > T(0)) ? num<T>::rcp(num<T>::sqrt(sig)) : T(0);
        T bl[S];
#pragma unroll
        for (int sl = 0; sl < S; ++sl) {
            T a = T(0);
#pragma unroll
            for (int k = 0; k < N1; ++k) a = num<T>::fma(VL[sl][k], dh[k] * isd, a);
            bl[sl] = a;
        }
        T beta[N1];
        static_for<0, N1>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            beta[i] = qbcast<i % LG, LG>(bl[i / LG]);
            Ua[i] = qbcast<i % LG, LG>(Ul[i / LG]);
        });
        T fds = T(0), rs = T(0);
        if constexpr (ROWDIST) {
            // every lane tests ITS rows; the passing columns of the environment as one number, bit g = column g (a group sum
            // of disjoint powers of two: exact), first fit = its lowest set bit
            T fdo[RS], reso[RS], valo[RS], mf = T(0);
            bool tno[RS];
#pragma unroll
            for (int t = 0; t < RS; ++t) {
                T a = T(0), fu = T(0);
#pragma unroll
                for (int i = 0; i < NQ; ++i) {
                    if (slot_zero(t, i)) continue;
                    a = num<T>::fma(Ao[t][i], beta[i], a);
                    fu = num<T>::fma(Ao[t][i], Ua[i], fu);
                }
                fdo[t] = ispo[t] ? beta[NQ] : a;
                reso[t] = ispo[t] ? Ua[NQ] - tv_last : num<T>::fma(so[t], tv_last, fu);
                valo[t] = fdo[t] * fdo[t];
                tno[t] = !ispo[t] && (num<T>::abs(so[t]) < CC::TINY * ao[t]);
                const T thr = tol2 * (ispo[t] ? T(1) : so[t] * so[t]);
                const bool pass = hro[t] && !selo[t] && (tno[t] || (valo[t] > thr));
                mf = num<T>::fma(pass ? T(1) : T(0), bito[t], mf);
            }
            int low = (int)qsum<LG>(mf);
            low = low & (0 - low);
            // nothing passed (a numerically rank-deficient remainder): the untaken column with the largest projection
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(need1 && (low == 0)) != 0ull, 0)) {
                T vb = T(-1);
#pragma unroll
                for (int t = 0; t < RS; ++t) vb = (hro[t] && !selo[t]) ? num<T>::max(vb, valo[t]) : vb;
                vb = qmax<LG>(vb);
                T mb = T(0);
#pragma unroll
                for (int t = 0; t < RS; ++t)
                    mb = num<T>::fma((hro[t] && !selo[t] && (valo[t] == vb)) ? T(1) : T(0), bito[t], mb);
                int lb = (int)qsum<LG>(mb);
                lb = lb & (0 - lb);
                low = (low == 0) ? lb : low;
            }
            T fdp = T(0), rsp = T(0);
#pragma unroll
            for (int t = 0; t < RS; ++t) {
                const bool pk = need1 && hro[t] && (low == (int)bito[t]);
                fdp = pk ? (tno[t] ? T(0) : fdo[t]) : fdp;
                rsp = pk ? reso[t] : rsp;
                wto[t] = pk ? tv_last : wto[t];
                selo[t] = selo[t] || pk;
            }
            fds = qsum<LG>(fdp);                               // one lane contributes: exact
            rs = qsum<LG>(rsp);
        } else {
            // (per row on purpose: the column that passes here is typically a LATE one -- the slack of a joint limit, column
            // 16 of 17 in the iiwa chart (0, 1, 2, 3, 16) -- so a per-lane first-fit scan runs ten trips before it: measured
            // 4 - 5 us per execution against 1.3 us for testing all eleven columns side by side)
            T fd[NG], res[NG], val[NG];
            bool pick[NG];
            bool any = false;
            T vbest = T(-1);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int r = NF + g;
                T a = T(0), fu = T(0);
#pragma unroll
                for (int i = 0; i < NQ; ++i) {
                    if (E::jac_zero(r, i)) continue;
                    a = num<T>::fma(A[r][i], beta[i], a);
                    fu = num<T>::fma(A[r][i], Ua[i], fu);
                }
                fd[g] = isp[g] ? beta[NQ] : a;
                res[g] = isp[g] ? Ua[NQ] - tv_last : num<T>::fma(s[g], tv_last, fu);
                val[g] = fd[g] * fd[g];
                const T thr = tol2 * (isp[g] ? T(1) : s[g] * s[g]);
                const bool pass = !sel[g] && (tiny(g) || (val[g] > thr));
                pick[g] = need1 && pass && !any;
                any = any || pass;
                vbest = sel[g] ? vbest : num<T>::max(vbest, val[g]);
            }
            bool taken = false;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const bool fb = need1 && !any && !taken && !sel[g] && (val[g] == vbest);
                pick[g] = pick[g] || fb;
                taken = taken || fb;
            }
            fds = T(0); rs = T(0);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                fds = pick[g] ? (tiny(g) ? T(0) : fd[g]) : fds; rs = pick[g] ? res[g] : rs;
                wtgt[g] = pick[g] ? tv_last : wtgt[g];
                sel[g] = sel[g] || pick[g];
            }
        }
        const T coef = (need1 && fds != T(0)) ? num<T>::div(rs, fds) : T(0);
#pragma unroll
        for (int sl = 0; sl < S; ++sl) Ul[sl] = num<T>::fma(-bl[sl], coef, Ul[sl]);
    }
    ATACOM_MARK("CH_asm");
    gather_all();
    // ---- assembly (replicated), as in canonical_mu
    if constexpr (NF == 1) {
        T aa = T(0), au = y[0], aU = T(0);
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            if (E::jac_zero(0, i)) continue;
            aa = num<T>::fma(A[0][i], A[0][i], aa);
            au = num<T>::fma(A[0][i], xa[i], au);
            aU = num<T>::fma(A[0][i], Ua[i], aU);
        }
        const T iaa = (aa > T(0)) ? num<T>::rcp(aa) : T(0);
        au *= iaa; aU *= iaa;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            if (E::jac_zero(0, i)) continue;
            xa[i] = num<T>::fma(-A[0][i], au, xa[i]);
            Ua[i] = num<T>::fma(-A[0][i], aU, Ua[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) mu[i] = xa[i] + Ua[i];
    if (rowd) {
        // the slack velocities of the lane's own rows, then one broadcast per row
        T wo[RS];
#pragma unroll
        for (int t = 0; t < RS; ++t) {
            T wm = yo[t], wa = T(0);
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                if (slot_zero(t, i)) continue;
                wm = num<T>::fma(Ao[t][i], xa[i], wm);
                wa = num<T>::fma(Ao[t][i], Ua[i], wa);
            }
            const T inv_s = (hro[t] && (num<T>::abs(so[t]) >= CC::TINY * ao[t])) ? num<T>::rcp(so[t]) : T(0);
            const T w = selo[t] ? num<T>::fma(-wm, inv_s, wto[t]) : -(wm + wa) * inv_s;
            wo[t] = ispo[t] ? xa[NQ] + Ua[NQ] : w;
        }
        static_for<0, NG>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            mu[NQ + g] = qbcast<g % LG, LG>(wo[g / LG]);
        });
    } else {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int r = NF + g;
            T wm = y[r], wa = T(0);
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                if (E::jac_zero(r, i)) continue;
                wm = num<T>::fma(A[r][i], xa[i], wm);
                wa = num<T>::fma(A[r][i], Ua[i], wa);
            }
            const T inv_s = (num<T>::abs(s[g]) >= CC::TINY * arow[g]) ? num<T>::rcp(s[g]) : T(0);
            const T w = sel[g] ? num<T>::fma(-wm, inv_s, wtgt[g]) : -(wm + wa) * inv_s;
            mu[NQ + g] = isp[g] ? xa[NQ] + Ua[NQ] : w;
        }
    }
}

}  // namespace atacom

#include "atacom_chart_group.h"
