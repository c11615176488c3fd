// Third form of the canonical chart's lane-group kernels (LG = 2 / 4 / 8 lanes per environment; round 4).
//
// Same recursion, same decisions as canonical_mu / canonical_mu_group (atacom_chart.h; specification
// oracle/canonical_chart.py; reference semantics outside the chart's tolerance band: null_space_coordinate.py:40-79,
// atacom.py:127-133).  What changed is WHO computes what.  Measured on the bench workload (8192 iiwa environments, 8 lanes,
// profiles/r04_bench_probe_baseline.log): the data-dependent parts are not rare there -- the reset pose stands 7 cm from
// the table's near boundary, so under random actions 79 % of the wavefronts run a stiff-row step and 85 % slack stage B in
// (nearly) every sub-step -- and a third of the plain path was a prologue every lane of a group repeated.  Now:
//   * OWN COLUMNS.  Lane l keeps column i = LG sl + l of A = K J (At[sl][r]; built once per env step, A is held over
//     the sub-steps).  The coefficient with which coordinate i enters a row functional is then a register, not a one-hot
//     blend over the row (eight multiply-adds per functional, 66 + 22 per stiff-row step).
//   * METRIC BY COLUMNS.  Lane l accumulates column i of M = I + sum A_g^T A_g / s_g^2 (six entries) and entry i of
//     b = sum A_g^T y_g / s_g^2 from its own column of A; the lower triangle is gathered (21 broadcasts) for the replicated
//     6 x 6 Cholesky factor -- instead of every lane accumulating all 21 + 6 entries.
//   * NO INVERSE FACTOR.  v_i = L^-1 e_i is a forward substitution with the lane's own unit vector (the one-hot of its
//     position): 21 instructions instead of the replicated inverse (50) plus the blends that picked its column (64).
//     x_0 = -Gamma b: z = L^-1 b as the group sum of b_i v_i, then x_i = -v_i . z.
//   * ROW SLOTS ALWAYS.  Inequality row g belongs to lane g % LG, slot g / LG (as in the second form), but its copy of the
//     row is built once per env step for every wavefront, and which rows are the coordinate slack / already taken travels
//     as two bit masks tested against the lane's own bit -- no boolean blends (the second form built the slots inside the
//     stage-B branch, 150 instructions each time, to spare quiet wavefronts; on the bench workload that branch is the rule).
//   * the equality row's final exact correction works on the distributed coordinates (two group sums) before they are
//     gathered, instead of on replicated copies after.
// Everything a decision depends on is replicated or group-summed, so the lanes of a group still agree bit for bit.
#pragma once

namespace atacom {

// Cholesky factor of the SPD matrix whose LOWER triangle is Ml (replicated): L and 1 / L_jj.
template <typename T, int N>
__device__ __forceinline__ void chol_factor(const T (&Ml)[N][N], T (&L)[N][N], T (&inv)[N]) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
        T d = Ml[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) d = num<T>::fma(-L[j][k], L[j][k], d);
        d = num<T>::max(d, chart_const<T>::FLOOR);
        const T r = num<T>::rcp(num<T>::sqrt(d));
        inv[j] = r;
        L[j][j] = d * r;
#pragma unroll
        for (int i = j + 1; i < N; ++i) {
            T a = Ml[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) a = num<T>::fma(-L[i][k], L[j][k], a);
            L[i][j] = a * r;
        }
    }
}

// What a lane keeps of A = K J over the sub-steps of an env step (A is held: quirk Q1; with hold_q = 0 it is rebuilt per
// sub-step along with A).  Coordinate i of the extended state (i < NQ: joint i; i = NQ: the coordinate slack) lives in lane
// i % LG, slot i / LG; inequality row g in lane g % LG, slot g / LG.
template <typename T, typename E, int LG>
struct ChartPre {
    static constexpr int NQ = E::NQ, NF = E::NF, NG = E::NG, NC = E::NC, N1 = NQ + 1;
    static constexpr int S = (N1 + LG - 1) / LG;
    static constexpr int RS = (NG + LG - 1) / LG;
    T oh[LG];               // one-hot of the lane's position in its group (blends, never a switch: atacom_chart.h)
    T At[S][NC];            // own columns of A: At[sl][r] = A[r][LG sl + lq]  (0 for the coordinate slack and past the end)
    T e6[S];                // 1 where the slot's coordinate is the coordinate slack
    T Ao[RS][NQ];           // own rows of A: Ao[t][i] = A[NF + LG t + lq][i]  (0 where the lane has no row)
    T ao[RS], ybo[RS], kco[RS];   // of the own rows: max |A_g|, the slack-independent part of y, K_c
    unsigned obit[RS];      // bit (LG t + lq) of the own row; 0 where the lane has no row
    T bito[RS];             // the same as a number (the passing columns travel as a group sum of disjoint powers of two)
    T aa, iaa;              // |a|^2 of the equality row and its reciprocal (0 for a vanishing row)
};

template <typename T, typename E, int LG>
__device__ __forceinline__ void chart_prepare(const T (&A)[E::NC][E::NQ], const T (&arow)[E::NG], const T (&yb)[E::NC],
                                              const T (&Kc)[12], const int lq, ChartPre<T, E, LG>& cp) {
    using CP = ChartPre<T, E, LG>;
    constexpr int NQ = E::NQ, NF = E::NF, NG = E::NG, NC = E::NC, S = CP::S, RS = CP::RS;
#pragma unroll
    for (int l = 0; l < LG; ++l) cp.oh[l] = (lq == l) ? T(1) : T(0);
#pragma unroll
    for (int sl = 0; sl < S; ++sl) {
        T e = T(0);
#pragma unroll
        for (int l = 0; l < LG; ++l)
            if (LG * sl + l == NQ) e = cp.oh[l];
        cp.e6[sl] = e;
#pragma unroll
        for (int r = 0; r < NC; ++r) {
            T v = T(0);
#pragma unroll
            for (int l = 0; l < LG; ++l) {
                const int i = LG * sl + l;
                if (i < NQ && !E::jac_zero(r, i < NQ ? i : 0)) v = num<T>::fma(cp.oh[l], A[r][i < NQ ? i : 0], v);
            }
            cp.At[sl][r] = v;
        }
    }
#pragma unroll
    for (int t = 0; t < RS; ++t) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            T v = T(0);
#pragma unroll
            for (int l = 0; l < LG; ++l) {
                const int g = LG * t + l;
                if (g < NG && !E::jac_zero(NF + (g < NG ? g : 0), i)) v = num<T>::fma(cp.oh[l], A[NF + (g < NG ? g : 0)][i], v);
            }
            cp.Ao[t][i] = v;
        }
        T a = T(0), yv = T(0), kc = T(0);
#pragma unroll
        for (int l = 0; l < LG; ++l) {
            const int g = LG * t + l;
            if (g < NG) {
                a = num<T>::fma(cp.oh[l], arow[g < NG ? g : 0], a);
                yv = num<T>::fma(cp.oh[l], yb[NF + (g < NG ? g : 0)], yv);
                kc = num<T>::fma(cp.oh[l], Kc[NF + (g < NG ? g : 0)], kc);
            }
        }
        cp.ao[t] = a; cp.ybo[t] = yv; cp.kco[t] = kc;
        cp.obit[t] = (LG * t + lq < NG) ? (1u << (LG * t + lq)) : 0u;
        cp.bito[t] = (T)cp.obit[t];
    }
    T aa = T(0);
    if constexpr (NF == 1) {
#pragma unroll
        for (int i = 0; i < NQ; ++i)
            if (!E::jac_zero(0, i)) aa = num<T>::fma(A[0][i], A[0][i], aa);
    }
    cp.aa = aa;
    cp.iaa = (aa > T(0)) ? num<T>::rcp(aa) : T(0);
}

// group sums of n values, in place (float: the grouped v_add_f32_dpp sequences of atacom_quad.h -- 6 / 4 / 2 sums per sequence)
template <int LG> struct gsum_fn;
template <> struct gsum_fn<8> { template <typename... A> static __device__ __forceinline__ void run(A&... a) { osum_n(a...); } };
template <> struct gsum_fn<4> { template <typename... A> static __device__ __forceinline__ void run(A&... a) { qsum_n(a...); } };
template <> struct gsum_fn<2> { template <typename... A> static __device__ __forceinline__ void run(A&... a) { psum_n(a...); } };
template <int LG, typename T, int N>
__device__ __forceinline__ void gsum_all(T (&w)[N]) {
    if constexpr (std::is_same<T, float>::value) {
        constexpr int N6 = N / 6 * 6;
        static_for<0, N6 / 6>([&](auto bc) {
            constexpr int b = decltype(bc)::value * 6;
            gsum_fn<LG>::run(w[b], w[b + 1], w[b + 2], w[b + 3], w[b + 4], w[b + 5]);
        });
        if constexpr (N - N6 >= 4) {
            gsum_fn<LG>::run(w[N6], w[N6 + 1], w[N6 + 2], w[N6 + 3]);
            if constexpr (N - N6 == 5) w[N6 + 4] = qsum<LG>(w[N6 + 4]);
        } else if constexpr (N - N6 >= 2) {
            gsum_fn<LG>::run(w[N6], w[N6 + 1]);
            if constexpr (N - N6 == 3) w[N6 + 2] = qsum<LG>(w[N6 + 2]);
        } else if constexpr (N - N6 == 1) {
            w[N6] = qsum<LG>(w[N6]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < N; ++k) w[k] = qsum<LG>(w[k]);
    }
}

template <typename T, typename E, int LG, bool STATIC_A = false>
__device__ __forceinline__ void canonical_mu_group3(const T (&A)[E::NC][E::NQ], const ChartPre<T, E, LG>& cp,
                                                    const T (&arow)[E::NG], const T (&s)[E::NG], const T (&y)[E::NC],
                                                    const T (&alpha)[E::NK], const T tol, T (&mu)[E::NN],
                                                    const int lq ATACOM_DBG_PARAM) {
    using CP = ChartPre<T, E, LG>;
    constexpr int NQ = E::NQ, NF = E::NF, NG = E::NG, NK = NQ - NF, N1 = NQ + 1;
    constexpr int S = CP::S, RS = CP::RS;
    constexpr int S6 = (NQ + LG - 1) / LG;             // slots that hold a joint column
    static_assert(NF <= 1, "at most one equality row");
    static_assert(NG <= 31, "row masks are 32-bit");
    using CC = chart_const<T>;
    const T tol2 = tol * tol;
    // entry (row slot t, joint i) is structurally zero for every lane of the group
    auto slot_zero = [](int t, int i) constexpr -> bool {
        bool z = true;
        for (int l = 0; l < LG; ++l)
            if (LG * t + l < NG) z = z && E::jac_zero(NF + LG * t + l, i);
        return z;
    };
    // one-hot of "the slot's coordinate is joint i" -- a compile-time choice of cp.oh[]
    auto eo = [&](int sl, int i) -> T { return (i - LG * sl >= 0 && i - LG * sl < LG) ? cp.oh[(i - LG * sl >= 0 && i - LG * sl < LG) ? i - LG * sl : 0] : T(0); };
    ATACOM_MARK("C3_metric");
    // ---- weights of the soft rows (replicated); which rows are stiff, which is the coordinate slack
    T om[NG];
    unsigned stiffm = 0u, pmask = 0u;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const bool soft = num<T>::abs(s[g]) >= CC::THETA * arow[g];
        om[g] = soft ? num<T>::rcp(s[g] * s[g]) : T(0);
        stiffm |= soft ? 0u : (1u << g);
    }
    pmask = stiffm & (0u - stiffm);                        // the first stiff row: its slack is a coordinate of the state
    const bool has_stiff = stiffm != 0u;
    // ---- own columns of the metric M = I + sum_soft A_g^T A_g / s_g^2, own entries of b = sum_soft A_g^T y_g / s_g^2
    T Mc[S6][NQ], bo[S6];
#pragma unroll
    for (int sl = 0; sl < S6; ++sl) {
        bo[sl] = T(0);
#pragma unroll
        for (int i = 0; i < NQ; ++i) Mc[sl][i] = eo(sl, i);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int r = NF + g;
            const T wa = om[g] * cp.At[sl][r];
            bo[sl] = num<T>::fma(wa, y[r], bo[sl]);
#pragma unroll
            for (int i = 0; i < NQ; ++i)
                if (!E::jac_zero(r, i)) Mc[sl][i] = num<T>::fma(wa, A[r][i], Mc[sl][i]);
        }
    }
    ATACOM_MARK("C3_chol");
    // the lower triangle, gathered: column j from its owner
    T Ml[NQ][NQ], L[NQ][NQ], inv[NQ];
    static_for<0, NQ>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        static_for<j, NQ>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            Ml[i][j] = qbcast<j % LG, LG>(Mc[j / LG][i]);
        });
    });
    chol_factor<T, NQ>(Ml, L, inv);
    ATACOM_MARK("C3_vecs");
    // ---- v_own = L^-1 e_own (Gamma = M^-1 = L^-T L^-1, v_i = column i of L^-1: Gamma_ij = v_i . v_j)
    T VL[S][N1], xl[S], Ul[S];
    const T hp = has_stiff ? T(1) : T(0);
#pragma unroll
    for (int sl = 0; sl < S; ++sl) {
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            T acc = (sl < S6) ? eo(sl, k) : T(0);
#pragma unroll
            for (int j = 0; j < k; ++j) acc = num<T>::fma(-L[k][j], VL[sl][j], acc);
            VL[sl][k] = acc * inv[k];
        }
        VL[sl][NQ] = cp.e6[sl] * hp;
        Ul[sl] = T(0);
    }
    // x = -Gamma b:  z = L^-1 b by FORWARD SUBSTITUTION on the replicated factor with b gathered, x_i = -v_i . z.
    // (First written as the group sum z = sum_i b_i v_i -- one product per lane, rounded on its own, then a tree of adds.  With a
    // soft row of weight 1 / s^2 ~ 10^3 the entries of b are ~10^4 and cancel in z_k = (b_k - L_k0 z_0 - ...) / L_kk down to
    // O(10): every separately rounded product left 6e-8 x 10^4 in z, x_0 inherited it, and the slack velocity
    // w_g = -(y_g + A_g x) / s_g amplified it by 1 / s_g.  The planar float32 soak caught it: 64 samples of 295 k beyond the
    // quick sensitivity bound against 1 for the second form -- whose z = Li b is one fused chain per entry, a single large
    // rounding -- and 7 unexplained; found by giving the third form the second form's prologue piece by piece
    // (profiles/r04_chart_form3_planar.md).  In the substitution every product is fused into the running remainder: the only
    // roundings are of the small results.  Same instruction count: NQ broadcasts + NQ (NQ + 1) / 2 operations.)
    {
        T bb[NQ], z[NQ];
        static_for<0, NQ>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            bb[i] = qbcast<i % LG, LG>(bo[i / LG]);
        });
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            T a = bb[k];
#pragma unroll
            for (int j = 0; j < k; ++j) a = num<T>::fma(-L[k][j], z[j], a);
            z[k] = a * inv[k];
        }
#pragma unroll
        for (int sl = 0; sl < S; ++sl) {
            T a = T(0);
#pragma unroll
            for (int k = 0; k < NQ; ++k) a = num<T>::fma(VL[sl][k], z[k], a);
            xl[sl] = -a;
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
    // the vector of the functional with own coefficients c[sl] (coordinate i enters with A_r[i], the coordinate slack with
    // cw), its value on a distributed coordinate vector, and the squared norm of its coefficients: group sums
    auto functional = [&](const T (&c)[S], T (&w)[N1], const T (&vl)[S], T& fval, T& cnrm) {
        T acc[N1 + 2];
#pragma unroll
        for (int k = 0; k < N1 + 2; ++k) acc[k] = T(0);
#pragma unroll
        for (int sl = 0; sl < S; ++sl) {
#pragma unroll
            for (int k = 0; k < N1; ++k) acc[k] = num<T>::fma(c[sl], VL[sl][k], acc[k]);
            acc[N1] = num<T>::fma(c[sl], vl[sl], acc[N1]);
            acc[N1 + 1] = num<T>::fma(c[sl], c[sl], acc[N1 + 1]);
        }
        gsum_all<LG>(acc);
#pragma unroll
        for (int k = 0; k < N1; ++k) w[k] = acc[k];
        fval = acc[N1];
        cnrm = acc[N1 + 1];
    };
    // exact rank-one conditioning of the state on  (row, cw) . x + (noise of variance s2) = -yr
    auto condition = [&](const T (&c)[S], const T s2, const T yr, const bool on) {
        T w[N1], g[S], fx, cn;
        functional(c, w, xl, fx, cn);
        T ww = T(0);
#pragma unroll
        for (int k = 0; k < N1; ++k) ww = num<T>::fma(w[k], w[k], ww);
        const T nrm = cn + s2;
        const T e = -yr - fx;
        const T Sv = s2 + ww;
        // Gamma <= I: a row that has nothing left to say (vanishing, or dependent on rows imposed before) is dropped
        const bool ok = on && (Sv > CC::REL * nrm);
        const T iS = ok ? num<T>::rcp(Sv) : T(0);
        // Potter: (I - c w w^T)^2 = I - w w^T / S  for  c = (1 / S) / (1 + sqrt(s2 / S))   (s2 = 0: the projection)
        const T cpj = iS * num<T>::rcp(T(1) + num<T>::sqrt(s2 * iS));
        project(w, cpj, g);
        const T ce = e * iS;
#pragma unroll
        for (int sl = 0; sl < S; ++sl) xl[sl] = num<T>::fma(g[sl], ce, xl[sl]);
    };
    // own coefficients of the inequality row whose bit is `low` (none: zeros), one-hot blend over the rows of the own columns
    auto lane_coeffs = [&](const unsigned low, const T cw, T (&c)[S]) {
        T h[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) h[g] = (low == (1u << g)) ? T(1) : T(0);
#pragma unroll
        for (int sl = 0; sl < S; ++sl) {
            T v = cw * cp.e6[sl];
            if (sl < S6) {
#pragma unroll
                for (int g = 0; g < NG; ++g) v = num<T>::fma(h[g], cp.At[sl < S6 ? sl : 0][NF + g], v);
            }
            c[sl] = v;
        }
    };
    ATACOM_MARK("C3_eq");
    if constexpr (NF == 1) {
        T c[S];
#pragma unroll
        for (int sl = 0; sl < S; ++sl) c[sl] = cp.At[sl][0];
        condition(c, T(0), y[0], true);
    }
    ATACOM_MARK("C3_stiff");
    if (__builtin_amdgcn_ballot_w64(has_stiff) != 0ull) {
        // PER-LANE ROWS (atacom_chart.h): trip n conditions every environment on ITS n-th stiff row; the first keeps its
        // slack velocity as a coordinate (prim: cw = s_p, no noise), further ones are measurement updates of variance s_g^2
        unsigned todo = stiffm;
        bool first = true;
#pragma unroll 1
        while (__builtin_amdgcn_ballot_w64(todo != 0u) != 0ull) {
            ATACOM_DBG_COUNT(0);
            const bool on = todo != 0u;
            const unsigned low = todo & (0u - todo);
            todo ^= low;
            const T sg = lane_pick<T, NG>([&](int g) { return s[g]; }, low);
            const T yg = lane_pick<T, NG>([&](int g) { return y[NF + g]; }, low);
            const bool prim = first && on;
            first = false;
            T c[S];
            lane_coeffs(low, prim ? sg : T(0), c);
            condition(c, prim ? T(0) : sg * sg, yg, on);
        }
    }
    ATACOM_MARK("C3_joints");
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
        T tv = T(0);                            // alpha[n_acc] as a one-hot blend (never a select chain: atacom_chart.h)
#pragma unroll
        for (int i = 0; i < NK; ++i) tv = num<T>::fma((n_acc == i) ? T(1) : T(0), alpha[i], tv);
        const T inv_d = acc ? num<T>::rcp(dj) : T(0);
        const T coef = (tv - qbcast<j % LG, LG>(Ul[j / LG])) * inv_d;
        project(w, inv_d, g);
#pragma unroll
        for (int sl = 0; sl < S; ++sl) Ul[sl] = num<T>::fma(g[sl], coef, Ul[sl]);
        n_acc += acc ? 1 : 0;
    });
    ATACOM_MARK("C3_stageA");
    // ---- free coordinates still missing after the joints: slack columns in column order, the first one that passes.
    // selm: the columns taken (replicated bits); wto: the target of an own row that was taken
    unsigned selm = 0u;
    T wto[RS];
#pragma unroll
    for (int t = 0; t < RS; ++t) wto[t] = T(0);
    // own-row copies of what changes per sub-step
    T so[RS], yo[RS];
#pragma unroll
    for (int t = 0; t < RS; ++t) {
        T v = T(0);
#pragma unroll
        for (int l = 0; l < LG; ++l)
            if (LG * t + l < NG) v = num<T>::fma(cp.oh[l], s[(LG * t + l < NG) ? LG * t + l : 0], v);
        so[t] = v;
        yo[t] = num<T>::fma(T(0.5) * cp.kco[t] * v, v, cp.ybo[t]);
    }
    bool done = false;
    if constexpr (NK >= 2) {
#pragma unroll 1
        for (int it = 0; it < NK - 1; ++it) {
            const bool want = (n_acc < NK - 1) && !done;
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(want) == 0ull, 1)) break;
            ATACOM_DBG_COUNT(1);
            T tv = T(0);
#pragma unroll
            for (int i = 0; i < NK; ++i) tv = num<T>::fma((n_acc == i) ? T(1) : T(0), alpha[i], tv);
            // ALL candidate rows at once, every lane its own (row slots): the vector of f_g = A_g u is a combination of ALL
            // vectors, so they are gathered first (N1 x N1 broadcasts) -- deterministic cost, whatever rows the environments
            // of the wavefront are at (the second form walked the rows: 80 instructions per row looked at, 190 per trip of
            // the per-lane scan, the passing column typically a late one)
            T Uall[N1];
            static_for<0, N1>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                Uall[i] = qbcast<i % LG, LG>(Ul[i / LG]);
            });
            T wg[RS][N1], vo[RS], fuo[RS], mf = T(0);
            bool tno[RS], ipo[RS];
#pragma unroll
            for (int t = 0; t < RS; ++t) { ipo[t] = (pmask & cp.obit[t]) != 0u; vo[t] = T(0); }
            static_for<0, N1>([&](auto kc) {                            // component k of every vector, then of every own row's
                constexpr int k = decltype(kc)::value;
                T col[N1];
                static_for<0, N1>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    col[i] = qbcast<i % LG, LG>(VL[i / LG][k]);
                });
#pragma unroll
                for (int t = 0; t < RS; ++t) {
                    T a = T(0);
#pragma unroll
                    for (int i = 0; i < NQ; ++i)
                        if (!slot_zero(t, i)) a = num<T>::fma(cp.Ao[t][i], col[i], a);
                    a = ipo[t] ? col[NQ] : a;                          // the coordinate slack: f_p = w_p itself
                    wg[t][k] = a;
                    vo[t] = num<T>::fma(a, a, vo[t]);
                }
            });
#pragma unroll
            for (int t = 0; t < RS; ++t) {
                T fu = T(0);
#pragma unroll
                for (int i = 0; i < NQ; ++i)
                    if (!slot_zero(t, i)) fu = num<T>::fma(cp.Ao[t][i], Uall[i], fu);
                fuo[t] = ipo[t] ? Uall[NQ] : fu;
                tno[t] = !ipo[t] && (num<T>::abs(so[t]) < CC::TINY * cp.ao[t]);
                const T thr = tol2 * (ipo[t] ? T(1) : so[t] * so[t]);
                const bool pass = want && (cp.obit[t] != 0u) && ((selm & cp.obit[t]) == 0u) && (tno[t] || (vo[t] > thr));
                mf = num<T>::fma(pass ? T(1) : T(0), cp.bito[t], mf);
            }
            unsigned low = (unsigned)qsum<LG>(mf);                      // disjoint powers of two: exact
            low = low & (0u - low);
            const bool any = low != 0u;
            // the picked row's vector and numbers reach the group as sums only its owner contributes to
            T acc[N1 + 3];
#pragma unroll
            for (int k = 0; k < N1 + 3; ++k) acc[k] = T(0);
#pragma unroll
            for (int t = 0; t < RS; ++t) {
                const bool pk = any && (low == cp.obit[t]);
#pragma unroll
                for (int k = 0; k < N1; ++k) acc[k] = pk ? wg[t][k] : acc[k];
                acc[N1] = pk ? vo[t] : acc[N1];
                // slack g: f_g(x) = -s_g target;  coordinate slack p: f_p(x) = +target
                acc[N1 + 1] = pk ? (ipo[t] ? fuo[t] - tv : num<T>::fma(so[t], tv, fuo[t])) : acc[N1 + 1];
                acc[N1 + 2] = pk ? (tno[t] ? T(1) : T(0)) : acc[N1 + 2];
                wto[t] = pk ? tv : wto[t];
            }
            gsum_all<LG>(acc);
            T wsel[N1];
#pragma unroll
            for (int k = 0; k < N1; ++k) wsel[k] = acc[k];
            const T vsel = acc[N1], rsel = acc[N1 + 1];
            const bool tnsel = acc[N1 + 2] != T(0);
            selm |= low;
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
    ATACOM_MARK("C3_stageB");
    // (B) exactly one missing: S is one-dimensional, every v_i = beta_i dhat -- a scalar test per row
    const bool need1 = (n_acc == NK - 1) && !done;
    const T tv_last = alpha[NK - 1];
    if (__builtin_amdgcn_ballot_w64(need1) != 0ull) {
        ATACOM_DBG_COUNT(2);
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
        T beta[N1], Ub[N1];
        static_for<0, N1>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            beta[i] = qbcast<i % LG, LG>(bl[i / LG]);
            Ub[i] = qbcast<i % LG, LG>(Ul[i / LG]);
        });
        // every lane tests ITS rows; the passing columns of the environment as one number, bit g = column g (a group sum
        // of disjoint powers of two: exact), first fit = its lowest set bit
        T fdo[RS], reso[RS], valo[RS], mf = T(0);
        bool tno[RS], ipo[RS], freeo[RS];
#pragma unroll
        for (int t = 0; t < RS; ++t) {
            ipo[t] = (pmask & cp.obit[t]) != 0u;
            freeo[t] = (cp.obit[t] != 0u) && ((selm & cp.obit[t]) == 0u);
            T a = T(0), fu = T(0);
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                if (slot_zero(t, i)) continue;
                a = num<T>::fma(cp.Ao[t][i], beta[i], a);
                fu = num<T>::fma(cp.Ao[t][i], Ub[i], fu);
            }
            fdo[t] = ipo[t] ? beta[NQ] : a;
            reso[t] = ipo[t] ? Ub[NQ] - tv_last : num<T>::fma(so[t], tv_last, fu);
            valo[t] = fdo[t] * fdo[t];
            tno[t] = !ipo[t] && (num<T>::abs(so[t]) < CC::TINY * cp.ao[t]);
            const T thr = tol2 * (ipo[t] ? T(1) : so[t] * so[t]);
            const bool pass = freeo[t] && (tno[t] || (valo[t] > thr));
            mf = num<T>::fma(pass ? T(1) : T(0), cp.bito[t], mf);
        }
        unsigned low = (unsigned)qsum<LG>(mf);
        low = low & (0u - low);
        // nothing passed (a numerically rank-deficient remainder): the untaken column with the largest projection
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(need1 && (low == 0u)) != 0ull, 0)) {
            T vb = T(-1);
#pragma unroll
            for (int t = 0; t < RS; ++t) vb = freeo[t] ? num<T>::max(vb, valo[t]) : vb;
            vb = qmax<LG>(vb);
            T mb = T(0);
#pragma unroll
            for (int t = 0; t < RS; ++t) mb = num<T>::fma((freeo[t] && (valo[t] == vb)) ? T(1) : T(0), cp.bito[t], mb);
            unsigned lb = (unsigned)qsum<LG>(mb);
            lb = lb & (0u - lb);
            low = (low == 0u) ? lb : low;
        }
        T pr[2] = {T(0), T(0)};
#pragma unroll
        for (int t = 0; t < RS; ++t) {
            const bool pk = need1 && (cp.obit[t] != 0u) && (low == cp.obit[t]);
            pr[0] = pk ? (tno[t] ? T(0) : fdo[t]) : pr[0];
            pr[1] = pk ? reso[t] : pr[1];
            wto[t] = pk ? tv_last : wto[t];
        }
        selm |= need1 ? low : 0u;
        gsum_all<LG>(pr);                                      // one lane contributes: exact
        // x -= Gamma f^T (f x - target value) / (f Gamma f^T)  =  beta (...) / (f beta)
        const T coef = (need1 && pr[0] != T(0)) ? num<T>::div(pr[1], pr[0]) : T(0);
#pragma unroll
        for (int sl = 0; sl < S; ++sl) Ul[sl] = num<T>::fma(-bl[sl], coef, Ul[sl]);
    }
    ATACOM_MARK("C3_asm");
    // ---- assembly.  The equality row once more, exactly (rounding only): a u = -y_0, a U = 0 -- on the distributed
    // coordinates: two group sums, the correction applied to the own coordinates
    if constexpr (NF == 1) {
        T pa[2] = {T(0), T(0)};
#pragma unroll
        for (int sl = 0; sl < S6; ++sl) {
            pa[0] = num<T>::fma(cp.At[sl][0], xl[sl], pa[0]);
            pa[1] = num<T>::fma(cp.At[sl][0], Ul[sl], pa[1]);
        }
        gsum_all<LG>(pa);
        const T au = (pa[0] + y[0]) * cp.iaa, aU = pa[1] * cp.iaa;
#pragma unroll
        for (int sl = 0; sl < S6; ++sl) {
            xl[sl] = num<T>::fma(-cp.At[sl][0], au, xl[sl]);
            Ul[sl] = num<T>::fma(-cp.At[sl][0], aU, Ul[sl]);
        }
    }
    // the coordinates, replicated
    T xa[N1], Ua[N1];
    static_for<0, N1>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        xa[i] = qbcast<i % LG, LG>(xl[i / LG]);
        Ua[i] = qbcast<i % LG, LG>(Ul[i / LG]);
    });
#pragma unroll
    for (int i = 0; i < NQ; ++i) mu[i] = xa[i] + Ua[i];
    // the slack velocities of the lane's own rows, then one broadcast per row:  w_g = -(y_g + A_g u) / s_g with the true
    // slack; a free slack coordinate takes its target itself; the coordinate slack is a component of the state
    T wo[RS];
#pragma unroll
    for (int t = 0; t < RS; ++t) {
        T wm = yo[t], wa = T(0);
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            if (slot_zero(t, i)) continue;
            wm = num<T>::fma(cp.Ao[t][i], xa[i], wm);
            wa = num<T>::fma(cp.Ao[t][i], Ua[i], wa);
        }
        const bool has = cp.obit[t] != 0u;
        const T inv_s = (has && (num<T>::abs(so[t]) >= CC::TINY * cp.ao[t])) ? num<T>::rcp(so[t]) : T(0);
        const bool selo = (selm & cp.obit[t]) != 0u;
        const T w = selo ? num<T>::fma(-wm, inv_s, wto[t]) : -(wm + wa) * inv_s;
        wo[t] = ((pmask & cp.obit[t]) != 0u) ? xa[NQ] + Ua[NQ] : w;
    }
    static_for<0, NG>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        mu[NQ + g] = qbcast<g % LG, LG>(wo[g / LG]);
    });
}

}  // namespace atacom
