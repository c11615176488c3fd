// Per-environment small dense linear algebra for the ATACOM tangent-space map, one environment per
// lane, every matrix held in VGPRs (all loops have compile-time trip counts and are fully unrolled so
// that no array is ever indexed dynamically).
//
// What it replaces in the reference (float64 numpy/scipy on the host, one call per env per sub-step):
//   * pinv_null  -- /root/reference/atacom/utils/null_space_coordinate.py:8-26  (LAPACK dgesdd SVD)
//   * rref       -- /root/reference/atacom/utils/null_space_coordinate.py:40-79 (tol = 0.05,
//                   /root/reference/atacom/atacom.py:128)
//
// Algorithm (DESIGN.md "Null-space numerics"): the reference's orthonormal null basis is whatever
// LAPACK's dgesdd returns, and the tolerance test inside rref looks at basis-dependent entries, so to
// reproduce the reference in the chart-switching regime the kernel has to produce the *same* basis.
// For an M x N matrix with M < N < 11M/6 dgesdd bidiagonalises A = Q B P^T with Householder
// reflectors (dgebd2/dlarfg) and returns vh[M:] = last N-M columns of P = G(1)...G(M).  We therefore
// run the same Golub-Kahan Householder bidiagonalisation per lane -- no SVD iteration is needed: the
// null basis is P[:, M:], and the pseudo-inverse solve is  x = P [B^{-1} Q^T r ; 0]  with B lower
// bidiagonal.  MFMA is deliberately not used: every lane owns a different 12x17 matrix and the work
// is a dependent chain of rank-1 updates, not a shared-operand contraction.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

// instruction-count profiling aid: -DATACOM_MARKS puts named comment lines into the ISA (and pins the schedule at
// them); build/isa tooling splits the kernel body at the marks.  Off in the product build.
#ifdef ATACOM_MARKS
#define ATACOM_MARK(name) do { __builtin_amdgcn_sched_barrier(0); asm volatile("; MARK " name); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define ATACOM_MARK(name) ((void)0)
#endif

// phase boundary of the rigid-body sub-step: the scheduler may not move instructions across it.  Left to itself it
// interleaves the chain, the Newton-Euler pass and the composite-inertia pass for instruction-level parallelism, and the
// union of their live ranges no longer fits the register file next to the held solver state (-DATACOM_NO_PHASE: A/B build)
#ifdef ATACOM_NO_PHASE
#define ATACOM_PHASE() ((void)0)
#else
#define ATACOM_PHASE() __builtin_amdgcn_sched_barrier(0)
#endif

namespace atacom {

template <typename T> struct num;
// float: v_rcp_f32 / v_sqrt_f32 (1 ulp) instead of the ~10-instruction IEEE division / sqrt expansions -- the
// step is a latency-bound dependent chain and 1-2 ulp is far inside the stated float32 parity tolerance.
// double (parity build): correctly rounded operations.
template <> struct num<float> {
#ifdef ATACOM_IEEE_DIV
    static __device__ __forceinline__ float sqrt(float x) { return __builtin_sqrtf(x); }
    static __device__ __forceinline__ float rcp(float x) { return 1.0f / x; }
    static __device__ __forceinline__ float div(float a, float b) { return a / b; }
#else
    static __device__ __forceinline__ float sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
    static __device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
    static __device__ __forceinline__ float div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
#endif
    static __device__ __forceinline__ float abs(float x) { return __builtin_fabsf(x); }
    static __device__ __forceinline__ float copysign(float m, float s) { return __builtin_copysignf(m, s); }
    static __device__ __forceinline__ float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
    static __device__ __forceinline__ float exp(float x) { return expf(x); }
    static __device__ __forceinline__ float log(float x) { return logf(x); }
    // tanh(x) = sign(x) (1 - 2 / (exp(2|x|) + 1)) on v_exp_f32 / v_rcp_f32: branch-free, |err| < 3e-7 (libm's tanhf
    // inlines ~100 instructions with divergent range branches -- per hidden unit)
    static __device__ __forceinline__ float tanh(float x) {
        const float e = __builtin_amdgcn_exp2f(__builtin_fabsf(x) * 2.885390081777927f);
        return __builtin_copysignf(1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f), x);
    }
#ifdef ATACOM_LIBM_TRIG
    static __device__ __forceinline__ void sincos(float x, float* s, float* c) { sincosf(x, s, c); }
#else
    // Joint angles are O(1) rad, so the library's Payne-Hanek large-argument path (~150 instructions per call,
    // 11 calls per env step = 7 % of the step) is dead weight.  Two-term Cody-Waite reduction by pi/2 with FMA,
    // then the minimax polynomials of the classic single-precision kernels on [-pi/4, pi/4]; absolute error
    // < 1e-7 for |x| <= 1e3 (checked against float64; kinematics parity: test_constraint_terms_against_oracle), ~25 instructions.
    static __device__ __forceinline__ void sincos(float x, float* s, float* c) {
        const float k = __builtin_rintf(x * 0.6366197723675814f);
        float r = __builtin_fmaf(k, -1.5707963705062866f, x);
        r = __builtin_fmaf(k, 4.371139000186241e-08f, r);
        const int q = (int)k;
        const float r2 = r * r;
        const float ps = __builtin_fmaf(__builtin_fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f);
        const float sn = __builtin_fmaf(ps * r2, r, r);
        const float pc = __builtin_fmaf(__builtin_fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2,
                                        4.166664568298827e-2f);
        const float cs = __builtin_fmaf(pc * r2, r2, __builtin_fmaf(r2, -0.5f, 1.0f));
        const bool sw = (q & 1) != 0;
        const float so = sw ? cs : sn, co = sw ? sn : cs;
        *s = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, so) ^ ((unsigned)(q & 2) << 30));
        *c = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, co) ^ ((unsigned)((q + 1) & 2) << 30));
    }
#endif
    static __device__ __forceinline__ float acos(float x) { return acosf(x); }
    static __device__ __forceinline__ float max(float a, float b) { return fmaxf(a, b); }
    static __device__ __forceinline__ float min(float a, float b) { return fminf(a, b); }
    // min(max(x, lo), hi) for lo <= hi as ONE v_med3_f32.  fminf / fmaxf are lowered with a canonicalising
    // v_max_f32 x, x, x in front of every operand the compiler cannot prove quiet (IEEE mode): 4-5 instructions per
    // clamp in the sub-step loop.  A NaN x gives lo with both forms (v_med3 falls back to min3, which ignores NaNs).
    static __device__ __forceinline__ float clamp(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
};
template <> struct num<double> {
    static __device__ __forceinline__ double sqrt(double x) { return __builtin_sqrt(x); }
    static __device__ __forceinline__ double rcp(double x) { return 1.0 / x; }
    static __device__ __forceinline__ double div(double a, double b) { return a / b; }
    static __device__ __forceinline__ double abs(double x) { return __builtin_fabs(x); }
    static __device__ __forceinline__ double copysign(double m, double s) { return __builtin_copysign(m, s); }
    static __device__ __forceinline__ double fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
    static __device__ __forceinline__ double exp(double x) { return ::exp(x); }
    static __device__ __forceinline__ double log(double x) { return ::log(x); }
    static __device__ __forceinline__ double tanh(double x) { return ::tanh(x); }
    static __device__ __forceinline__ void sincos(double x, double* s, double* c) { ::sincos(x, s, c); }
    static __device__ __forceinline__ double acos(double x) { return ::acos(x); }
    static __device__ __forceinline__ double max(double a, double b) { return fmax(a, b); }
    static __device__ __forceinline__ double min(double a, double b) { return fmin(a, b); }
    static __device__ __forceinline__ double clamp(double x, double lo, double hi) { return fmin(fmax(x, lo), hi); }
};

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

// ---- two-wide register vectors: on gfx950 arithmetic on them is one packed instruction (v_pk_fma_f32,
// v_pk_mul_f32, v_pk_add_f32; op_sel broadcasts a scalar operand for free).  The solver below keeps every
// per-row quantity as ROW PAIRS (rows 2p, 2p+1 of the same column slot in one register pair) so that all of
// its multiply-adds are packed by construction -- the auto-vectoriser found only ~2/3 of them.  (double has no
// packed form; the same code then simply compiles to two scalar operations.)
template <typename T> using vec2 = T __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ vec2<T> splat2(T v) { return vec2<T>{v, v}; }
template <typename T> __device__ __forceinline__ vec2<T> fma2(vec2<T> a, vec2<T> b, vec2<T> c) {
    return __builtin_elementwise_fma(a, b, c);
}

// Householder reflector of LAPACK's dlarfg on (alpha, x[0..L)) given ss = sum x^2:  H = I - tau [1;v][1;v]^T,
// H [alpha;x] = [beta;0],  beta = -sign(alpha) ||[alpha;x]||.  Returns the scale 1/(alpha-beta) to apply to x, writes
// beta and tau.
// Two departures from dlarfg's special cases, both branch-free (the old form spent 5 selects per reflector on them):
//   * x == 0, alpha != 0: dlarfg returns H = I (tau = 0, beta = alpha); this form returns the reflection through
//     alpha's axis (tau = 2, beta = -alpha, v = 0).  The two differ by the sign of ONE vector of the row-space basis
//     (P -> P D_i, Q -> Q D_i with D_i = diag(.., -1, ..)), which cancels in  x = P B^-1 Q^T y  and never reaches the
//     null basis P[:, M:] (column i < M of P is not a null vector) -- DESIGN.md section 3;
//   * the all-zero vector (rank-deficient matrix): tau = 0, v = 0, beta = -+TINY, i.e. H = I with a pivot the guarded
//     bidiagonal solve then treats as zero -- no NaN / Inf is ever produced here.
template <typename T> struct tiny_of;
template <> struct tiny_of<float> { static constexpr float value = 1e-30f; };
template <> struct tiny_of<double> { static constexpr double value = 1e-280; };
template <typename T>
__device__ __forceinline__ T larfg_scale(T alpha, T ss, T& beta, T& tau) {
    const T nrm = num<T>::sqrt(num<T>::fma(alpha, alpha, ss));
    const T b = num<T>::copysign(num<T>::max(nrm, tiny_of<T>::value), -alpha);   // = -sign(alpha) max(nrm, TINY)
    const T dm = b - alpha;                                                       // |dm| = |b| + |alpha| >= TINY
    beta = b;
    const T t = dm * num<T>::rcp(b);
    tau = (nrm > T(0)) ? t : T(0);
    return -num<T>::rcp(dm);
}

// z = B^+ y for the lower bidiagonal B (d on the diagonal, e below it) by forward substitution, with the reference's
// rank handling restated for the bidiagonal form: pinv_null drops singular values below eps * max(M, N) * sigma_max
// (null_space_coordinate.py:12-21); here a pivot |d_i| <= 8 eps (M + 5) max|d| is treated as zero -- its component of the
// solution is dropped (z_i = 0) instead of dividing by it.  For a full-rank matrix this changes nothing; for a
// rank-deficient one it keeps every downstream value finite (DESIGN.md "Rank handling").
template <typename T> struct eps_of;
template <> struct eps_of<float> { static constexpr float value = 1.1920929e-7f; };
template <> struct eps_of<double> { static constexpr double value = 2.220446049250313e-16; };
template <typename T, int M, typename YG>
__device__ __forceinline__ void bidiag_forward_solve(const T (&d)[M], const T (&e)[M], YG&& yat, T (&z)[M]) {
    T dmax = num<T>::abs(d[0]);
#pragma unroll
    for (int i = 1; i < M; ++i) dmax = num<T>::max(dmax, num<T>::abs(d[i]));
    const T dtol = dmax * (eps_of<T>::value * T(8 * (M + 5)));  // N <= M + 5 for every shape of this library; x8: a
                                                                // duplicated row leaves a pivot of a few eps, not 0
#pragma unroll
    for (int i = 0; i < M; ++i) {
        const T rd = (num<T>::abs(d[i]) > dtol) ? num<T>::rcp(d[i]) : T(0);
        const T rhs = (i == 0) ? yat(0) : num<T>::fma(-e[i > 0 ? i - 1 : 0], z[i > 0 ? i - 1 : 0], yat(i));
        z[i] = rhs * rd;
    }
}

// A: M x N (row i, col j), full row rank; y: right-hand side (length M) -- both handed over as generators
// aget(row, col) / yget(row) called with compile-time indices (std::integral_constant), so the operands are born
// in their registers (an intermediate T a[M][N] array invites the optimiser to merge stores into it and then fail
// to dissolve it, which lands it in scratch memory).
// On return  x = A^+ y  (length N)  and  nb = orthonormal null basis (N x K, K = N - M), equal to
// scipy.linalg.svd(A, full_matrices=True)[2][M:].T up to rounding.
// Every per-row quantity is kept as ROW PAIRS (vec2: rows 2p, 2p+1 of one column) and the K null vectors + x as
// pairs over the vector index, so all multiply-adds are packed (v_pk_fma_f32) by construction.
// PRE0 (see the lane-group solver in atacom_quad.h for the idea): the caller has generated the first right reflector
// G(0) and applied it to the rows below -- it is the same for all physics sub-steps of a step when row 0 is a slack-free
// equality row and q, dq are held.  Row 0 of the matrix handed over holds the reflector vector (columns 1 .. NV0-1;
// structurally zero from NV0 on), pre_d0 / pre_tau0 are its beta / tau.
template <typename T, int M, int N, bool PRE0 = false, int NV0 = N, typename AF, typename YF>
__device__ __forceinline__ void bidiag_solve_null(AF&& aget, YF&& yget, T (&x)[N], T (&nb)[N][N - M],
                                                  const T pre_d0 = T(0), const T pre_tau0 = T(0)) {
    constexpr int K = N - M;
    constexpr int MP = (M + 1) / 2;          // row pairs (a zero row pads an odd M: it is a fixed point of every step)
    constexpr int KP = (K + 2) / 2;          // pairs over [nb_0 .. nb_{K-1}, x]
    using V2 = vec2<T>;
    V2 a2[MP][N], y2[MP];
    static_for<0, MP>([&](auto pc) {
        constexpr int p = decltype(pc)::value;
        constexpr int r0 = 2 * p, r1 = (2 * p + 1 < M) ? 2 * p + 1 : 2 * p;
        constexpr bool two = 2 * p + 1 < M;
        const T ya = yget(std::integral_constant<int, r0>{});
        const T yb = two ? yget(std::integral_constant<int, r1>{}) : T(0);
        y2[p] = V2{ya, yb};
        static_for<0, N>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            const T va = aget(std::integral_constant<int, r0>{}, cc);
            const T vb = two ? aget(std::integral_constant<int, r1>{}, cc) : T(0);
            a2[p][c] = V2{va, vb};
        });
    });
    T d[M], e[M], taup[M];
    static_for<0, M>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int pi = i / 2, hi = i % 2;      // row i is half hi of row pair pi
        __builtin_amdgcn_sched_barrier(0);         // do not overlap reflector steps: keeps the live set near M*N
        // ---- right reflector G(i): annihilate A[i][i+1..N); v (v[i] = 1 implicit) stays in row i
        constexpr bool pre = PRE0 && i == 0;       // G(0) came with the matrix
        T v[N];
        T tp = pre_tau0;
        if constexpr (pre) { d[0] = pre_d0; taup[0] = pre_tau0; }
        else {
        T ss = T(0);
#pragma unroll
        for (int c = i + 1; c < N; ++c) ss = num<T>::fma(a2[pi][c][hi], a2[pi][c][hi], ss);
        T beta;
        const T sc = larfg_scale(a2[pi][i][hi], ss, beta, tp);
        d[i] = beta;
        taup[i] = tp;
#pragma unroll
        for (int c = i + 1; c < N; ++c) { v[c] = a2[pi][c][hi] * sc; a2[pi][c][hi] = v[c]; }
        }
        if constexpr (i < M - 1) {
            constexpr int p0 = (i + 1) / 2;        // first row pair holding a row > i
            if constexpr (!pre) {
#pragma unroll
            for (int p = p0; p < MP; ++p) {
                V2 w = a2[p][i];
#pragma unroll
                for (int c = i + 1; c < N; ++c) w = fma2(a2[p][c], splat2(v[c]), w);
                w *= splat2(tp);
                if (2 * p <= i) w.x = T(0);        // the pair's first row is row i itself: leave it alone
                a2[p][i] -= w;
#pragma unroll
                for (int c = i + 1; c < N; ++c) a2[p][c] = fma2(-w, splat2(v[c]), a2[p][c]);
            }
            }   // !pre
            // ---- left reflector H(i): annihilate A[i+2..M)[i]; u over the pairs p0..: 0 for rows <= i, 1 at row
            // i+1, scaled column entries below (not kept: Q is applied to y on the fly)
            V2 sq = splat2(T(0));
#pragma unroll
            for (int p = p0 + 1; p < MP; ++p) sq = fma2(a2[p][i], a2[p][i], sq);
            T su = sq.x + sq.y;
            if constexpr (hi == 1) su = num<T>::fma(a2[p0][i].y, a2[p0][i].y, su);     // row i+2 shares i+1's pair
            T betaq, tq;
            const T scq = larfg_scale(hi == 0 ? a2[p0][i].y : a2[p0][i].x, su, betaq, tq);
            e[i] = betaq;
            V2 u2[MP];
#pragma unroll
            for (int p = p0; p < MP; ++p) u2[p] = a2[p][i] * splat2(scq);
            if constexpr (hi == 0) u2[p0] = V2{T(0), T(1)};
            else u2[p0].x = T(1);
#pragma unroll
            for (int c = i + 1; c < N; ++c) {
                V2 acc = u2[p0] * a2[p0][c];
#pragma unroll
                for (int p = p0 + 1; p < MP; ++p) acc = fma2(u2[p], a2[p][c], acc);
                const T w = (acc.x + acc.y) * tq;
#pragma unroll
                for (int p = p0; p < MP; ++p) a2[p][c] = fma2(splat2(-w), u2[p], a2[p][c]);
            }
            {   // the same H(i) on the right-hand side: y <- Q^T y
                V2 acc = u2[p0] * y2[p0];
#pragma unroll
                for (int p = p0 + 1; p < MP; ++p) acc = fma2(u2[p], y2[p], acc);
                const T w = (acc.x + acc.y) * tq;
#pragma unroll
                for (int p = p0; p < MP; ++p) y2[p] = fma2(splat2(-w), u2[p], y2[p]);
            }
        }
    });
    // ---- z = B^{-1} (Q^T y), B lower bidiagonal (d on the diagonal, e below it)
    T z[M];
    bidiag_forward_solve<T, M>(d, e, [&](int i) { return y2[i / 2][i % 2]; }, z);
    V2 nx[N][KP];
#pragma unroll
    for (int c = 0; c < N; ++c) {
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            T h[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int k = 2 * j + t;
                h[t] = (k < K) ? ((c == M + k) ? T(1) : T(0)) : ((k == K && c < M) ? z[c < M ? c : 0] : T(0));
            }
            nx[c][j] = V2{h[0], h[1]};
        }
    }
    // ---- [nb | x] <- G(1) ... G(M) [nb | x]
    static_for<0, M>([&](auto kc) {
        constexpr int i = M - 1 - decltype(kc)::value;
        __builtin_amdgcn_sched_barrier(0);
        constexpr int CE = (PRE0 && i == 0) ? NV0 : N;      // the hoisted G(0) is zero from column NV0 on
        T v[N];
#pragma unroll
        for (int c = i + 1; c < CE; ++c) v[c] = a2[i / 2][c][i % 2];
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            V2 w = nx[i][j];
#pragma unroll
            for (int c = i + 1; c < CE; ++c) w = fma2(splat2(v[c]), nx[c][j], w);
            w *= splat2(taup[i]);
            nx[i][j] -= w;
#pragma unroll
            for (int c = i + 1; c < CE; ++c) nx[c][j] = fma2(-w, splat2(v[c]), nx[c][j]);
        }
    });
#pragma unroll
    for (int c = 0; c < N; ++c) {
        x[c] = nx[c][K / 2][K % 2];
#pragma unroll
        for (int k = 0; k < K; ++k) nb[c][k] = nx[c][k / 2][k % 2];
    }
}

// "Chart" of the null basis: (Nc @ alpha) with Nc = rref(nb, row_vectors=False, tol) as the reference computes it
// (null_space_coordinate.py:40-79 called from atacom.py:128,131) -- WITHOUT running Gauss-Jordan over all N columns.
//
// The reference eliminates on V = nb^T (K x N), column by column: the best remaining pivot p of column j is tested
// against tol; p <= tol zeroes the column in the unused rows and moves on, otherwise the pivot row is scaled and
// eliminated from every row, over all columns >= j.  Observations that make this cheap:
//   * row operations act on every column independently, so column j of the running matrix is  T V[:, j]  with T
//     the K x K product of the elimination steps taken so far (a rank-1 update per pivot);
//   * a pivot column ends as a unit vector:            (Nc alpha)[j] = alpha[#pivots before it];
//   * a skipped column is frozen at the moment it is skipped (its entries in the not-yet-used rows are zeroed, so
//     no later pivot row can change it):               (Nc alpha)[j] = sum over USED rows of alpha~[r] v[r];
//   * a column reached after the K-th pivot is just  T_final V[:, j]  -- the same formula with every row used.
// So one uniform step per column -- v = T V[:, j] (K^2 FMA), arg-max over unused rows, one dot product -- plus a
// K^2 rank-1 update of T on the (at most K) columns that pivot, guarded by a wave-uniform ballot.  ~1.4 k
// instructions for 5 x 17 instead of ~4.3 k for the literal Gauss-Jordan, identical results in exact arithmetic
// (the tolerance test sees the same values: the candidates of column j are T V[:, j] restricted to unused rows).
template <typename T, int K>
struct ChartState {
    static constexpr int KP = (K + 1) / 2;   // rows of T as pairs (rows 2j, 2j+1; a zero row pads an odd K): the three
                                             // K x K loops -- T col, the pivot row pick, the rank-1 update -- are packed
    vec2<T> Tm2[KP][K];   // accumulated elimination steps
    T at[K];         // alpha~[r] = alpha[order of row r] for used rows, 0 for unused rows
    bool used[K];
    int cnt;
    T arem[K];       // action components not yet paired with a pivot row: arem[0] = alpha[cnt].  (A shifting
                     // queue, not alpha[cnt]: the optimiser turns a select chain over alpha[k] into a dynamically
                     // indexed load, which forces the array into scratch memory -- one global round trip per column.)
    __device__ __forceinline__ void init(const T (&alpha)[K]) {
#pragma unroll
        for (int j = 0; j < KP; ++j)
#pragma unroll
            for (int c = 0; c < K; ++c) Tm2[j][c] = vec2<T>{(2 * j == c) ? T(1) : T(0), (2 * j + 1 == c) ? T(1) : T(0)};
#pragma unroll
        for (int r = 0; r < K; ++r) { at[r] = T(0); used[r] = false; }
        cnt = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) arem[k] = alpha[k];
    }
    // examine one column (raw entries col[K]); returns its (Nc alpha) value
    __device__ __forceinline__ T examine(const T (&col)[K], const T (&alpha)[K], T tol) {
        using V2 = vec2<T>;
        V2 v2[KP];
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            V2 acc = Tm2[j][0] * splat2(col[0]);
#pragma unroll
            for (int c = 1; c < K; ++c) acc = fma2(Tm2[j][c], splat2(col[c]), acc);
            v2[j] = acc;
        }
        T p = T(-1), pv = T(1), dotv = T(0);
        int kk = 0;
#pragma unroll
        for (int r = 0; r < K; ++r) {
            const T vr = v2[r / 2][r % 2];
            const T av = used[r] ? T(-1) : num<T>::abs(vr);
            const bool gt = av > p;                              // strict: first maximum, like np.argmax
            p = gt ? av : p;
            kk = gt ? r : kk;
            pv = gt ? vr : pv;
            dotv = num<T>::fma(at[r], vr, dotv);
        }
        const bool piv = (cnt < K) && (p > tol);
        const T anext = arem[0];
        const T result = piv ? anext : dotv;
        if (__builtin_amdgcn_ballot_w64(piv) != 0ull) {           // wave-uniform: only ~K columns ever pivot
            const T invp = piv ? num<T>::rcp(pv) : T(0);         // lanes that do not pivot apply the identity
            // one-hot of the pivot row, scaled by 1 / pivot (all zero when this lane does not pivot), and the
            // update factors d = v - e_pivot (0 for the pad row)
            V2 ohi2[KP], d2[KP];
#pragma unroll
            for (int j = 0; j < KP; ++j) {
                T oh[2], dd[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int r = 2 * j + i;
                    const bool is_p = (r < K) && piv && (r == kk);
                    oh[i] = is_p ? invp : T(0);
                    dd[i] = (r < K) ? (is_p ? v2[j][i] - T(1) : v2[j][i]) : T(0);
                    if (r < K) {
                        at[r < K ? r : 0] = is_p ? anext : at[r < K ? r : 0];
                        used[r < K ? r : 0] = used[r < K ? r : 0] || is_p;
                    }
                }
                ohi2[j] = V2{oh[0], oh[1]};
                d2[j] = V2{dd[0], dd[1]};
            }
            T rk[K];
#pragma unroll
            for (int c = 0; c < K; ++c) {
                V2 t = ohi2[0] * Tm2[0][c];
#pragma unroll
                for (int j = 1; j < KP; ++j) t = fma2(ohi2[j], Tm2[j][c], t);
                rk[c] = t.x + t.y;
            }
#pragma unroll
            for (int j = 0; j < KP; ++j)
#pragma unroll
                for (int c = 0; c < K; ++c) Tm2[j][c] = fma2(-d2[j], splat2(rk[c]), Tm2[j][c]);
            cnt += piv ? 1 : 0;
#pragma unroll
            for (int k = 0; k + 1 < K; ++k) arem[k] = piv ? arem[k + 1] : arem[k];
        }
        return result;
    }
};

template <typename T, int N, int K>
__device__ __forceinline__ void rref_apply(T (&nb)[N][K], const T (&alpha)[K], T tol, T (&out)[N]) {
    ChartState<T, K> st;
    st.init(alpha);
#pragma unroll
    for (int j = 0; j < N; ++j) out[j] = st.examine(nb[j], alpha, tol);
}

}  // namespace atacom
