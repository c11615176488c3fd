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
    static __device__ __forceinline__ float max(float a, float b) { return fmaxf(a, b); }
    static __device__ __forceinline__ float min(float a, float b) { return fminf(a, b); }
};
template <> struct num<double> {
    static __device__ __forceinline__ double sqrt(double x) { return __builtin_sqrt(x); }
    static __device__ __forceinline__ double rcp(double x) { return 1.0 / x; }
    static __device__ __forceinline__ double div(double a, double b) { return a / b; }
    static __device__ __forceinline__ double abs(double x) { return __builtin_fabs(x); }
    static __device__ __forceinline__ double copysign(double m, double s) { return __builtin_copysign(m, s); }
    static __device__ __forceinline__ double fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
    static __device__ __forceinline__ double exp(double x) { return ::exp(x); }
    static __device__ __forceinline__ void sincos(double x, double* s, double* c) { ::sincos(x, s, c); }
    static __device__ __forceinline__ double max(double a, double b) { return fmax(a, b); }
    static __device__ __forceinline__ double min(double a, double b) { return fmin(a, b); }
};

// LAPACK dlarfg on (alpha, x[0..L)) given ss = sum x^2: H = I - tau [1;v][1;v]^T, H [alpha;x] = [beta;0].
// Returns the scale 1/(alpha-beta) to apply to x (0 when x == 0, i.e. H = I), writes beta and tau.
template <typename T>
__device__ __forceinline__ T larfg_scale(T alpha, T ss, T& beta, T& tau) {
    const bool nz = ss != T(0);
    const T nrm = num<T>::sqrt(num<T>::fma(alpha, alpha, ss));
    const T b = -num<T>::copysign(nrm, alpha);
    beta = nz ? b : alpha;
    const T safe_b = nz ? b : T(1);
    tau = nz ? num<T>::div(b - alpha, safe_b) : T(0);
    const T den = nz ? (alpha - b) : T(1);
    return nz ? num<T>::rcp(den) : T(0);
}

// a: M x N (row i, col j), full row rank; y: right-hand side (length M).
// On return  x = a^+ y  (length N)  and  nb = orthonormal null basis (N x K, K = N - M), equal to
// scipy.linalg.svd(a, full_matrices=True)[2][M:].T up to rounding.  a and y are destroyed.
template <typename T, int M, int N>
__device__ __forceinline__ void bidiag_solve_null(T (&a)[M][N], T (&y)[M], T (&x)[N], T (&nb)[N][N - M]) {
    constexpr int K = N - M;
    T d[M], e[M], taup[M];
#pragma unroll
    for (int i = 0; i < M; ++i) {
        __builtin_amdgcn_sched_barrier(0);     // do not overlap reflector steps: keeps the live set near M*N
        // ---- right reflector G(i): annihilate a[i][i+1..N)
        T ss = T(0);
#pragma unroll
        for (int c = i + 1; c < N; ++c) ss = num<T>::fma(a[i][c], a[i][c], ss);
        T beta, tp;
        const T sc = larfg_scale(a[i][i], ss, beta, tp);
        d[i] = beta;
        taup[i] = tp;
#pragma unroll
        for (int c = i + 1; c < N; ++c) a[i][c] *= sc;          // v (v[i] = 1 implicit) kept in place
        if (i < M - 1) {
            // apply G(i) from the right to rows i+1..M-1
#pragma unroll
            for (int r = i + 1; r < M; ++r) {
                T w = a[r][i];
#pragma unroll
                for (int c = i + 1; c < N; ++c) w = num<T>::fma(a[r][c], a[i][c], w);
                w *= tp;
                a[r][i] -= w;
#pragma unroll
                for (int c = i + 1; c < N; ++c) a[r][c] = num<T>::fma(-w, a[i][c], a[r][c]);
            }
            // ---- left reflector H(i): annihilate a[i+2..M)[i]
            T su = T(0);
#pragma unroll
            for (int r = i + 2; r < M; ++r) su = num<T>::fma(a[r][i], a[r][i], su);
            T betaq, tq;
            const T scq = larfg_scale(a[i + 1][i], su, betaq, tq);
            e[i] = betaq;
#pragma unroll
            for (int r = i + 2; r < M; ++r) a[r][i] *= scq;      // u (u[i+1] = 1 implicit)
#pragma unroll
            for (int c = i + 1; c < N; ++c) {
                T w = a[i + 1][c];
#pragma unroll
                for (int r = i + 2; r < M; ++r) w = num<T>::fma(a[r][i], a[r][c], w);
                w *= tq;
                a[i + 1][c] -= w;
#pragma unroll
                for (int r = i + 2; r < M; ++r) a[r][c] = num<T>::fma(-w, a[r][i], a[r][c]);
            }
            {   // the same H(i) on the right-hand side: y <- Q^T y
                T w = y[i + 1];
#pragma unroll
                for (int r = i + 2; r < M; ++r) w = num<T>::fma(a[r][i], y[r], w);
                w *= tq;
                y[i + 1] -= w;
#pragma unroll
                for (int r = i + 2; r < M; ++r) y[r] = num<T>::fma(-w, a[r][i], y[r]);
            }
        }
    }
    // ---- z = B^{-1} (Q^T y), B lower bidiagonal (d on the diagonal, e below it)
    x[0] = num<T>::div(y[0], d[0]);
#pragma unroll
    for (int i = 1; i < M; ++i) x[i] = num<T>::div(num<T>::fma(-e[i - 1], x[i - 1], y[i]), d[i]);
#pragma unroll
    for (int c = M; c < N; ++c) x[c] = T(0);
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
        for (int k = 0; k < K; ++k) nb[r][k] = (r == M + k) ? T(1) : T(0);
    // ---- [x | nb] <- G(1) ... G(M) [x | nb]
#pragma unroll
    for (int i = M - 1; i >= 0; --i) {
        __builtin_amdgcn_sched_barrier(0);
        {
            T w = x[i];
#pragma unroll
            for (int c = i + 1; c < N; ++c) w = num<T>::fma(a[i][c], x[c], w);
            w *= taup[i];
            x[i] -= w;
#pragma unroll
            for (int c = i + 1; c < N; ++c) x[c] = num<T>::fma(-w, a[i][c], x[c]);
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            T w = nb[i][k];
#pragma unroll
            for (int c = i + 1; c < N; ++c) w = num<T>::fma(a[i][c], nb[c][k], w);
            w *= taup[i];
            nb[i][k] -= w;
#pragma unroll
            for (int c = i + 1; c < N; ++c) nb[c][k] = num<T>::fma(-w, a[i][c], nb[c][k]);
        }
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
    T Tm[K][K];      // accumulated elimination steps
    T at[K];         // alpha~[r] = alpha[order of row r] for used rows, 0 for unused rows
    bool used[K];
    int cnt;
    T arem[K];       // action components not yet paired with a pivot row: arem[0] = alpha[cnt].  (A shifting
                     // queue, not alpha[cnt]: the optimiser turns a select chain over alpha[k] into a dynamically
                     // indexed load, which forces the array into scratch memory -- one global round trip per column.)
    __device__ __forceinline__ void init(const T (&alpha)[K]) {
#pragma unroll
        for (int r = 0; r < K; ++r) {
#pragma unroll
            for (int c = 0; c < K; ++c) Tm[r][c] = (r == c) ? T(1) : T(0);
            at[r] = T(0);
            used[r] = false;
        }
        cnt = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) arem[k] = alpha[k];
    }
    // examine one column (raw entries col[K]); returns its (Nc alpha) value
    __device__ __forceinline__ T examine(const T (&col)[K], const T (&alpha)[K], T tol) {
        T v[K];
#pragma unroll
        for (int r = 0; r < K; ++r) {
            T acc = Tm[r][0] * col[0];
#pragma unroll
            for (int c = 1; c < K; ++c) acc = num<T>::fma(Tm[r][c], col[c], acc);
            v[r] = acc;
        }
        T p = T(-1), pv = T(1), dotv = T(0);
        int kk = 0;
#pragma unroll
        for (int r = 0; r < K; ++r) {
            const T av = used[r] ? T(-1) : num<T>::abs(v[r]);
            const bool gt = av > p;                              // strict: first maximum, like np.argmax
            p = gt ? av : p;
            kk = gt ? r : kk;
            pv = gt ? v[r] : pv;
            dotv = num<T>::fma(at[r], v[r], dotv);
        }
        const bool piv = (cnt < K) && (p > tol);
        const T anext = arem[0];
        const T result = piv ? anext : dotv;
        if (__builtin_amdgcn_ballot_w64(piv) != 0ull) {           // wave-uniform: only ~K columns ever pivot
            const T invp = piv ? num<T>::rcp(pv) : T(0);         // lanes that do not pivot apply the identity
            T rk[K], d[K];
#pragma unroll
            for (int c = 0; c < K; ++c) {
                T t = T(0);
#pragma unroll
                for (int r = 0; r < K; ++r) t = (r == kk) ? Tm[r][c] : t;
                rk[c] = t * invp;
            }
#pragma unroll
            for (int r = 0; r < K; ++r) {
                const bool is_p = piv && (r == kk);
                d[r] = is_p ? v[r] - T(1) : v[r];
                at[r] = is_p ? anext : at[r];
                used[r] = used[r] || is_p;
            }
#pragma unroll
            for (int r = 0; r < K; ++r)
#pragma unroll
                for (int c = 0; c < K; ++c) Tm[r][c] = num<T>::fma(-d[r], rk[c], Tm[r][c]);
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
