"""Null-space numerics of ATACOM, float64 (oracle; test infrastructure only).

Restates /root/reference/atacom/utils/null_space_coordinate.py:
  * ``pinv_null``  (lines 8-26)  -- SVD pseudo-inverse + orthonormal null basis
  * ``rref``       (lines 40-79) -- Gauss-Jordan with a pivot tolerance ("chart" of the null space)

and adds the closed-form description of *which* orthonormal null basis LAPACK returns, which the
HIP kernels reproduce without an SVD:

  scipy.linalg.svd(a, full_matrices=True) -> LAPACK dgesdd.  For an M x N input with
  M < N < 11*M/6 (6x9 and 12x17 here) dgesdd takes its "path 5t": dgebrd reduces A to lower
  bidiagonal form  A = Q B P^T  with Householder reflectors H(i) (left) and G(i) (right), the SVD
  of B is computed, and VT = blkdiag(VT_B, I_{N-M}) * P^T.  Hence, for a full-rank A, the rows
  ``vh[M:]`` -- the null basis the reference uses -- are exactly the last N-M columns of
  P = G(1) G(2) ... G(M), independent of the singular vectors.  ``bidiag_null`` below restates
  dgebd2 / dlarfg (LAPACK 3.x, the netlib reference algorithm; scipy 1.15.3 / OpenBLAS in this
  image) and is checked against scipy to 1e-12 in tests/test_oracle_nullspace.py.
"""
import numpy as np
from scipy import linalg


def pinv_null(a):
    """null_space_coordinate.py:8-26.  Returns (pinv (N x M), null basis (N x (N-rank)))."""
    u, s, vh = linalg.svd(a, full_matrices=True, check_finite=False)
    m, n = u.shape[0], vh.shape[1]
    cutoff = np.amax(s) * np.finfo(s.dtype).eps * max(m, n)          # :12-14
    rank = int(np.sum(s > cutoff))                                   # :16
    null = vh[rank:, :].T.conj()                                     # :17
    ur = u[:, :rank] / s[:rank]                                      # :19-20
    pinv = np.transpose(np.conjugate(ur @ vh[:rank]))                # :21
    return pinv, null


def rref(a, row_vectors=True, tol=None):
    """null_space_coordinate.py:40-79 (the MATLAB-style fast rref with a pivot tolerance)."""
    v = np.array(a, dtype=np.float64, copy=True)
    if not row_vectors:
        v = v.T.copy()
    m, n = v.shape
    if tol is None:
        tol = max(m, n) * np.finfo(v.dtype).eps * linalg.norm(v, np.inf)      # :49-50
    i = j = 0
    while i < m and j < n:                                                   # :55
        k = int(np.argmax(np.abs(v[i:m, j]))) + i                            # :57-58
        p = abs(v[k, j])
        if p <= tol:                                                         # :61-64
            v[i:m, j] = 0.0
            j += 1
            continue
        if k != i:                                                           # :69
            v[[i, k], j:n] = v[[k, i], j:n]
        piv = v[i, j:n] / v[i, j]                                            # :71
        v[:, j:n] -= np.outer(v[:, j], piv)                                  # :73
        v[i, j:n] = piv                                                      # :74
        i += 1
        j += 1
    return v if row_vectors else v.T


# ------------------------------------------------------------------ LAPACK's null basis, restated
def _larfg(alpha, x):
    """LAPACK dlarfg: H = I - tau [1;v][1;v]^T with H [alpha;x] = [beta;0] (no safmin rescaling)."""
    xnorm = np.sqrt(np.dot(x, x))
    if xnorm == 0.0:
        return alpha, np.zeros_like(x), 0.0
    beta = -np.copysign(np.hypot(alpha, xnorm), alpha)
    tau = (beta - alpha) / beta
    return beta, x / (alpha - beta), tau


def bidiag_factor(a):
    """LAPACK dgebd2 for M <= N: lower-bidiagonal reduction.  Returns (d, e, G, H) with the
    right reflectors G = [(v, tau)] (v over columns i..N-1, v[0] = 1) and the left reflectors
    H = [(u, tau)] (u over rows i+1..M-1, u[0] = 1)."""
    a = np.array(a, dtype=np.float64, copy=True)
    m, n = a.shape
    d = np.zeros(m)
    e = np.zeros(max(m - 1, 0))
    G, H = [], []
    for i in range(m):
        beta, v, taup = _larfg(a[i, i], a[i, i + 1:])
        vv = np.concatenate([[1.0], v])
        G.append((vv, taup))
        d[i] = beta
        a[i, i] = beta
        a[i, i + 1:] = 0.0
        if i < m - 1:
            w = a[i + 1:, i:] @ vv
            a[i + 1:, i:] -= taup * np.outer(w, vv)
            beta, u, tauq = _larfg(a[i + 1, i], a[i + 2:, i])
            uu = np.concatenate([[1.0], u])
            H.append((uu, tauq))
            e[i] = beta
            a[i + 1, i] = beta
            a[i + 2:, i] = 0.0
            w = uu @ a[i + 1:, i + 1:]
            a[i + 1:, i + 1:] -= tauq * np.outer(uu, w)
    return d, e, G, H


def _apply_P(G, x):
    """x <- G(1) G(2) ... G(M) x  (x has N rows; may be a matrix)."""
    x = np.array(x, dtype=np.float64, copy=True)
    for i in range(len(G) - 1, -1, -1):
        vv, tau = G[i]
        w = vv @ x[i:]
        x[i:] -= tau * np.multiply.outer(vv, w)
    return x


def bidiag_null(a):
    """The orthonormal null basis scipy.linalg.svd(a, full_matrices=True)[2][M:].T returns for a
    full-rank M x N matrix on dgesdd's path 5t: the last N-M columns of P."""
    m, n = a.shape
    _, _, G, _ = bidiag_factor(a)
    return _apply_P(G, np.eye(n)[:, m:])


def bidiag_pinv_apply(a, r):
    """a^+ r through the same factorisation (full row rank): x = P [B^{-1} Q^T r; 0]."""
    m, n = a.shape
    d, e, G, H = bidiag_factor(a)
    y = np.array(r, dtype=np.float64, copy=True)
    for i, (uu, tau) in enumerate(H):
        w = uu @ y[i + 1:]
        y[i + 1:] -= tau * uu * w
    z = np.zeros(n)
    dtol = np.abs(d).max() * np.finfo(np.float64).eps * 8 * (m + 5)
    for i in range(m):
        # rank handling of this build (DESIGN.md "Rank handling"): a pivot below 8 eps (M + 5) max|d| is treated as
        # zero and its solution component dropped -- pinv_null's singular-value cutoff (:12-21) restated for the
        # bidiagonal form; irrelevant for a full-rank matrix
        z[i] = (y[i] - (e[i - 1] * z[i - 1] if i > 0 else 0.0)) / d[i] if abs(d[i]) > dtol else 0.0
    return _apply_P(G, z)
