"""The opt-in canonical chart (`chart_mode = 1`, SURVEY.md section 7.3 H1 "null_mode = exact"): float64 specification.

TEST INFRASTRUCTURE (oracle) -- and the exact specification of `rl_on_manifold_amd/csrc/atacom_chart.h`.

What the reference computes per sub-step (atacom/atacom.py:123-133):

    mu = -Jc^+ (psi + Kc c) + N_c alpha,     N_c = rref(null(Jc), tol = 0.05)        (null_space_coordinate.py:8-79)

`null(Jc)` is LAPACK's orthonormal basis and `rref` compares ITS entries with the tolerance: basis-dependent, and in the
tolerance branch the reference zeroes sub-threshold entries, leaving N_c outside the null space.  Wherever no
pivot comes near the tolerance the result is basis-INDEPENDENT: N_c is THE reduced echelon basis of the null space whose
free ("pivot") coordinates are the first k = dim_q - n_f joints, and Jc^+ y is the unique minimum-norm solution.  What
rref's scan decides -- accept column j as the next free coordinate if the part of the null space still available has
an entry > tol there -- has a basis-independent counterpart: ||P_S e_j|| > tol, the norm of the projection of the
coordinate vector onto S = {v in null(Jc): v = 0 on the free coordinates chosen so far} (identical for k = 1; for k > 1
rref looks at the max-norm of a non-orthonormal basis of S instead -- on states of the three tasks the two pick the same
free coordinates for 98 % of the sub-steps, profiles/r03_chart_agreement.md).

With the slack structure of Jc none of this needs a factorisation of Jc:

    Jc = [[ a   , 0       ],        a = K_f J_f (the equality row, n_f <= 1),  A = K_g J_g  (n_g x dim_q),  s = slacks
          [ A   , diag(s) ]]

  * eliminate the slack velocities, w_g = -(y_g + A_g u) / s_g: a null vector / a solution is determined by its joint
    part u in R^dim_q, and |v|^2 = u^T M u + ..., M = I + sum_g A_g^T A_g / s_g^2   (dim_q x dim_q, SPD);
  * Gamma = M^-1 is the Gram matrix of the coordinate functionals, <e_i, e_j> = e_i Gamma e_j^T ("covariance" of u under
    the unit prior on v, every row imposed); a further row `p x + s w = -y` is imposed by ONE exact rank-one step
    (t = Gamma p^T, S = s^2 + p t;  x += t (-y - p x) / S;  Gamma -= t t^T / S) -- the equality row is the case s = 0;
  * the chart is a Cholesky factorisation of Gamma in joint order that SKIPS a joint whose current diagonal -- exactly
    ||P_S e_j||^2 -- is <= tol^2 (accepting = conditioning on "u_j = its alpha", skipping = leaving the joint to
    follow); free coordinates still missing after the joints go to the first slack columns that pass
    (||P_S e_col||^2 = A_g Gamma A_g^T / s_g^2).  alpha_i is the value of the i-th free coordinate in column order, exactly
    as column i of the reference's N_c has its 1 in the i-th pivot column;
  * N alpha comes out of the same conditioning recursion -- no chart bookkeeping, no row exchanges, and nothing is ever
    zeroed: N is an exact null basis.

Stiff rows.  Eliminating w_g puts the weight 1 / s_g^2 into M: cond(M) ~ (|A_g| / s_g)^2, the SQUARE of what the row does
to cond(Jc), and a slack at zero cannot be eliminated at all.  Rows with |s_g| < theta max|A_g| (theta = 3e-2) are
therefore kept out of M and imposed by conditioning steps.  For the FIRST stiff row p of an environment the slack
velocity stays a coordinate of its own: the state is x = (u, w_p) with unit prior variance on w_p, and row p is the exact
constraint A_p u + s_p w_p = -y_p on it -- so the variance of w_p (what the chart tests), its coupling to the joints and
its value are carried exactly for every s_p >= 0, with no division by s_p anywhere.  Further stiff rows of the same
environment (two constraints active within theta at once) are imposed as measurements of variance s_g^2 and keep the
division w_g = -(y_g + A_g u) / s_g, guarded at |s_g| < 1e-6 max|A_g| (w_g := its alpha if it is a free coordinate, else
0 -- what the reference's min-norm solution tends to).  A row with nothing left to say (S <= 64 eps (s^2 + |p|^2)) is
dropped, as the reference's pinv drops a zero singular value.  If no column passes for a missing free coordinate (a
rank-deficient Jc) it is left without a direction: N has fewer columns and what it has stays exact, as the reference's
rref leaves rows of zeros.
"""
import numpy as np

THETA = 3e-2
TINY = 1e-6
REL = 64 * np.finfo(np.float64).eps      # the kernels use 64 eps of their own precision


def _cholesky_inverse(M):
    L = np.linalg.cholesky(M)
    Li = np.linalg.inv(L)
    return np.einsum('bki,bkj->bij', Li, Li)


def canonical_mu(A_full, s, y, alpha, tol, nf, margin=None, info=None, want_basis=False):
    """A_full [B, nc, nq] = K J (equality row first), s [B, ng], y [B, nc] = psi + Kc c, alpha [B, k]
    -> mu [B, nq + ng] = -Jc^+ y + N alpha in the canonical chart (module docstring).
    info (optional dict): 'fcol' [B, k] free columns (-1: not found), 'n_slack' [B], 'default' [B];
    want_basis: return N [B, n, k] instead (y is ignored)."""
    A_full = np.asarray(A_full, dtype=np.float64)
    s = np.asarray(s, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    alpha = np.asarray(alpha, dtype=np.float64)
    B, nc, nq = A_full.shape
    ng = s.shape[1]
    k = nq - nf
    n1 = nq + 1                                   # extended state x = (u, w_p)
    assert nf in (0, 1) and nc == nf + ng
    ar = np.arange(B)
    A = A_full[:, nf:, :]
    yg = y[:, nf:] * (0.0 if want_basis else 1.0)
    y0 = y[:, 0] * (0.0 if want_basis else 1.0) if nf else None
    tol2 = tol * tol

    def note(val, where=None):
        if margin is not None:
            v = np.abs(val)
            margin[:] = np.minimum(margin, v if where is None else np.where(where, v, np.inf))

    arow = np.abs(A).max(2)
    soft = np.abs(s) >= THETA * arow
    stiff = ~soft
    has_p = stiff.any(1)
    gp = np.where(has_p, stiff.argmax(1), -1)                      # the first stiff row: its slack is a coordinate
    isp = np.arange(ng)[None, :] == gp[:, None]                    # [B, ng]
    om = np.where(soft, 1.0 / np.where(soft, s * s, 1.0), 0.0)
    M = np.eye(nq)[None] + np.einsum('bg,bgi,bgj->bij', om, A, A)
    b = np.einsum('bg,bgi,bg->bi', om, A, yg)
    G = np.zeros((B, n1, n1))
    G[:, :nq, :nq] = _cholesky_inverse(M)
    G[:, nq, nq] = has_p * 1.0                                    # (no stiff row: the extra coordinate is inert)
    x = np.zeros((B, n1))                                          # minimum-norm solution (extended)
    x[:, :nq] = -np.einsum('bij,bj->bi', G[:, :nq, :nq], b)

    def condition(p_ext, s2r, yr, idx):
        """exact rank-one conditioning of (Gamma, x) on  p_ext . x + (noise of variance s2r) = -yr  for the samples idx"""
        t = np.einsum('bij,bj->bi', G[idx], p_ext)
        S = s2r + (p_ext * t).sum(1)
        ok = S > REL * (s2r + (p_ext * p_ext).sum(1))              # Gamma <= I: a row that has nothing left to say
        iS = np.where(ok, 1.0 / np.where(ok, S, 1.0), 0.0)
        e = -yr - (p_ext * x[idx]).sum(1)
        x[idx] += t * (e * iS)[:, None]
        G[idx] -= t[:, :, None] * t[:, None, :] * iS[:, None, None]

    if nf == 1:
        a = A_full[:, 0, :]
        condition(np.concatenate([a, np.zeros((B, 1))], 1), 0.0, y0, ar)
    else:
        a = np.zeros((B, nq))
    for g in range(ng):
        idx = np.nonzero(stiff[:, g])[0]
        if len(idx):
            prim = isp[idx, g]
            p_ext = np.concatenate([A[idx, g, :], np.where(prim, s[idx, g], 0.0)[:, None]], 1)
            condition(p_ext, np.where(prim, 0.0, s[idx, g] ** 2), yg[idx, g], idx)

    # ---- the chart: conditioning recursion over the joints with the skip rule; the targets are either alpha (the action
    # part N alpha) or unit vectors (the columns of N themselves, for the invariant tests)
    n_rhs = k if want_basis else 1
    tgt = np.eye(k)[None].repeat(B, 0) if want_basis else alpha[:, :, None]        # [B, k, n_rhs]
    U = np.zeros((B, n1, n_rhs))
    n_acc = np.zeros(B, dtype=np.int64)
    fcol = np.full((B, k), -1, dtype=np.int64)
    for j in range(nq):
        dj = G[:, j, j]
        need = k - n_acc                              # free coordinates still to find
        acc = (need > 0) & (dj > tol2)
        note(dj - tol2, need > 0)
        idx = np.nonzero(acc)[0]
        if len(idx) == 0:
            continue
        col = G[idx, :, j]
        gain = col / dj[idx][:, None]
        tv = tgt[idx, n_acc[idx], :]                  # target of this free coordinate
        U[idx] += gain[:, :, None] * (tv - U[idx, j, :])[:, None, :]
        G[idx] -= gain[:, :, None] * col[:, None, :]
        fcol[idx, n_acc[idx]] = j
        n_acc[idx] += 1
    # ---- free coordinates still missing after the joints: slack columns, in column order, the first one that passes.
    # The functional of slack column g on the extended state: f_g(x) = A_g u (then w_g = -f_g / s_g and
    # ||P_S e_col||^2 = f_g Gamma f_g^T / s_g^2), except for the coordinate slack p: f_p(x) = w_p itself.
    tiny = (np.abs(s) < TINY * arow) & ~isp
    taken = np.zeros((B, ng), dtype=bool)
    w_tgt = np.zeros((B, ng, n_rhs))
    done = np.zeros(B, dtype=bool)
    F = np.concatenate([np.where(isp[:, :, None], 0.0, A), isp[:, :, None] * 1.0], 2)          # [B, ng, n1]
    thr = tol2 * np.where(isp, 1.0, s * s)
    # (A) more than one missing (0.1 % of the iiwa sub-steps): the general step, Gamma of any rank
    for _ in range(max(k - 1, 0)):
        idx = np.nonzero((n_acc < k - 1) & ~done)[0]
        if len(idx) == 0:
            break
        tg = np.einsum('bij,bgj->bgi', G[idx], F[idx])
        val = (F[idx] * tg).sum(2)
        passed = ~taken[idx] & (tiny[idx] | (val > thr[idx]))
        has = passed.any(1)
        done[idx[~has]] = True
        idx, tg, val, passed = idx[has], tg[has], val[has], passed[has]
        if len(idx) == 0:
            break
        ab = np.arange(len(idx))
        gs = passed.argmax(1)
        tv = tgt[idx, n_acc[idx], :]
        tsel, vsel, tn, pr = tg[ab, gs, :], val[ab, gs], tiny[idx, gs], isp[idx, gs]
        fU = np.einsum('bi,bir->br', F[idx, gs, :], U[idx])
        # slack g: f_g(x) = -s_g target;  coordinate slack p: f_p(x) = +target
        res = np.where(pr[:, None], fU - tv, s[idx, gs][:, None] * tv + fU)
        live = (vsel > 0) & ~tn
        iv = np.where(live, 1.0 / np.where(live, vsel, 1.0), 0.0)
        U[idx] -= tsel[:, :, None] * (res * iv[:, None])[:, None, :]
        G[idx] -= tsel[:, :, None] * tsel[:, None, :] * iv[:, None, None]
        taken[idx, gs] = True
        w_tgt[idx, gs, :] = tv
        fcol[idx, n_acc[idx]] = nq + gs
        n_acc[idx] += 1
    # (B) exactly one missing: S is one-dimensional, Gamma = d d^T / sig -- a scalar test per row
    n_slack = k - n_acc
    idx = np.nonzero((n_acc == k - 1) & ~done)[0] if k >= 1 else np.zeros(0, dtype=np.int64)
    if len(idx):
        Gs = G[idx]
        dg = np.einsum('bii->bi', Gs)
        jm = dg.argmax(1)                             # best-conditioned column of the rank-one remainder
        ab = np.arange(len(idx))
        d = Gs[ab, :, jm]
        sig = dg[ab, jm]
        fd = np.einsum('bgi,bi->bg', F[idx], d)
        val = fd * fd / np.where(sig > 0, sig, 1.0)[:, None]       # = f_g Gamma f_g^T
        passed = ~taken[idx] & (tiny[idx] | (val > thr[idx]))
        if margin is not None:
            rel = np.abs(val / np.maximum(thr[idx] / tol2, 1e-300) - tol2).min(1)
            margin[idx] = np.minimum(margin[idx], rel)
        # nothing passes (a numerically rank-deficient remainder): the untaken column with the largest projection
        gs = np.where(passed.any(1), passed.argmax(1), np.where(taken[idx], -1.0, val).argmax(1))
        fds = fd[ab, gs]
        pr = isp[idx, gs]
        tv = tgt[idx, k - 1, :]
        fU = np.einsum('bi,bir->br', F[idx, gs, :], U[idx])
        res = np.where(pr[:, None], fU - tv, s[idx, gs][:, None] * tv + fU)
        live = (fds != 0) & ~tiny[idx, gs]
        coef = np.where(live[:, None], res / np.where(live, fds, 1.0)[:, None], 0.0)
        U[idx] -= d[:, :, None] * coef[:, None, :]
        taken[idx, gs] = True
        w_tgt[idx, gs, :] = tv
        fcol[idx, k - 1] = nq + gs
    # ---- assembly.  The equality row once more, exactly (rounding only): a u_mn = -y_0, a U = 0
    u_mn, Uj = x[:, :nq].copy(), U[:, :nq, :].copy()
    if nf == 1:
        aa = (a * a).sum(1)
        iaa = np.where(aa > 0, 1.0 / np.where(aa > 0, aa, 1.0), 0.0)
        Uj -= a[:, :, None] * (np.einsum('bi,bir->br', a, Uj) * iaa[:, None])[:, None, :]
        u_mn -= a * (((a * u_mn).sum(1) + y0) * iaa)[:, None]
    big = np.abs(s) >= TINY * arow
    inv_s = np.where(big, 1.0 / np.where(big, s, 1.0), 0.0)
    w_mn = -(yg + np.einsum('bgi,bi->bg', A, u_mn)) * inv_s
    w_al = -np.einsum('bgi,bir->bgr', A, Uj) * inv_s[:, :, None]
    w_al = np.where(taken[:, :, None], w_tgt, w_al)               # a free slack coordinate takes its target itself
    # the coordinate slack: its value is a component of the extended state
    w_mn = np.where(isp, x[:, nq][:, None], w_mn)
    w_al = np.where(isp[:, :, None], U[:, nq, :][:, None, :], w_al)
    if want_basis:
        out = np.concatenate([Uj, w_al], 1)
    else:
        out = np.concatenate([u_mn[:, :, None] + Uj, w_mn[:, :, None] + w_al], 1)
    if info is not None:
        info.update({'fcol': fcol, 'n_slack': n_slack, 'default': (fcol == np.arange(k)[None, :]).all(1)})
    return out if want_basis else out[:, :, 0]


def null_basis(A_full, s, tol, nf):
    """The chart's null basis N [B, n, k] (exact: Jc N = 0) and its free columns -- for the invariant tests."""
    B, nc, nq = np.asarray(A_full).shape
    info = {}
    N = canonical_mu(A_full, s, np.zeros((B, nc)), np.zeros((B, nq - nf)), tol, nf, info=info, want_basis=True)
    return N, info['fcol']
