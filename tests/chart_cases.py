"""Inputs for the canonical-chart tests (CPU and GPU): (A, s, y, alpha) systems sampled from oracle rollouts of the three
tasks -- the states the engine actually visits, chart switches included -- plus hand-made degenerate ones."""
import numpy as np

from oracle import atacom_batched as ob
from oracle import atacom_scalar as osc

IIWA_INIT_Q = np.array([0.0, 0.7135214629060707, 0.0, -0.5024756033561426, 0.0, 1.9256631778550268])
SPECS = {'circle': osc.circle_spec, 'planar': osc.planar_spec, 'iiwa': osc.iiwa_spec}
_CACHE = {}


def init_q(name, B, rng, sigma=0.05):
    q0 = {'circle': np.array([-1.0, 0.0]), 'planar': ob.robots.PLANAR_INIT_Q, 'iiwa': IIWA_INIT_Q}[name]
    return q0 + (rng.normal(0, sigma, (B, len(q0))) if name != 'circle' else 0.0)


def away_init_q(B, rng, sigma=0.3, min_abs=0.1):
    """iiwa joint states AWAY from the reset pose: q_init + N(0, sigma^2), Newton steps onto the
    equality constraint (tip height), kept only if every inequality holds with a margin and NO joint is within `min_abs`
    rad of zero -- the reset pose is exactly planar (q1 = q3 = q5 = 0), the regime where LAPACK's null basis is
    ill-determined.  About one draw in ten survives."""
    spec = SPECS['iiwa']()
    keep, have = [], 0
    while have < B:
        q = IIWA_INIT_Q + rng.normal(0, sigma, (4 * B, 6))
        for _ in range(3):
            fun, J, _ = ob.constraint_terms(spec, q, np.zeros_like(q))
            Jf = J[:, 0, :]
            q = q - Jf * (fun[:, :1] / (Jf * Jf).sum(1, keepdims=True))
        fun, _, _ = ob.constraint_terms(spec, q, np.zeros_like(q))
        ok = (fun[:, 1:] < -1e-3).all(1) & (np.abs(fun[:, 0]) < 1e-6) & (np.abs(q) > min_abs).all(1)
        keep.append(q[ok])
        have += int(ok.sum())
    return np.concatenate(keep)[:B]


def rollout_systems(name, B=256, T=40, seed=0, stride=3):
    """Every `stride`-th (Jc, rhs, N) the reference-chart oracle factorised during a B x T rollout, as
    dict(A [n, c, q], s [n, g], y [n, c], Jc [n, c, q + g], Nr [n, q + g, k] = the reference's rref'd basis,
         skipped [n] = the reference took its tolerance branch)."""
    key = (name, B, T, seed, stride)
    if key in _CACHE:
        return _CACHE[key]
    spec = SPECS[name]()
    rng = np.random.default_rng(seed)
    env = ob.BatchedAtacomEnv(spec, B, init_q=init_q(name, B, rng))
    store = []
    orig = ob.bidiag_solve_null

    def hook(Jc, rhs, k, cond=None):
        x, N = orig(Jc, rhs, k, cond)
        store.append((Jc.copy(), rhs.copy(), N.copy()))
        return x, N
    ob.bidiag_solve_null = hook
    try:
        for t in range(T):
            a = rng.uniform(-1.2, 1.2, (B, spec.n_null))
            _, _, ab, _ = env.step(a)
            last = ab | (env.t >= spec.horizon)
            if last.any():
                env.reset(last)
    finally:
        ob.bidiag_solve_null = orig
    Jc = np.concatenate([s_[0] for s_ in store])[::stride]
    rhs = np.concatenate([s_[1] for s_ in store])[::stride]
    N = np.concatenate([s_[2] for s_ in store])[::stride]
    nq, nf, ng = spec.dim_q, spec.n_f, spec.n_g
    skipped = np.zeros(len(Jc), dtype=bool)
    Nr = ob.rref_tol(N, spec.rref_tol, skipped=skipped)
    out = {'A': Jc[:, :, :nq].copy(), 's': Jc[:, nf + np.arange(ng), nq + np.arange(ng)].copy(), 'y': rhs, 'Jc': Jc,
           'Nr': Nr, 'skipped': skipped, 'spec': spec}
    _CACHE[key] = out
    return out


def degenerate_systems(name, seed=1):
    """Hand-made hard cases on top of rollout systems: slacks at exactly zero, tiny slacks (1e-9 ... 1e-2 of the row),
    several at once, a vanishing equality row, the pair of mutually negative table rows both nearly active."""
    base = rollout_systems(name)
    spec = base['spec']
    rng = np.random.default_rng(seed)
    n = 96
    idx = rng.choice(len(base['A']), n, replace=False)
    A, s, y = base['A'][idx].copy(), base['s'][idx].copy(), base['y'][idx].copy()
    ng, nf = spec.n_g, spec.n_f
    arow = np.abs(A[:, nf:, :]).max(2)
    for i in range(n):
        kind = i % 6
        g = rng.integers(ng)
        if kind == 0:
            s[i, g] = 0.0
        elif kind == 1:
            s[i, g] = arow[i, g] * 10.0 ** rng.uniform(-9, -2) * rng.choice([-1, 1])
        elif kind == 2:
            gs = rng.choice(ng, min(3, ng), replace=False)
            s[i, gs] = arow[i, gs] * 10.0 ** rng.uniform(-6, -2, len(gs))
        elif kind == 3 and nf:
            A[i, 0, :] = 0.0                      # the straight-up pose: the equality row vanishes
        elif kind == 4 and ng > 2:
            # the table rows "-y_w - b" and "y_w - b" are exact negatives of each other in both arm tasks (rows 1, 2 of
            # the inequality block): both slacks small -> two dependent near-equalities
            s[i, 1:3] = arow[i, 1:3] * 10.0 ** rng.uniform(-4, -2, 2)
        elif kind == 5:
            s[i, rng.choice(ng, 2, replace=False) if ng > 1 else 0] = 0.0      # two slacks at zero (an infeasible reset)
    return {'A': A, 's': s, 'y': y, 'spec': spec, 'kind': np.arange(n) % 6}


def jc_of(A, s, nf):
    n, nc, nq = A.shape
    ng = s.shape[1]
    Jc = np.zeros((n, nc, nq + ng))
    Jc[:, :, :nq] = A
    Jc[:, nf + np.arange(ng), nq + np.arange(ng)] = s
    return Jc


def reference_golden(name):
    """Golden set G12 (tests/golden/chart_reference.npz, oracle/gen_golden.py:gen_chart): J_c systems with what the
    REFERENCE's own pinv_null + rref(tol = 0.05) + atacom.py:127-133 computed for them.
    -> dict(A, s, y, alpha, mu, N, Jc, exact [n] = the reference zeroed nothing (J_c N = 0),
            free [n, k] = the free ("pivot") coordinates of its reduced echelon basis (-1: not in that form))."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'chart_reference.npz'))
    spec = SPECS[name]()
    nq, nf, ng = spec.dim_q, spec.n_f, spec.n_g
    Jc, N = g[name + '_Jc'], g[name + '_N']
    k = N.shape[2]
    resid = np.abs(np.einsum('bcn,bnk->bck', Jc, N)).max((1, 2))
    free = np.full((len(Jc), k), -1)
    for b in range(len(Jc)):
        for i in range(k):
            rows = [r for r in range(N.shape[1]) if N[b, r, i] == 1.0 and np.count_nonzero(N[b, r]) == 1]
            if rows:
                free[b, i] = rows[0]
    return {'A': Jc[:, :, :nq].copy(), 's': Jc[:, nf + np.arange(ng), nq + np.arange(ng)].copy(), 'y': g[name + '_y'],
            'alpha': g[name + '_alpha'], 'mu': g[name + '_mu'], 'N': N, 'Jc': Jc, 'exact': resid < 1e-9, 'free': free,
            'spec': spec}
