"""Batch-vectorised float64 restatement of the ATACOM step (oracle; test infrastructure only).

Same arithmetic as oracle/atacom_scalar.py (which follows the reference function by function), but
vectorised over a leading batch axis so GPU parity tests at thousands of environments finish in
seconds.  Two deliberate differences in *how* (not what) it computes, both verified against the
scalar oracle in tests/test_oracle_batched.py:

  * the orthonormal null basis and the pseudo-inverse solve come from the Householder
    bidiagonalisation that LAPACK's dgesdd performs internally (oracle/nullspace.py docstring),
    not from a per-environment scipy SVD call;
  * -Jc^+ psi - Jc^+ (Kc c) is evaluated as one solve with the summed right-hand side.

This is also exactly the algorithm the HIP kernels implement, so it doubles as their specification.
"""
import numpy as np

from . import robots
from .atacom_scalar import (ENV_CIRCLE, ENV_PLANAR, ENV_IIWA, TABLE_LENGTH, TABLE_WIDTH, GOAL_WIDTH,
                            MALLET_RADIUS, PUCK_RADIUS, UNIVERSAL_HEIGHT, HIT_RANGE, GOAL_POS, E_MALLET, E_RIM, DEFEND_START_RANGE,
                            MODE_ATACOM, MODE_ERROR_CORRECTION, MODE_TERMINATED)


def _diag_batch(x):
    out = np.zeros(x.shape + (x.shape[-1],))
    idx = np.arange(x.shape[-1])
    out[..., idx, idx] = x
    return out


def constraint_terms(spec, q, dq):
    """Batched (fun_origin[B,c], J[B,c,q], b_state[B,c]); see atacom_scalar.constraint_terms."""
    q = np.asarray(q, dtype=np.float64)
    dq = np.asarray(dq, dtype=np.float64)
    B = q.shape[0]
    if spec.env_id == ENV_CIRCLE:
        fun = np.stack([q[:, 0] ** 2 + q[:, 1] ** 2 - 1.0, -q[:, 1] - 0.5], -1)
        J = np.zeros((B, 2, 2))
        J[:, 0, 0], J[:, 0, 1], J[:, 1, 1] = 2 * q[:, 0], 2 * q[:, 1], -1.0
        b = np.stack([2 * dq[:, 0] ** 2 + 2 * dq[:, 1] ** 2, np.zeros(B)], -1)
        return fun, J, b
    bx = TABLE_LENGTH / 2 - MALLET_RADIUS
    by = TABLE_WIDTH / 2 - MALLET_RADIUS
    if spec.env_id == ENV_PLANAR:
        p, _ = robots.planar_fk(q)
        pw = p + spec.base_xy
        Je = robots.planar_jacobian(q)
        acc = robots.planar_bias(q, dq, spec.bias_mode)
        lim = robots.PLANAR_POS_LIMIT
        fun = np.concatenate([np.stack([-pw[:, 0] - bx, -pw[:, 1] - by, pw[:, 1] - by], -1),
                              q ** 2 - lim ** 2], -1)
        J = np.concatenate([np.stack([-Je[:, 0], -Je[:, 1], Je[:, 1]], 1), 2 * _diag_batch(q)], 1)
        b = np.concatenate([np.stack([-acc[:, 0], -acc[:, 1], acc[:, 1]], -1), 2 * dq ** 2], -1)
        return fun, J, b
    if spec.env_id == ENV_IIWA:
        pe, _ = robots.iiwa_frame(q, 'ee')
        p4, _ = robots.iiwa_frame(q, 'link_4')
        p7, _ = robots.iiwa_frame(q, 'link_7')
        Je = robots.iiwa_frame_jacobian(q, 'ee')
        J4 = robots.iiwa_frame_jacobian(q, 'link_4')
        J7 = robots.iiwa_frame_jacobian(q, 'link_7')
        ae = robots.iiwa_frame_bias(q, dq, 'ee', spec.bias_mode)
        a4 = robots.iiwa_frame_bias(q, dq, 'link_4', spec.bias_mode)
        a7 = robots.iiwa_frame_bias(q, dq, 'link_7', spec.bias_mode)
        xw = pe[:, 0] + spec.base_xy[0]
        yw = pe[:, 1] + spec.base_xy[1]
        lim = robots.IIWA_POS_LIMIT[:6]
        fun = np.concatenate([np.stack([pe[:, 2] - UNIVERSAL_HEIGHT, -xw - bx, -yw - by, yw - by,
                                        -p4[:, 2] + 0.36, -p7[:, 2] + 0.25], -1), q ** 2 - lim ** 2], -1)
        J = np.concatenate([np.stack([Je[:, 2], -Je[:, 0], -Je[:, 1], Je[:, 1], -J4[:, 2], -J7[:, 2]], 1),
                            2 * _diag_batch(q)], 1)
        b = np.concatenate([np.stack([ae[:, 2], -ae[:, 0], -ae[:, 1], ae[:, 1], -a4[:, 2], -a7[:, 2]], -1),
                            2 * dq ** 2], -1)
        return fun, J, b
    raise ValueError(spec.env_id)


def mallet_xy_world(spec, q):
    if spec.env_id == ENV_PLANAR:
        return robots.planar_fk(q)[0] + spec.base_xy
    return robots.iiwa_frame(q, 'ee')[0][:, :2] + spec.base_xy


# ------------------------------------------------------------------ batched dgebd2 / rref
def _larfg(alpha, x):
    """Batched dlarfg.  alpha[B], x[B,L] -> beta[B], v[B,L], tau[B]."""
    xnorm = np.sqrt((x * x).sum(-1))
    nz = xnorm != 0.0
    beta = np.where(nz, -np.copysign(np.hypot(alpha, xnorm), alpha), alpha)
    safe = np.where(nz, beta, 1.0)
    tau = np.where(nz, (beta - alpha) / safe, 0.0)
    den = np.where(nz, alpha - beta, 1.0)
    v = np.where(nz[:, None], x / den[:, None], 0.0)
    return beta, v, tau


def bidiag_solve_null(Jc, rhs, k, cond=None):
    """For each Jc[b] (c x n, full row rank): x = Jc^+ rhs[b]  and the orthonormal null basis N[b]
    (n x k) = last k columns of P = G(1)...G(c) -- the basis LAPACK's SVD returns (nullspace.py)."""
    a = np.array(Jc, dtype=np.float64, copy=True)
    y = np.array(rhs, dtype=np.float64, copy=True)
    B, m, n = a.shape
    d = np.zeros((B, m))
    e = np.zeros((B, max(m - 1, 0)))
    Gv, Gt = [], []
    for i in range(m):
        beta, v, taup = _larfg(a[:, i, i], a[:, i, i + 1:])
        vv = np.concatenate([np.ones((B, 1)), v], -1)
        Gv.append(vv)
        Gt.append(taup)
        d[:, i] = beta
        if i < m - 1:
            w = np.einsum('brc,bc->br', a[:, i + 1:, i:], vv)
            a[:, i + 1:, i:] -= taup[:, None, None] * w[:, :, None] * vv[:, None, :]
            beta, u, tauq = _larfg(a[:, i + 1, i], a[:, i + 2:, i])
            uu = np.concatenate([np.ones((B, 1)), u], -1)
            e[:, i] = beta
            w = np.einsum('br,brc->bc', uu, a[:, i + 1:, i + 1:])
            a[:, i + 1:, i + 1:] -= tauq[:, None, None] * uu[:, :, None] * w[:, None, :]
            wy = (uu * y[:, i + 1:]).sum(-1)
            y[:, i + 1:] -= (tauq * wy)[:, None] * uu
    if cond is not None:
        # conditioning of the solve: ratio of the extreme singular values of the bidiagonal factor (= those of Jc)
        Bm = np.zeros((B, m, m))
        ii = np.arange(m)
        Bm[:, ii, ii] = d
        if m > 1:
            Bm[:, ii[1:], ii[:-1]] = e
        sv = np.linalg.svd(Bm, compute_uv=False)
        cond[:] = np.maximum(cond, sv[:, 0] / np.maximum(sv[:, -1], 1e-300))
    z = np.zeros((B, n))
    dtol = np.abs(d).max(-1) * np.finfo(np.float64).eps * 8 * (m + 5)       # guarded solve, see nullspace.bidiag_pinv_apply
    for i in range(m):
        prev = e[:, i - 1] * z[:, i - 1] if i > 0 else 0.0
        ok = np.abs(d[:, i]) > dtol
        z[:, i] = np.where(ok, (y[:, i] - prev) / np.where(ok, d[:, i], 1.0), 0.0)
    X = np.zeros((B, n, k + 1))
    X[:, :, 0] = z
    X[:, np.arange(m, n), np.arange(1, k + 1)] = 1.0
    for i in range(m - 1, -1, -1):
        vv, tau = Gv[i], Gt[i]
        w = np.einsum('bc,bck->bk', vv, X[:, i:, :])
        X[:, i:, :] -= tau[:, None, None] * vv[:, :, None] * w[:, None, :]
    return X[:, :, 0], X[:, :, 1:]


def rref_tol(N, tol, margin=None, skipped=None, forced_skip=None):
    """Batched null_space_coordinate.rref(N, row_vectors=False, tol) (lines 40-79).
    forced_skip (optional bool [B, n]; parity tests only): take the pivot-or-skip decision of column j from forced_skip[:, j]
    instead of from the tolerance test -- the float64 elimination on SOMEBODY ELSE'S chart decisions (a float32 device's),
    which separates "the device took the other side of the reference's tolerance test" from arithmetic error.
    margin (optional, [B], updated in place with minimum): how far the DISCRETE decisions of the elimination were from
    `skipped` (optional bool [B], OR-updated): the elimination took the tolerance branch (:56-63) at least once, i.e. the
    result is NOT the reduced echelon basis with the first k coordinates free (and has been zeroed somewhere).
    going the other way -- |p - tol| of every pivot-or-skip test (:56-63) and the gap between the largest and the
    second largest candidate of every arg-max (:55).  A float32 evaluation of the same matrix can only take a different
    branch where this margin is of the order of its rounding error; the parity tests use it to separate "arithmetic
    error" from "the reference's own discontinuity"."""
    V = np.array(np.swapaxes(N, 1, 2), dtype=np.float64, copy=True)       # B x k x n
    B, m, n = V.shape
    i = np.zeros(B, dtype=np.int64)
    ar = np.arange(B)
    rows = np.arange(m)[None, :]
    for j in range(n):
        active = i < m
        if not active.any():
            break
        col = np.abs(V[:, :, j])
        col = np.where(rows >= i[:, None], col, -1.0)
        kk = np.argmax(col, axis=1)                       # first maximum, like np.argmax
        p = col[ar, kk]
        if margin is not None:
            second = col.copy()
            second[ar, kk] = -1.0
            gap = p - second.max(axis=1)                  # arg-max tie margin (inf-like when a single candidate is left)
            gap = np.where(second.max(axis=1) < 0, np.inf, gap)
            mg = np.minimum(np.abs(p - tol), np.where(p > tol, gap, np.inf))
            margin[:] = np.where(active, np.minimum(margin, mg), margin)
        piv = active & (p > tol)
        if forced_skip is not None:
            piv = active & ~forced_skip[:, j] & (p > 0)
        skip = active & ~piv
        if skipped is not None:
            skipped |= skip
        # negligible column: zero it from row i down
        zero_mask = skip[:, None] & (rows >= i[:, None])
        V[:, :, j] = np.where(zero_mask, 0.0, V[:, :, j])
        if piv.any():
            b = ar[piv]
            ib, kb = i[piv], kk[piv]
            tmp = V[b, ib, :].copy()
            V[b, ib, :] = V[b, kb, :]
            V[b, kb, :] = tmp
            prow = V[b, ib, j:] / V[b, ib, j][:, None]
            colj = V[b, :, j].copy()
            V[b, :, j:] -= colj[:, :, None] * prow[:, None, :]
            V[b, ib, j:] = prow
            i[piv] += 1
    return np.swapaxes(V, 1, 2)


def device_uniform(seed, env, episode, draw):
    """The engine's counter-based generator (atacom_kernels.h: device_uniform), restated with numpy uint32 arithmetic."""
    def h(x):
        x = np.asarray(x, dtype=np.uint64) & 0xffffffff
        x ^= x >> 16; x = (x * 0x7feb352d) & 0xffffffff
        x ^= x >> 15; x = (x * 0x846ca68b) & 0xffffffff
        x ^= x >> 16
        return x
    key = (np.uint64(seed) + np.asarray(env, dtype=np.uint64) * 0x9E3779B9 + np.asarray(episode, dtype=np.uint64) * 0x85EBCA6B
           + np.asarray(draw, dtype=np.uint64) * 0xC2B2AE35) & 0xffffffff
    return (h(h(key)) >> 8).astype(np.float64) / 16777216.0


OBS_NOISE_STD = 0.001        # env_single.py:105-107
ENV_NOISE_FORCE = 0.0005     # env_base.py:176-180
OBS_DELAY_ALPHA = 0.5        # env_single.py:114-117


def device_normal(seed, env, episode, t, idx, n_idx):
    """Standard normal draw number `idx` (of n_idx per env step) of step t of an episode: Box-Muller on two draws of the
    engine's counter-based generator (atacom_kernels.h: device_normal).  The reference draws from numpy's global, unseeded
    generator (np.random.randn), so only the distribution can match."""
    d = 8 + 2 * (np.asarray(t, dtype=np.int64) * n_idx + idx)
    u1 = device_uniform(seed, env, episode, d)
    u2 = device_uniform(seed, env, episode, d + 1)
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)


# ------------------------------------------------------------------ the batched environment
class BatchedAtacomEnv:
    def __init__(self, spec, batch, init_q=None, init_dq=None, init_puck=None, random_init=False, seed=0):
        self.spec, self.B = spec, batch
        self.random_init, self.seed = random_init, seed
        self.episode = np.zeros(batch, dtype=np.int64)
        self.ep_cur = np.zeros(batch, dtype=np.int64)       # id of the running episode (keys the noise draws)
        self.env_index = np.arange(batch)                   # per-env (survives tests/parity_tools.slice_env): keys the draws
        self.noisy = bool(getattr(spec, 'obs_noise', False) or getattr(spec, 'obs_delay', False)
                          or getattr(spec, 'env_noise', False))
        if self.noisy and spec.env_id == ENV_CIRCLE:
            raise ValueError('obs_noise / obs_delay / env_noise exist for the air-hockey environments only')
        self.n_idx = 3 + 2 * spec.substeps                   # normal draws per env step: 3 (observation) + 2 per sub-step (force)
        nq = spec.dim_q
        if init_q is None:
            init_q = {ENV_CIRCLE: np.array([-1.0, 0.0]), ENV_PLANAR: robots.PLANAR_INIT_Q,
                      ENV_IIWA: np.zeros(6)}[spec.env_id]
        self.init_q = np.broadcast_to(np.asarray(init_q, dtype=np.float64), (batch, nq)).copy()
        self.init_dq = np.zeros((batch, nq)) if init_dq is None else \
            np.broadcast_to(np.asarray(init_dq, dtype=np.float64), (batch, nq)).copy()
        self.defend = spec.env_id == ENV_PLANAR and getattr(spec, 'task', 0) == 1
        if init_puck is not None:
            pk = np.asarray(init_puck, dtype=np.float64)
        elif self.defend:           # AirHockeyDefend.setup [upstream]: middle of start_range in x, y = 0, velocity (-1, 0)
            pk = np.array([DEFEND_START_RANGE[0].mean(), 0.0, 0, -1.0, 0, 0.0])
        else:
            pk = np.array([HIT_RANGE[0].mean(), HIT_RANGE[1].mean(), 0, 0, 0, 0.0])
        self.init_puck = np.broadcast_to(pk, (batch, 6)).copy()
        self.q = np.zeros((batch, nq))
        self.dq = np.zeros((batch, nq))
        self.s = np.zeros((batch, spec.n_g))
        self.puck = np.zeros((batch, 6))
        # row N4 (dynamics_mode = 1): the three servo joints -- joint 7 and the striker's universal joint
        self.qx = np.zeros((batch, 3))
        self.dqx = np.zeros((batch, 3))
        # obs_delay: the low-pass state obs_prev[3:6] (puck velocity) and obs_prev[robot velocities] of env_single.py:114-119
        self.fv = np.zeros((batch, 3 + nq))
        self.has_hit = np.zeros(batch, dtype=bool)
        self.has_bounce = np.zeros(batch, dtype=bool)       # task 'D' only
        self.r_hit = np.zeros(batch)
        self.vel_hit_x = np.zeros(batch)
        self.t = np.zeros(batch, dtype=np.int64)
        self.stat_sum = np.zeros(batch)
        self.stat_cnt = np.zeros(batch, dtype=np.int64)
        self.stat_cmax = np.full(batch, -np.inf)
        self.stat_dqmax = np.full(batch, -np.inf)
        self.reset()

    def slack_init(self, q, dq):
        sp = self.spec
        fun, J, _ = constraint_terms(sp, q, dq)
        g = (fun + sp.K * np.einsum('bcq,bq->bc', J, dq))[:, sp.n_f:]
        return np.sqrt(np.maximum(-2.0 * g, 0.0))

    def reset(self, mask=None):
        m = np.ones(self.B, dtype=bool) if mask is None else np.asarray(mask, dtype=bool)
        self.q[m], self.dq[m], self.puck[m] = self.init_q[m], self.init_dq[m], self.init_puck[m]
        self.has_hit[m], self.r_hit[m], self.vel_hit_x[m], self.t[m] = False, 0.0, 0.0, 0
        self.has_bounce[m] = False
        self.qx[m], self.dqx[m] = 0.0, 0.0
        if self.random_init and m.any():
            env, ep = self.env_index[m], self.episode[m]
            u = [device_uniform(self.seed, env, ep, i) for i in range(5)]
            if self.spec.env_id == ENV_CIRCLE:              # circle_base.py:36-42
                y = -0.5 + 1.5 * u[0]
                x = np.sqrt(np.maximum(1 - y * y, 0)) * np.where(u[1] < 0.5, -1.0, 1.0)
                dx = -1 + 2 * u[2]
                dy = -x * dx / y
                sp = u[3] / np.sqrt(dx * dx + dy * dy)
                self.q[m] = np.stack([x, y], -1)
                self.dq[m] = np.stack([dx * sp, dy * sp], -1)
            elif self.defend:                               # AirHockeyDefend.setup [upstream]
                self.puck[m, 0] = 0.25 + 0.4 * u[0]
                self.puck[m, 1] = -0.4 + 0.8 * u[1]
                v, ang = 1.0 + 1.2 * u[2], -0.5 + u[3]
                self.puck[m, 3] = -np.cos(ang) * v
                self.puck[m, 4] = np.sin(ang) * v
                self.puck[m, 5] = -1.0 + 2.0 * u[4]
            else:                                           # env_hitting.py:24-25
                self.puck[m, 0] = -0.6 + 0.4 * u[0]
                self.puck[m, 1] = -0.4 + 0.8 * u[1]
            self.ep_cur[m] = ep
            self.episode[m] += 1
        elif self.noisy and m.any():
            self.ep_cur[m] = self.episode[m]                # every reset starts a new episode of the noise streams
            self.episode[m] += 1
        if m.any():
            self.s[m] = self.slack_init(self.q[m], self.dq[m])
            # the first observation of an episode is unfiltered (the reference's obs_prev is None there -- and its
            # _create_observation would raise on it; oracle/__init__.py "obs_delay")
            self.fv[m] = np.concatenate([self.puck[m, 3:6], self.dq[m]], 1)
        return self.observation()

    def set_state(self, q, dq, s=None, puck=None):
        self.q[:], self.dq[:] = q, dq
        if puck is not None:
            self.puck[:] = puck
        self.s[:] = self.slack_init(self.q, self.dq) if s is None else s

    def observation(self):
        sp = self.spec
        if sp.env_id == ENV_CIRCLE:
            return np.concatenate([self.q, self.dq], -1)
        pk = self.puck
        pose = np.stack([pk[:, 0] - sp.base_xy[0], pk[:, 1] - sp.base_xy[1], pk[:, 2]], -1)
        if getattr(sp, 'obs_noise', False):                  # env_single.py:105-107
            env = self.env_index
            pose = pose + OBS_NOISE_STD * np.stack(
                [device_normal(self.seed, env, self.ep_cur, self.t, c, self.n_idx) for c in range(3)], -1)
        if getattr(sp, 'obs_delay', False):                  # :114-117 (the filtered values; updated where the reference
            pv, rv = self.fv[:, :3], self.fv[:, 3:]          # calls _create_observation: see step)
        else:
            pv, rv = pk[:, 3:6], self.dq
        return np.concatenate([pose, pv, self.q, rv], -1)

    def tangent_space_accel(self, q, dq, s, alpha, terms=None):
        sp = self.spec
        nq, nf, ng, nc, B = sp.dim_q, sp.n_f, sp.n_g, sp.n_c, q.shape[0]
        fun, J, bst = constraint_terms(sp, q, dq) if terms is None else terms
        Jdq = np.einsum('bcq,bq->bc', J, dq)
        Jc = np.zeros((B, nc, nq + ng))
        Jc[:, :, :nq] = sp.K[None, :, None] * J + 0.0     # '+ 0.0': -0.0 -> +0.0 like the reference's matmul
        idx = np.arange(ng)
        Jc[:, nf + idx, nq + idx] = s
        noise = getattr(self, 'jc_noise', None)
        if noise is not None:
            # sensitivity probes only (tests/parity_tools.py): an UNSTRUCTURED perturbation of J_c -- every entry,
            # structural zeros included -- of size eps * max|J_c|: what rounding inside a float32 factorisation amounts to
            eps, rng = noise
            Jc = Jc + eps * np.abs(Jc).max((1, 2), keepdims=True) * rng.choice([-1.0, 1.0], Jc.shape)
        psi = Jdq + sp.K * bst
        c = fun + sp.K * Jdq
        c[:, nf:] += 0.5 * s ** 2
        if sp.mode == MODE_ATACOM and getattr(sp, 'chart_mode', 0) == 1:
            # opt-in canonical chart: same mu wherever the reference's rref takes no tolerance branch, an exact null basis
            # everywhere (oracle/canonical_chart.py)
            from .canonical_chart import canonical_mu
            A = Jc[:, :, :nq]
            if noise is not None:
                A = (sp.K[None, :, None] * J + 0.0) * (1.0 + noise[0] * noise[1].choice([-1.0, 1.0], A.shape))
            info = getattr(self, 'chart_info', None)
            track = getattr(self, 'chart_default', None)
            if track is not None and info is None:
                info = {}
            mu = canonical_mu(A, s, psi + sp.Kc * c, alpha, sp.rref_tol, nf, getattr(self, 'decision_margin', None), info)
            if track is not None:
                track &= info['default']          # the default chart (first k joints free) in EVERY sub-step of the step
            return mu
        if sp.mode == MODE_ERROR_CORRECTION:       # error_correction_wrapper.py:117-130
            x, _ = bidiag_solve_null(Jc, sp.Kc * c, sp.n_null)
            return np.concatenate([alpha, np.zeros((B, ng))], -1) - x
        x, N = bidiag_solve_null(Jc, psi + sp.Kc * c, sp.n_null, getattr(self, 'cond_number', None))
        follow = getattr(self, 'chart_follow', None)       # parity tests: Jc -> bool [B, n], the decisions to follow
        Nr = rref_tol(N, sp.rref_tol, getattr(self, 'decision_margin', None), getattr(self, 'chart_skipped', None),
                      None if follow is None else follow(Jc))
        return -x + np.einsum('bnk,bk->bn', Nr, alpha)

    def acc_truncation(self, dq, ddq):
        sp = self.spec
        up = np.maximum(np.minimum(sp.acc_max, -sp.Kq * (dq - sp.vel_max)), -sp.acc_max)
        lo = np.minimum(np.maximum(-sp.acc_max, -sp.Kq * (dq + sp.vel_max)), sp.acc_max)
        return np.clip(ddq, lo, up)

    def track_margins(self, on=True):
        """Per env step, record how close the step's discrete decisions came to flipping:
        decision_margin [B] -- the rref pivot / arg-max tests of every sub-step (see rref_tol);
        cond_number [B]     -- conditioning of the pseudo-inverse solve (sigma_max / sigma_min of J_c);
        contact_margin [B]  -- the puck model's tests (contact distance, approach speed, rims, goal mouth, hit latch,
                               absorbing thresholds), in metres / metres per second."""
        if on:
            self.chart_skipped = np.zeros(self.B, dtype=bool)   # the reference's rref took its tolerance branch this step
            self.chart_default = np.ones(self.B, dtype=bool)    # canonical chart: the default chart in every sub-step
            self.decision_margin = np.full(self.B, np.inf)
            self.contact_margin = np.full(self.B, np.inf)
            self.cond_number = np.zeros(self.B)          # largest sigma_max / sigma_min of J_c over the sub-steps
        else:
            for k in ('decision_margin', 'contact_margin', 'cond_number', 'chart_skipped', 'chart_default'):
                self.__dict__.pop(k, None)

    def _cm(self, values, where=None):
        cm = getattr(self, 'contact_margin', None)
        if cm is not None:
            v = np.abs(values)
            cm[:] = np.minimum(cm, v if where is None else np.where(where, v, np.inf))

    def step(self, action):
        sp = self.spec
        nq = sp.dim_q
        if hasattr(self, 'decision_margin'):
            self.chart_skipped[:] = False
            self.chart_default[:] = True
            self.decision_margin[:] = np.inf
            self.contact_margin[:] = np.inf
            self.cond_number[:] = 0.0
        act = np.clip(np.asarray(action, dtype=np.float64), -1.0, 1.0)
        alpha = act * (sp.alpha_max if sp.mode == MODE_ATACOM else (sp.acc_max if sp.mode == MODE_ERROR_CORRECTION else 1.0))
        if sp.env_id == ENV_CIRCLE and sp.mode == MODE_TERMINATED:
            c_pre = np.stack([np.abs(self.q[:, 0] ** 2 + self.q[:, 1] ** 2 - 1), -self.q[:, 1] - 0.5], -1)
            dq_pre = np.abs(self.dq) - 1.0
            self._log(c_pre.max(-1), c_pre.max(-1), dq_pre.max(-1))
            a = alpha * 10.0
            self.q = self.q + (self.dq * sp.dt_base + a * sp.dt_base ** 2 / 2)
            self.dq = self.dq + a * sp.dt_base
            absorbing = np.maximum(c_pre.max(-1), dq_pre.max(-1)) > sp.term_tol
            reward = np.where(absorbing, -100.0, np.exp(-np.hypot(1.0 - self.q[:, 0], self.q[:, 1])))
        elif sp.env_id == ENV_CIRCLE:
            c_pre = np.stack([np.abs(self.q[:, 0] ** 2 + self.q[:, 1] ** 2 - 1), -self.q[:, 1] - 0.5], -1)
            dq_pre = np.abs(self.dq) - 1.0
            self._log(c_pre.max(-1), c_pre.max(-1), dq_pre.max(-1))
            mu = self.tangent_space_accel(self.q, self.dq, self.s, alpha)
            self.s = self.s + mu[:, nq:] * sp.dt
            ddq = self.acc_truncation(self.dq, mu[:, :nq])
            a = np.clip(ddq / sp.acc_max, -1.0, 1.0) * 10.0
            self.q = self.q + (self.dq * sp.dt_base + a * sp.dt_base ** 2 / 2)
            self.dq = self.dq + a * sp.dt_base
            reward = np.exp(-np.hypot(1.0 - self.q[:, 0], self.q[:, 1]))
            absorbing = np.zeros(self.B, dtype=bool)
        else:
            delay = getattr(sp, 'obs_delay', False)
            # the wrapper's q, dq are read off the observation the previous step (or the reset) returned
            # (atacom.py:95-96,111-112): with obs_delay the controller sees the FILTERED joint velocities
            q_ctl, dq_ctl = self.q.copy(), (self.fv[:, 3:].copy() if delay else self.dq.copy())
            q_sim, dq_sim = self.q.copy(), self.dq.copy()
            terms = constraint_terms(sp, q_ctl, dq_ctl)
            m0 = mallet_xy_world(sp, self.q)
            for k in range(sp.substeps):
                if delay:
                    # step_action_function calls env._create_observation(sim_state) in every sub-step (atacom.py:124): the
                    # low-pass advances on the velocities at the START of the sub-step (env_single.py:114-119)
                    self.fv[:, 3:] = OBS_DELAY_ALPHA * dq_sim + (1 - OBS_DELAY_ALPHA) * self.fv[:, 3:]
                if not sp.hold_q and k > 0:
                    q_ctl, dq_ctl = q_sim.copy(), (self.fv[:, 3:].copy() if delay else dq_sim.copy())
                    terms = constraint_terms(sp, q_ctl, dq_ctl)
                mu = self.tangent_space_accel(q_ctl, dq_ctl, self.s, alpha, terms)
                self.s = self.s + mu[:, nq:] * sp.dt
                ddq = self.acc_truncation(dq_ctl, mu[:, :nq])
                if sp.dynamics_mode >= 1:
                    ddq = self._rigid_body_substep(q_sim, dq_sim, ddq)
                dq_sim = np.clip(dq_sim + ddq * sp.dt, -1.5 * sp.vel_max, 1.5 * sp.vel_max)
                q_sim = q_sim + dq_sim * sp.dt
            self.q, self.dq = q_sim, dq_sim
            m1 = mallet_xy_world(sp, self.q)
            for k in range(sp.substeps):
                self._puck_substep(m0 + (m1 - m0) * ((k + 1) / sp.substeps), (m1 - m0) / (sp.substeps * sp.dt), k)
            absorbing = self._is_absorbing()
            reward = self._reward(alpha, absorbing)
            if delay:                                        # the observation the step returns (PyBullet.step [upstream])
                self.fv = OBS_DELAY_ALPHA * np.concatenate([self.puck[:, 3:6], self.dq], 1) + (1 - OBS_DELAY_ALPHA) * self.fv
            fun, _, _ = constraint_terms(sp, self.q, np.zeros_like(self.q))
            c_i = fun.copy()
            c_i[:, :sp.n_f] = np.abs(c_i[:, :sp.n_f])
            cm = c_i.max(-1)
            # _update_constraint_stats gets the wrapper's dq = the observation's (atacom.py:111-114)
            dq_seen = self.fv[:, 3:] if delay else self.dq
            self._log(cm, cm, (np.abs(dq_seen) - sp.vel_max).max(-1))
        self.t += 1
        return self.observation(), reward, absorbing, {}

    SERVO_GAIN = 0.1          # PyBullet's default positionGain of POSITION_CONTROL (env_base.py:64-70)
    SERVO_EFFORT = np.array([40.0, 10.0, 10.0])      # urdf/iiwa_1.urdf:297,384,400: joint_7, striker_joint_1 / _2

    def _rigid_body_substep(self, q_sim, dq_sim, ddq_des):
        """Row N4, dynamics_mode >= 1 (the model of this build where Bullet was), per sub-step:
          tau  = inverse dynamics of the nine-joint chain for [ddq_des, 0, 0, 0]   (acc_to_ctrl_action,
                 iiwa_hit_atacom.py:58-63); dynamics_mode 2: for [ddq_des, ddq_b] -- the controller knows what the servo
                 joints are about to do (feed-forward of their reaction; NOT what the reference computes)
          servo joints (joint 7, universal joint; POSITION_CONTROL, env_base.py:64-70): velocity set-point
                 v* = clip(gain (target - q) / dt, 1.5 v_max), targets from env_single.py:137-185; the motor realises
                 ddq_b = (v* - dq_b) / dt as far as its torque allows: |M_bb,ii ddq_b,i + h_b,i| <= URDF effort limit
                 (diagonal estimate of the motor torque; h = gravity + Coriolis at the simulated state)
          ddq_a = forward dynamics of the six controlled joints under tau, URDF joint damping and the servo joints'
                 accelerations.
        Advances the servo joints; returns ddq_a (the caller integrates the controlled joints)."""
        from . import dynamics as D
        sp = self.spec
        B = q_sim.shape[0]
        q9 = np.concatenate([q_sim, self.qx], 1)
        dq9 = np.concatenate([dq_sim, self.dqx], 1)
        tgt = np.concatenate([D.joint7_target(q_sim, self.qx[:, 0])[:, None],
                              D.universal_joint_target(q9[:, :7])], 1)
        vmax = 1.5 * np.array([robots.IIWA_VEL_LIMIT[6], 3.1415926, 3.1415926])     # iiwa_1.urdf:297,384,397
        vstar = np.clip(self.SERVO_GAIN * (tgt - self.qx) / sp.dt, -vmax, vmax)
        ddq_b = (vstar - self.dqx) / sp.dt
        h = D.rnea(q9, dq9, np.zeros((B, 9)))
        Mbb = np.einsum('bii->bi', D.mass_matrix(q9))[:, 6:]
        lim = np.maximum(self.SERVO_EFFORT - np.abs(h[:, 6:]), 0.0) / Mbb
        ddq_b = np.clip(ddq_b, -lim, lim)
        ff = ddq_b if sp.dynamics_mode == 2 else np.zeros((B, 3))
        tau = D.rnea(q9, dq9, np.concatenate([ddq_des, ff], 1))[:, :6]
        tau = np.clip(tau, -D.EFFORT_LIMIT, D.EFFORT_LIMIT)
        ddq_a = D.forward_dynamics(q9, dq9, tau, ddq_b)
        self.dqx = self.dqx + ddq_b * sp.dt
        self.qx = self.qx + self.dqx * sp.dt
        return ddq_a

    def _puck_substep(self, mallet, mallet_vel, k=0):
        """Batched version of atacom_scalar.ScalarAtacomEnv._puck_substep (contact model of this build, row N1)."""
        sp = self.spec
        pk = self.puck
        if getattr(sp, 'obs_delay', False):                  # the sub-step's _create_observation (see step)
            self.fv[:, :3] = OBS_DELAY_ALPHA * pk[:, 3:6] + (1 - OBS_DELAY_ALPHA) * self.fv[:, :3]
        if getattr(sp, 'env_noise', False):
            # _simulation_pre_step (env_base.py:176-180): force 0.0005 [randn, randn, 0] on the puck for this sub-step
            env = self.env_index
            dv = ENV_NOISE_FORCE * sp.dt / sp.puck_mass
            for c in range(2):
                pk[:, 3 + c] += dv * device_normal(self.seed, env, self.ep_cur, self.t, 3 + 2 * k + c, self.n_idx)
        pk[:, 0:3] += pk[:, 3:6] * sp.dt
        d = pk[:, 0:2] - mallet
        dist = np.hypot(d[:, 0], d[:, 1])
        R = PUCK_RADIUS + MALLET_RADIUS
        hit = dist < R
        safe = np.where(dist > 0, dist, 1.0)
        n = np.where((dist > 0)[:, None], d / safe[:, None], np.array([1.0, 0.0]))
        vrel = ((pk[:, 3:5] - mallet_vel) * n).sum(-1)
        imp = hit & (vrel < 0)
        self._cm(dist - R)
        self._cm(vrel, hit)
        pk[:, 3:5] = np.where(imp[:, None], pk[:, 3:5] - ((1 + E_MALLET) * vrel)[:, None] * n, pk[:, 3:5])
        pk[:, 0:2] = np.where(hit[:, None], mallet + n * R, pk[:, 0:2])
        ylim = TABLE_WIDTH / 2 - PUCK_RADIUS
        oy = np.abs(pk[:, 1]) > ylim
        self._cm(np.abs(pk[:, 1]) - ylim)
        self._cm(pk[:, 4], oy)
        sg = np.sign(pk[:, 1])
        pk[:, 1] = np.where(oy, sg * (2 * ylim - np.abs(pk[:, 1])), pk[:, 1])
        pk[:, 4] = np.where(oy & (pk[:, 4] * sg > 0), -E_RIM * pk[:, 4], pk[:, 4])
        xlim = TABLE_LENGTH / 2 - PUCK_RADIUS
        ox = (np.abs(pk[:, 0]) > xlim) & (np.abs(pk[:, 1]) >= GOAL_WIDTH)
        self._cm(np.abs(pk[:, 0]) - xlim)
        self._cm(np.abs(pk[:, 1]) - GOAL_WIDTH, np.abs(pk[:, 0]) > xlim)
        self._cm(pk[:, 3], ox)
        sg = np.sign(pk[:, 0])
        pk[:, 0] = np.where(ox, sg * (2 * xlim - np.abs(pk[:, 0])), pk[:, 0])
        pk[:, 3] = np.where(ox & (pk[:, 3] * sg > 0), -E_RIM * pk[:, 3], pk[:, 3])
        if self.defend:
            # AirHockeyDefend._simulation_post_step [upstream]: has_hit latches on puck / mallet contact, has_bounce on
            # contact with the end rims of the agent's side (t_down_rim_l / _r)
            self.has_hit |= hit
            self.has_bounce |= ox & (sg < 0)
            return
        v = np.hypot(pk[:, 3], pk[:, 4])
        self._cm(v - 0.1, ~self.has_hit)
        new_hit = (~self.has_hit) & (v > 0.1)
        self.vel_hit_x = np.where(new_hit, pk[:, 3], self.vel_hit_x)
        self.has_hit |= new_hit

    def _is_absorbing(self):
        sp = self.spec
        bnd = np.array([TABLE_LENGTH, TABLE_WIDTH]) / 2
        out = np.any(np.abs(self.puck[:, :2]) > bnd, -1)
        mal = np.abs(mallet_xy_world(sp, self.q)) - bnd
        out |= np.any(mal > 0.02, -1)
        spd = np.hypot(self.puck[:, 3], self.puck[:, 4])
        self._cm((np.abs(self.puck[:, :2]) - bnd).T[0]); self._cm((np.abs(self.puck[:, :2]) - bnd).T[1])
        self._cm(mal[:, 0] - 0.02); self._cm(mal[:, 1] - 0.02)
        if self.defend:             # AirHockeyDefend.is_absorbing [upstream]: hit or bounced, and back in the other half
            out |= (self.has_hit | self.has_bounce) & (self.puck[:, 0] > 0)
            self._cm(self.puck[:, 0], self.has_hit | self.has_bounce)
            return out
        out |= self.has_hit & (spd < 0.01)
        self._cm(spd - 0.01, self.has_hit)
        return out

    def _reward_defend(self, alpha, absorbing):
        """AirHockeyDefend.reward [upstream, restated from memory -- mushroom_rl is not in the reference tree]:
        absorbing: -50 if the puck is in the agent's goal, else 0; has_bounce: -1; has_hit: r_x + r_y + r_vel + 1 while the puck
        is in -0.8 < x < -0.4 (resting near x = -0.6, y = 0), else 0; before the hit: the mallet on the line x = -0.6 at the
        puck's y (0.3 exp(-3 |dx|) + 0.7 N(|dy| - 0.08; sigma 0.2) / 2)."""
        sp = self.spec
        pp, pv = self.puck[:, :2], self.puck[:, 3:5]
        conceded = (pp[:, 0] + TABLE_LENGTH / 2 < 0) & (np.abs(pp[:, 1]) - GOAL_WIDTH < 0)
        self._cm(np.abs(pp[:, 1]) - GOAL_WIDTH, absorbing & (pp[:, 0] + TABLE_LENGTH / 2 < 0))
        r_y = 3 * np.exp(-3 * np.abs(pp[:, 1]))
        r_x = np.exp(-5 * np.abs(pp[:, 0] + 0.6))
        r_vel = 5 * np.exp(-25 * (pv * pv).sum(-1))
        zone = (pp[:, 0] > -0.8) & (pp[:, 0] < -0.4)
        self._cm(pp[:, 0] + 0.8, self.has_hit & ~self.has_bounce); self._cm(pp[:, 0] + 0.4, self.has_hit & ~self.has_bounce)
        r_hit = np.where(zone, r_x + r_y + r_vel + 1, 0.0)
        ee = mallet_xy_world(sp, self.q)
        ex, ey = np.abs(-0.6 - ee[:, 0]), np.abs(pp[:, 1] - ee[:, 1])
        sig = 0.2
        r_app = 0.3 * np.exp(-3 * ex) + 0.7 * (np.exp(-((ey - 0.08) / sig) ** 2 / 2) / (np.sqrt(2 * np.pi) * sig) / 2)
        r = np.where(absorbing, np.where(conceded, -50.0, 0.0),
                     np.where(self.has_bounce, -1.0, np.where(self.has_hit, r_hit, r_app)))
        return r - sp.action_penalty * np.sqrt((alpha * alpha).sum(-1))

    def _reward(self, alpha, absorbing):
        if self.defend:
            return self._reward_defend(alpha, absorbing)
        sp = self.spec
        pp = self.puck[:, :2]
        goal = (pp[:, 0] - TABLE_LENGTH / 2 > 0) & (np.abs(pp[:, 1]) - GOAL_WIDTH < 0)
        self._cm(np.abs(pp[:, 1]) - GOAL_WIDTH, absorbing & (pp[:, 0] - TABLE_LENGTH / 2 > 0))
        ee = mallet_xy_world(sp, self.q)
        d = pp - ee
        dist = np.sqrt((d * d).sum(-1))
        g = GOAL_POS - pp
        gn = np.sqrt((g * g).sum(-1))
        cosang = np.clip(((g / gn[:, None]) * (d / dist[:, None])).sum(-1), 0, 1)
        r_app = np.exp(-8 * (dist - 0.08)) * cosang
        upd = (~absorbing) & (~self.has_hit)
        self.r_hit = np.where(upd, r_app, self.r_hit)
        r = np.where(absorbing, np.where(goal, 80.0, 0.0),
                     np.where(self.has_hit, 1 + self.r_hit + self.vel_hit_x * 0.1, r_app))
        return r - sp.action_penalty * np.sqrt((alpha * alpha).sum(-1))

    def _log(self, c_for_avg, c_for_max, dq_for_max):
        self.stat_sum += c_for_avg
        self.stat_cnt += 1
        self.stat_cmax = np.maximum(self.stat_cmax, c_for_max)
        self.stat_dqmax = np.maximum(self.stat_dqmax, dq_for_max)

    def get_constraints_logs(self, clear=True):
        out = (float(self.stat_sum.sum() / max(self.stat_cnt.sum(), 1)), float(self.stat_cmax.max()),
               float(self.stat_dqmax.max()))
        if clear:
            self.stat_sum[:] = 0
            self.stat_cnt[:] = 0
            self.stat_cmax[:] = -np.inf
            self.stat_dqmax[:] = -np.inf
        return out
