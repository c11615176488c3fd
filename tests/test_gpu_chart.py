"""GPU parity tests of the opt-in canonical chart (cfg.chart_mode = 1), through the C ABI.

Parity definition, in three parts:
  (1) HIP == its float64 specification (oracle/canonical_chart.py) on EVERY sample: float64 build 1e-8; float32 build by
      the sensitivity rule of tests/parity_tools.py (every sample within C x the oracle's own response to float32-sized
      perturbations of the same inputs) -- primitive and whole env step, all four kernel mappings;
  (2) == the REFERENCE's chart (the default mode's oracle, pinned to the reference's golden vectors) on every sample where
      the reference's rref takes no tolerance branch and the canonical chart stays on the default chart too (the CPU test
      tests/test_oracle_chart.py bounds how often the second condition fails: < 2 % of the clear samples);
  (3) invariants on ALL samples, computed from the HIP outputs: Jc mu + y = 0 and Jc N = 0 (the reference's N_c leaks by
      up to O(10) in its tolerance branch), and a closed loop that keeps the constraints at least as well.
"""
import copy

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from chart_cases import rollout_systems, degenerate_systems, jc_of, init_q, SPECS          # noqa: E402
from oracle import atacom_batched as ob                                                     # noqa: E402
from oracle import canonical_chart as cc                                                    # noqa: E402

DEV = 'cuda:0'
DT = {'f64': torch.float64, 'f32': torch.float32}


def _dev_mu(name, dt, A, s, y, alpha, tol=0.05):
    from rl_on_manifold_amd import canonical_mu
    t = lambda x: torch.tensor(x, device=DEV, dtype=DT[dt])            # noqa: E731
    return canonical_mu(name, t(A), t(s), t(y), t(alpha), tol=tol).double().cpu().numpy()


@pytest.mark.parametrize('dt', ['f64', 'f32'])
@pytest.mark.parametrize('name', ['circle', 'planar', 'iiwa'])
def test_canonical_mu_primitive_against_its_specification(name, dt):
    """atacom_canonical_mu vs oracle.canonical_chart.canonical_mu on the systems the rollouts visit (chart switches, slack
    coordinates included) and on hand-made degenerate ones; invariants from the DEVICE outputs on all of them."""
    from parity_tools import C_SENS, FLOOR, QUICK_SCALES, DEEP_SCALES
    for sy, hand_made in ((rollout_systems(name), False), (degenerate_systems(name), True)):
        spec = sy['spec']
        nf, k = spec.n_f, spec.n_null
        A, s, y = sy['A'], sy['s'], sy['y']
        if not hand_made:
            A, s, y = A[:6000], s[:6000], y[:6000]
        n = len(A)
        rng = np.random.default_rng(4)
        alpha = rng.uniform(-10, 10, (n, k))
        ref = cc.canonical_mu(A, s, y, alpha, spec.rref_tol, nf)
        dev = _dev_mu(name, dt, A, s, y, alpha)
        assert np.isfinite(dev).all()
        scale = np.maximum(1.0, np.abs(ref).max(1))
        err = np.abs(dev - ref).max(1) / scale
        if dt == 'f64':
            # the hand-made systems include decisions taken on exact zeros, where two correct float64 evaluations may
            # differ; the rollout systems must agree everywhere
            assert (np.quantile(err, 0.97) if hand_made else err.max()) < 1e-8, (err.max(), np.quantile(err, 0.97))
        else:
            def sens(idx, scales, draws):
                out = np.zeros(len(idx))
                for sc in scales:
                    for _ in range(draws):
                        p = lambda x: x[idx] * (1.0 + sc * rng.choice([-1.0, 1.0], x[idx].shape))      # noqa: E731
                        o = cc.canonical_mu(p(A), p(s), p(y), p(alpha), spec.rref_tol, nf)
                        out = np.maximum(out, np.abs(o - ref[idx]).max(1) / scale[idx])
                return out
            S = sens(np.arange(n), QUICK_SCALES, 2)
            bad = np.nonzero(err > C_SENS * S + FLOOR)[0]
            if len(bad):
                S[bad] = np.maximum(S[bad], sens(bad, DEEP_SCALES, 48))
            still = bad[err[bad] > C_SENS * S[bad] + FLOOR]
            msg = '%s %s: %d systems, err median %.2e p99 %.2e max %.2e, %d deep, %d unexplained' % (
                name, 'hand-made' if hand_made else 'rollout', n, np.median(err), np.quantile(err, 0.99), err.max(), len(bad),
                len(still))
            if hand_made and len(still):
                msg += ' (kinds %s)' % sorted(sy['kind'][still].tolist())
            print(msg)
            # hand-made systems: SEVERAL slacks below theta at once (kinds 2, 4) are the float32 build's weak spot -- only the
            # first stiff row's slack is carried as a coordinate, the others keep the division by s;
            # a bounded share of those may stay unexplained.  Every system the rollouts visit must be explained.
            assert len(still) <= (0.2 * n if hand_made else 0), msg
            if hand_made and len(still):
                assert set(sy['kind'][still].tolist()) <= {2, 4, 5}, msg
            assert np.median(err) < 2e-5
        # invariants from the device outputs, every system
        Jc = jc_of(A, s, nf)
        live = np.abs(Jc).max(2) > 0
        if hand_made:
            live &= (sy['kind'] != 5)[:, None]
        res = np.abs(np.einsum('bcn,bn->bc', Jc, dev) + y) * live
        rel = res.max(1) / np.maximum(1.0, np.abs(y).max(1))
        cols = [_dev_mu(name, dt, A, s, np.zeros_like(y), np.eye(k)[i][None].repeat(n, 0)) for i in range(k)]
        N = np.stack(cols, 2)
        leak = np.abs(np.einsum('bcn,bnk->bck', Jc, N)).max((1, 2)) / np.maximum(1.0, np.abs(N).max((1, 2)))
        tol_eq = (2e-4 if hand_made else 1e-8) if dt == 'f64' else 2e-3
        assert rel.max() < tol_eq and leak.max() < tol_eq, (rel.max(), leak.max())
        if dt == 'f32':
            assert np.median(rel) < 1e-5 and np.median(leak) < 1e-5


def _step_outputs(p, inputs):
    oo, orr, oab, _ = p.step(inputs[0])
    return np.concatenate([oo, p.s, orr[:, None], oab[:, None].astype(np.float64)], 1)


_TF = {}


def _teacher_forced(name, B, T):
    """Oracle side (canonical chart), shared by the kernel mappings: states, actions, outputs + sensitivities, and the
    REFERENCE-chart oracle's outputs of the same steps with its 'took a tolerance branch' flag."""
    if name not in _TF:
        from parity_tools import SensitivityRecorder, slice_env
        spec = SPECS[name]()
        spec.chart_mode = 1
        rng = np.random.default_rng(21)
        o = ob.BatchedAtacomEnv(spec, B, init_q=init_q(name, B, rng))
        rec = SensitivityRecorder(_step_outputs, seed=6)
        states, acts, ref_out, clear = [], [], [], []
        for t in range(T):
            a = rng.uniform(-1.3, 1.3, (B, spec.n_null))
            a[: B // 8] = np.sign(a[: B // 8])
            acts.append(a)
            states.append((o.q.copy(), o.dq.copy(), o.s.copy(), o.puck.copy(), o.has_hit.copy(), o.r_hit.copy(),
                           o.vel_hit_x.copy(), o.t.copy()))
            # the same step under the reference's chart
            r = slice_env(o, np.arange(B))
            r.spec = copy.deepcopy(spec)
            r.spec.chart_mode = 0
            r.track_margins()
            ref_out.append(_step_outputs(r, (a,)))
            c = slice_env(o, np.arange(B))
            c.track_margins()
            _step_outputs(c, (a,))
            clear.append(~r.chart_skipped & c.chart_default)
            rec.prepare(o, (a,))
            o.step(a)
            last = o.t >= spec.horizon
            if last.any():
                o.reset(last)
        _TF[name] = (spec, rec, states, acts, np.array(ref_out), np.array(clear))
    return _TF[name]


@pytest.mark.parametrize('name', ['circle', 'planar', 'iiwa'])
def test_canonical_mu_primitive_against_the_references_own_outputs(name):
    """Golden set G12 through the C ABI: atacom_canonical_mu (float64 kernel) on J_c systems for which the imported
    reference's pinv_null + rref(tol = 0.05) + atacom.py:127-133 were recorded.  Wherever the reference zeroed nothing and
    its reduced echelon basis sits on the same free coordinates, the device's mu IS the reference's (1e-8); the free
    coordinates are read off the device's own null basis (unit rows), not taken from the oracle."""
    from chart_cases import reference_golden
    g = reference_golden(name)
    A, s, y, alpha = g['A'], g['s'], g['y'], g['alpha']
    n, k = alpha.shape
    dev = _dev_mu(name, 'f64', A, s, y, alpha)
    N = np.stack([_dev_mu(name, 'f64', A, s, np.zeros_like(y), np.eye(k)[i][None].repeat(n, 0)) for i in range(k)], 2)
    assert np.abs(np.einsum('bcn,bnk->bck', g['Jc'], N)).max() < 1e-8          # the device's basis: exact on every system
    free = np.full((n, k), -1)
    for b in range(n):
        for i in range(k):
            rows = [r for r in range(N.shape[1]) if abs(N[b, r, i] - 1.0) < 1e-12 and (np.abs(N[b, r]) > 1e-12).sum() == 1]
            free[b, i] = rows[0] if rows else -1
    same = g['exact'] & (free == g['free']).all(1)
    assert same.mean() > {'circle': 0.03, 'planar': 0.9, 'iiwa': 0.5}[name], same.mean()
    err = np.abs(dev - g['mu']).max(1) / np.maximum(1.0, np.abs(g['mu']).max(1))
    print('%s: %d of %d systems comparable, max err %.2e' % (name, same.sum(), n, err[same].max()))
    assert err[same].max() < 1e-8


@pytest.mark.parametrize('lanes', [1, 2, 4, 8])
@pytest.mark.parametrize('dt', ['f64', 'f32'])
@pytest.mark.parametrize('name', ['circle', 'planar', 'iiwa'])
def test_env_step_canonical_chart(name, dt, lanes):
    """One atacom_step in chart_mode 1 from injected states, 512 x 30 per environment, every kernel mapping:
    (1) vs the canonical oracle on every sample (float64 1e-8, float32 sensitivity rule);
    (2) vs the REFERENCE-chart oracle wherever its rref took no tolerance branch (float64: 1e-8)."""
    from rl_on_manifold_amd import BatchedAtacomEnv
    B, T = 512, 30
    spec, rec0, states, acts, ref_out, clear = _teacher_forced(name, B, T)
    rec = rec0.fresh()
    env = BatchedAtacomEnv(name, B, device=DEV, dtype=DT[dt], lanes_per_env=lanes, chart_mode='canonical')
    assert env.lanes_per_env == lanes
    nq, ng = spec.dim_q, spec.n_g
    worst_ref = 0.0
    for t in range(T):
        q, dq, s, puck, has_hit, r_hit, vhx, tt = states[t]
        full = np.zeros((B, env.state_dim))
        full[:, :nq], full[:, nq:2 * nq], full[:, 2 * nq:2 * nq + ng] = q, dq, s
        full[:, 2 * nq + ng:2 * nq + ng + 6] = puck
        full[:, 2 * nq + ng + 6], full[:, 2 * nq + ng + 7], full[:, 2 * nq + ng + 8], full[:, -1] = has_hit, r_hit, vhx, tt
        env.set_state(full)
        obs, r, ab, info = env.step(acts[t])
        s_dev = env.get_state().cpu().numpy()[:, 2 * nq:2 * nq + ng]
        dev = np.concatenate([obs.cpu().numpy(), s_dev, r.cpu().numpy()[:, None], ab.cpu().numpy()[:, None] * 1.0], 1)
        rec.compare(t, dev)
        if dt == 'f64' and clear[t].any():
            e = np.abs(dev - ref_out[t]) / np.maximum(1.0, np.abs(ref_out[t]))
            worst_ref = max(worst_ref, e[clear[t]].max())
    if dt == 'f64':
        assert np.max(rec.err) < 1e-8, np.max(rec.err)
        assert worst_ref < 1e-8, worst_ref
        assert clear.mean() > {'circle': 0.01, 'planar': 0.7, 'iiwa': 0.3}[name], clear.mean()
        print('%s: reference-chart parity on %.1f %% of the samples (the rest: the reference takes its tolerance branch)'
              % (name, 100 * clear.mean()))
    else:
        print(rec.finish('%s canonical, lanes %d' % (name, lanes), max_vacuous={'circle': 0.0, 'planar': 0.01, 'iiwa': 0.10}[name]))      # measured 6.6 % (round 4)


@pytest.mark.parametrize('name', ['planar', 'iiwa'])
def test_canonical_rollout_kernels_equal_the_step_kernel(name):
    """k_rollout / k_rollout_mlp in chart_mode 1 == repeated atacom_step (bitwise per mapping), packed records too."""
    from rl_on_manifold_amd import BatchedAtacomEnv, MlpPolicy
    B, T = 200, 12
    for lanes in (1, 4, 8):
        env = BatchedAtacomEnv(name, B, device=DEV, lanes_per_env=lanes, chart_mode='canonical', auto_reset=True, horizon=7)
        k, D = env.dims['null'], env.obs_dim
        g = torch.Generator(device=DEV).manual_seed(3)
        acts = torch.rand((T, B, k), device=DEV, generator=g) * 2.4 - 1.2
        st = env.get_state().clone()
        ref = env.rollout(acts)
        env.set_state(st)
        for t in range(T):
            obs, r, ab, info = env.step(acts[t])
            assert torch.isfinite(r).all() and torch.isfinite(obs).all(), (lanes, t)
            assert torch.equal(obs, ref['next_obs'][t]), (lanes, t, float((obs - ref['next_obs'][t]).abs().max()))
            assert torch.equal(r, ref['reward'][t]), (lanes, t, float((r - ref['reward'][t]).abs().max()))
            assert torch.equal(info['last'], ref['last'][t].bool())
        # the policy kernel: same network evaluated on the host, then fed to the action kernel
        gw = torch.Generator().manual_seed(0)
        pol = MlpPolicy(torch.randn(64, D, generator=gw) * 0.2, torch.zeros(64), torch.randn(64, 64, generator=gw) * 0.1,
                        torch.zeros(64), torch.randn(k, 64, generator=gw) * 0.1, torch.zeros(k), std=torch.full((k,), 0.3))
        eps = torch.randn((T, B, k), device=DEV, generator=g)
        env.set_state(st)
        out = env.rollout_policy(pol, T, noise=eps)
        env.set_state(st)
        again = env.rollout(out['action'])
        assert torch.allclose(again['next_obs'], out['next_obs'], atol=1e-5) and torch.isfinite(out['reward']).all()


def test_canonical_chart_closed_loop_at_config_4():
    """8192 iiwa environments x 120 steps from the feasible initial states of BASELINE config 4, float32, free-running:
    the canonical chart keeps the constraints at least as well as the reference's chart (same states, same actions) and
    never exceeds a velocity limit."""
    import bench
    from rl_on_manifold_amd import BatchedAtacomEnv
    B, T = 8192, 120
    gen = torch.Generator(device=DEV); gen.manual_seed(0)
    init = bench.feasible_init('iiwa', B, torch.device(DEV), gen)[0]
    acts = torch.rand((T, B, 5), device=DEV, generator=gen) * 2 - 1
    stats = {}
    for mode in ('reference', 'canonical'):
        for lanes in (1, 2, 4, 8):
            env = BatchedAtacomEnv('iiwa', B, device=DEV, chart_mode=mode, auto_reset=True, lanes_per_env=lanes)
            env.reset(state=init)
            out = env.rollout(acts)
            assert torch.isfinite(out['obs']).all() and torch.isfinite(out['reward']).all()
            stats[mode, lanes] = env.get_constraints_logs()
    col = lambda mode, i: np.array([stats[mode, l][i] for l in (1, 2, 4, 8)])
    for mode in ('reference', 'canonical'):
        print('%-9s chart, 1 / 2 / 4 / 8 lanes: c_avg %s  c_max %s  c_dq_max %s' % (mode, col(mode, 0).round(5), col(mode, 1).round(4),
                                                                                 col(mode, 2)))
    # c_max is the largest of 1e6 env-steps on float32 trajectories that part ways at the first rounding difference: a
    # heavy-tailed statistic -- the REFERENCE chart's own four kernel mappings spread over 0.010 ... 0.033 on these states
    # (profiles/r03_chart_closed_loop.log) -- so the charts are compared by the median over the mappings (at the 1.5x the
    # other free-running tests use) and every run against the absolute bound; the mean is the stable number
    a0, a1 = col('reference', 0), col('canonical', 0)
    assert (a1 <= 1.05 * a0.max()).all(), stats
    assert np.median(col('canonical', 1)) <= 1.5 * np.median(col('reference', 1)), stats
    m1, d1 = col('canonical', 1).max(), col('canonical', 2).max()
    assert d1 <= 1e-4 and m1 < 0.05


@pytest.mark.parametrize('lanes', [1, 2, 4, 8])
@pytest.mark.parametrize('dt', ['f64', 'f32'])
@pytest.mark.parametrize('name', ['planar', 'iiwa'])
def test_canonical_chart_with_refreshed_state_and_exact_bias(name, dt, lanes):
    """The HOLD = false instantiations of the canonical-chart kernels (hold_q = 0: q, dq and with them A = K J change in
    every sub-step, so the row-slot copies of the group kernels are rebuilt per sub-step instead of once per step) with
    bias_mode = exact, against the canonical oracle under the same flags: float64 1e-8 on every sample, float32 close on
    the bulk (the per-sample float32 rule runs in test_env_step_canonical_chart)."""
    import dataclasses
    from oracle import atacom_scalar as osc
    from rl_on_manifold_amd import BatchedAtacomEnv
    from test_gpu_parity import _full_state
    spec = {'planar': osc.planar_spec, 'iiwa': osc.iiwa_spec}[name](bias_mode='exact')
    spec = dataclasses.replace(spec, chart_mode=1, hold_q=False)
    B, T = 384, 12
    env = BatchedAtacomEnv(name, B, device=DEV, dtype=DT[dt], lanes_per_env=lanes, chart_mode='canonical', hold_q=False,
                           bias_mode='exact')
    nq = spec.dim_q
    rng = np.random.default_rng(17)
    q0 = env.get_state().cpu().numpy().astype(np.float64)[:, :nq] + rng.normal(0, 0.05, (B, nq))
    o = ob.BatchedAtacomEnv(spec, B, init_q=q0, init_puck=np.array([0.8, 0.4, 0, 0, 0, 0.0]))
    errs = []
    for t in range(T):
        a = rng.uniform(-1.2, 1.2, (B, spec.n_null))
        env.set_state(_full_state(env, o))
        obs, r, ab, _ = env.step(a)
        oo, orr, oab, _ = o.step(a)
        e = np.maximum(np.abs(obs.double().cpu().numpy() - oo).max(1), np.abs(r.double().cpu().numpy() - orr))
        errs.append(e)
    errs = np.concatenate(errs)
    if dt == 'f64':
        assert errs.max() < 1e-8, errs.max()
    else:
        assert np.median(errs) < 2e-5 and np.quantile(errs, 0.99) < 5e-3, (np.median(errs), np.quantile(errs, 0.99))
    # and the variant really differs from the held one
    held = BatchedAtacomEnv(name, B, device=DEV, dtype=DT[dt], lanes_per_env=lanes, chart_mode='canonical')
    held.set_state(_full_state(held, o)); env.set_state(_full_state(env, o))
    a = rng.uniform(-1.0, 1.0, (B, spec.n_null))
    assert (held.step(a)[0] - env.step(a)[0]).abs().max() > 1e-6
