"""GPU parity tests proper: the HIP path, called through the C ABI, against (i) golden vectors captured from
the reference and (ii) the float64 oracle on identical seeded inputs.

Stated tolerances
  float64 build : identical algorithm in double -> 1e-8 abs on states / observations (observed ~1e-12).
  float32 build : (production) EVERY sample of every step test within  C x sens + floor, sens = the float64 oracle's own
                  response to float32-sized perturbations of the same inputs (tests/parity_tools.py: C = 4, floor =
                  5e-6 relative to max(1, |value|); the bulk sits at 1e-6).  No percentage of samples is exempt: the
                  reference's discontinuities (the 0.05 rref pivot tolerance, atacom.py:128; contact decisions) and
                  its 1/s slack dynamics show up in sens, a kernel bug does not.
                  Kinematics primitives: fixed absolute bounds stated in the tests.
"""
import os

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from oracle import atacom_scalar as osc          # noqa: E402
from oracle import atacom_batched as ob          # noqa: E402

DEV = 'cuda:0'
SPECS = {'circle': osc.circle_spec, 'planar': osc.planar_spec, 'iiwa': osc.iiwa_spec}
SHAPES = {'circle': (2, 3, 1), 'planar': (6, 9, 3), 'iiwa': (12, 17, 5)}
DT = {'f64': torch.float64, 'f32': torch.float32}


def _env(name, B, dt, **kw):
    from rl_on_manifold_amd import BatchedAtacomEnv
    return BatchedAtacomEnv(name, B, device=DEV, dtype=DT[dt], **kw)


def _full_state(env, o):
    """oracle env -> [B, state_dim] array for atacom_set_state."""
    nq, ng = o.spec.dim_q, o.spec.n_g
    full = np.zeros((o.B, env.state_dim if env is not None else 2 * nq + ng + 10))
    full[:, :nq], full[:, nq:2 * nq], full[:, 2 * nq:2 * nq + ng] = o.q, o.dq, o.s
    full[:, 2 * nq + ng:2 * nq + ng + 6] = o.puck
    full[:, 2 * nq + ng + 6] = o.has_hit
    full[:, 2 * nq + ng + 7], full[:, 2 * nq + ng + 8] = o.r_hit, o.vel_hit_x
    full[:, -1] = o.t
    return full


@pytest.mark.parametrize('lanes', [1, 2, 4, 8])
@pytest.mark.parametrize('dt', ['f64', 'f32'])
@pytest.mark.parametrize('name', ['circle', 'planar', 'iiwa'])
def test_nullspace_against_reference_golden(golden, name, dt, lanes):
    """A5 + A6: pinv_null and rref(tol=0.05) of the reference (LAPACK SVD) vs the HIP bidiagonalisation, through
    both kernel mappings (one lane per matrix / one DPP quad per matrix)."""
    from rl_on_manifold_amd import nullspace
    g = golden('nullspace')
    c, n, k = SHAPES[name]
    Jc = torch.tensor(g[name + '_Jc'], device=DEV, dtype=DT[dt])
    rhs = torch.tensor(np.tile(np.arange(1, c + 1, dtype=float), (len(Jc), 1)), device=DEV, dtype=DT[dt])
    x, nb, rr = nullspace(name, Jc, rhs, tol=0.05, lanes_per_env=lanes)
    x, nb, rr = x.cpu().numpy(), nb.cpu().numpy(), rr.cpu().numpy()
    xr = np.einsum('bnc,bc->bn', g[name + '_pinv'], rhs.cpu().numpy().astype(np.float64))
    scale = np.maximum(1.0, np.abs(xr).max(-1, keepdims=True))
    nref = g[name + '_null']
    if k == 1:
        nb = nb * np.sign((nb * nref).sum(1, keepdims=True))
    rscale = np.maximum(1.0, np.abs(g[name + '_rref']).reshape(len(Jc), -1).max(-1))
    rerr = np.abs(rr - g[name + '_rref']).reshape(len(Jc), -1).max(-1) / rscale
    if dt == 'f64':
        assert (np.abs(x - xr) / scale).max() < 1e-9
        assert np.abs(nb - nref).max() < 1e-10          # the SAME orthonormal basis LAPACK returns
        assert rerr.max() < 1e-9
    else:
        from parity_tools import assert_matrix_fn_explained
        assert (np.abs(x - xr) / scale).max() < 5e-3
        assert np.abs(nb - nref).max() < 2e-3
        # the chart: every matrix within C x the float64 chart's own response to float32-sized perturbations of Jc
        Jc64 = g[name + '_Jc']
        ok = np.array([np.linalg.matrix_rank(m) == c for m in Jc64])      # full-rank inputs (the rank-deficient golden
        chart = lambda A: ob.rref_tol(ob.bidiag_solve_null(A, np.zeros((len(A), c)), k)[1], 0.05)   # noqa: E731
        assert np.abs(chart(Jc64[ok]) - g[name + '_rref'][ok]).max() < 1e-9                     # cases: test below)
        print(assert_matrix_fn_explained(chart, Jc64[ok], rr[ok], 'chart %s lanes %d' % (name, lanes)))


@pytest.mark.parametrize('dt', ['f64', 'f32'])
@pytest.mark.parametrize('bias', ['reference', 'exact'])
@pytest.mark.parametrize('name', ['circle', 'planar', 'iiwa'])
def test_constraint_terms_against_oracle(name, bias, dt):
    """A9-A11: fun / J / b callables (forward kinematics, frame Jacobians, bias) vs the float64 oracle."""
    from rl_on_manifold_amd import constraint_terms
    spec = SPECS[name]() if name == 'circle' else SPECS[name](bias_mode=bias)
    rng = np.random.default_rng(3)
    n = 777                                           # ragged: not a multiple of the wavefront
    q = rng.uniform(-1.5, 1.5, (n, spec.dim_q))
    dq = rng.uniform(-1.5, 1.5, (n, spec.dim_q))
    q[0] = 0.0                                        # singular / symmetric poses: exact zeros in J
    if name == 'iiwa':
        q[1] = [0.0, 0.7135214629060707, 0.0, -0.5024756033561426, 0.0, 1.9256631778550268]
    fun, J, b = constraint_terms(name, torch.tensor(q, device=DEV, dtype=DT[dt]),
                                 torch.tensor(dq, device=DEV, dtype=DT[dt]), bias_mode=bias)
    fo, Jo, bo = ob.constraint_terms(spec, q, dq)
    tol = 1e-11 if dt == 'f64' else 2e-5
    assert np.abs(fun.cpu().numpy() - fo).max() < tol * 10
    assert np.abs(J.cpu().numpy() - Jo).max() < tol * 10
    assert np.abs(b.cpu().numpy() - bo).max() < tol * 50
    # structurally zero Jacobian entries are (numerically) zero on both sides -- they steer LAPACK-style sign
    # choices in the bidiagonalisation.  (Sample 0, the all-zero singular pose, is exempt: noise on both sides.)
    zd, zo = np.abs(J.cpu().numpy()) < 1e-15, np.abs(Jo) < 1e-15
    if dt == 'f64':
        assert (zd == zo)[1:].all()
    assert (J.cpu().numpy()[1:][(Jo == 0)[1:]] == 0).all()       # exact zeros of the oracle are exact on the device


@pytest.mark.parametrize('dt', ['f64', 'f32'])
@pytest.mark.parametrize('bias', ['reference', 'exact'])
def test_iiwa_constraint_terms_against_the_reference_urdf(golden, bias, dt):
    """A9 pinned to a reference-held file: golden set G10 is the reference's own urdf/iiwa_1.urdf run through the generic
    URDF evaluator (oracle/gen_golden.py urdf); the HIP callables must reproduce the constraint values, Jacobians and
    bias terms the reference's formulas (iiwa_hit_atacom.py:70-139) give on those frames.
    Tolerances: f64 1e-10; f32 3e-6 on values / Jacobian entries (O(1) m, 7 chained rotations at eps = 6e-8) and 2e-5
    on the bias terms (products of two velocities up to ~2.4 rad/s each)."""
    from rl_on_manifold_amd import constraint_terms
    from test_oracle_urdf import expected_terms
    g = golden('iiwa_urdf')
    fun_e, J_e, b_e = expected_terms(g, bias)
    fun, J, b = constraint_terms('iiwa', torch.tensor(g['q'], device=DEV, dtype=DT[dt]),
                                 torch.tensor(g['dq'], device=DEV, dtype=DT[dt]), bias_mode=bias)
    tol = (1e-10, 1e-10, 1e-10) if dt == 'f64' else (3e-6, 3e-6, 2e-5)
    assert np.abs(fun.cpu().numpy() - fun_e).max() < tol[0]
    assert np.abs(J.cpu().numpy() - J_e).max() < tol[1]
    assert np.abs(b.cpu().numpy() - b_e).max() < tol[2]


def _step_outputs(p, inputs):
    """Everything a teacher-forced env step is compared on: observation, slack, reward, absorbing flag."""
    oo, orr, oab, _ = p.step(inputs[0])
    return np.concatenate([oo, p.s, orr[:, None], oab[:, None].astype(np.float64)], 1)


_TF_CACHE = {}


def _teacher_forced_reference(name, B, T):
    """The oracle side of the teacher-forced step test: a free-running float64 trajectory (states, actions, outputs) with
    the sensitivity of every step (tests/parity_tools.py).  Computed once per environment, shared by all kernel mappings."""
    if name not in _TF_CACHE:
        from parity_tools import SensitivityRecorder
        spec = SPECS[name]()
        nq = spec.dim_q
        rng = np.random.default_rng(11)
        init_q = {'circle': np.array([-1.0, 0.0]), 'planar': ob.robots.PLANAR_INIT_Q, 'iiwa': IIWA_INIT_Q}[name]
        init_q = init_q + (rng.normal(0, 0.05, (B, nq)) if name != 'circle' else 0.0)
        o = ob.BatchedAtacomEnv(spec, B, init_q=init_q)
        rec = SensitivityRecorder(_step_outputs, seed=5)
        states, acts = [], []
        for t in range(T):
            a = rng.uniform(-1.3, 1.3, (B, spec.n_null))
            a[: B // 8] = np.sign(a[: B // 8])             # saturated actions push against the limits
            states.append(o)                               # placeholder, replaced below by the full-state rows
            acts.append(a)
            rec.prepare(o, (a,))
            states[-1] = (o.q.copy(), o.dq.copy(), o.s.copy(), o.puck.copy(), o.has_hit.copy(), o.r_hit.copy(),
                          o.vel_hit_x.copy(), o.t.copy())
            o.step(a)
        _TF_CACHE[name] = (spec, rec, states, acts, o.get_constraints_logs())
    return _TF_CACHE[name]


IIWA_INIT_Q = np.array([0.0, 0.7135214629060707, 0.0, -0.5024756033561426, 0.0, 1.9256631778550268])


def _device_chart_decisions(name, lanes):
    """Jc [b, c, n] (float64) -> the pivot / skip pattern the device's FLOAT32 rref takes on these matrices, read off the
    atacom_nullspace primitive of the kernel mapping under test (tests/parity_tools.skip_pattern_of_rref)."""
    from rl_on_manifold_amd import nullspace
    from parity_tools import skip_pattern_of_rref

    def follow(Jc):
        J32 = torch.tensor(Jc, dtype=torch.float32, device=DEV)
        return skip_pattern_of_rref(nullspace(name, J32, lanes_per_env=lanes)[2].double().cpu().numpy(), tol=1e-4)
    return follow


MIN_SAME_DECISIONS = 0.99      # share of the chart evaluations behind the vacuous samples on which the float32 device takes
                               # the float64 oracle's own pivot / skip pattern (asserted, not printed)


def _followed_chart_report(rec, name, lanes, tol=1e-4, max_outside=0.07, p99_max=3e-4):
    """Where the sensitivity bound is vacuous (the reference's tolerance regime), the device is compared with the float64
    oracle FORCED ONTO THE DEVICE'S OWN CHART DECISIONS: discrete disagreement removed, what is left
    is arithmetic.  First measurement (1024 x 40 states around the reset pose, one environment per lane, 12960 samples with a
    vacuous bound): median 5e-6, p90 5e-5, p99 3e-4, 4.1 % above 1e-4 -- and EXACTLY the same numbers against the plain
    oracle: on these samples the float32 device and the float64 oracle already take the same pivot / skip decisions; what
    makes the bound vacuous is the continuous sensitivity of LAPACK's null basis inside the tolerance branch (zeroing
    basis-dependent entries), not a discrete flip.  Round 5 ASSERTS that claim instead of printing it: the
    device's whole pivot / skip pattern equals the oracle's own on >= 99 % of the chart evaluations behind these samples
    (one per physics sub-step), and the residual against the oracle forced onto the device's decisions is bounded on top of the
    per-sample rule: at most 7 % of these samples above 1e-4, p99 below `p99_max`, median at rounding level."""
    from parity_tools import followed_chart_errors
    dec = {}
    n, e_f, e_p = followed_chart_errors(rec, _device_chart_decisions(name, lanes), decisions=dec)
    if n == 0:
        return 'no vacuous samples'
    same = dec['same'] / max(dec['total'], 1)
    msg = ('%s lanes %d, %d samples with a vacuous bound: device pivot / skip pattern == the oracle\'s own on %.3f %% of %d chart '
           'evaluations; against the oracle on the device\'s chart decisions median %.2e / p90 '
           '%.2e / p99 %.2e / max %.2e, %.2f %% above %.0e (against the plain oracle: median %.2e, p99 %.2e, %.2f %% above)'
           % (name, lanes, n, 100 * same, dec['total'], np.median(e_f), np.quantile(e_f, 0.9), np.quantile(e_f, 0.99), e_f.max(),
              100 * np.mean(e_f > tol), tol, np.median(e_p), np.quantile(e_p, 0.99), 100 * np.mean(e_p > tol)))
    assert same >= MIN_SAME_DECISIONS, msg
    assert np.mean(e_f > tol) <= max_outside and np.quantile(e_f, 0.99) < p99_max and np.median(e_f) < 2e-5, msg
    return msg


@pytest.mark.parametrize('lanes', [1, 2, 4, 8])
@pytest.mark.parametrize('dt', ['f64', 'f32'])
@pytest.mark.parametrize('name', ['circle', 'planar', 'iiwa'])
def test_env_step_teacher_forced_against_oracle(name, dt, lanes):
    """A1-A8, A12-A15 end to end: one atacom_step from identical injected states, 1024 x 40 states per environment.
    float64: 1e-8 on every sample.  float32: EVERY sample within C x (the oracle's own sensitivity to float32-sized input
    perturbations) + rounding floor -- no percentage of samples is exempt (tests/parity_tools.py)."""
    B, T = 1024, 40
    spec, rec0, states, acts, c_or = _teacher_forced_reference(name, B, T)
    rec = rec0.fresh()
    env = _env(name, B, dt, lanes_per_env=lanes)
    nq, ng = spec.dim_q, spec.n_g
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
    if dt == 'f64':
        assert np.max(rec.err) < 1e-8, np.max(rec.err)
    else:
        print(rec.finish('%s lanes %d' % (name, lanes), max_vacuous={'circle': 0.0, 'planar': 0.005, 'iiwa': 0.35}[name]))      # measured 31.64 % (iiwa), 0.19 % (planar)
        if name == 'iiwa':
            print(_followed_chart_report(rec, name, lanes))
    # constraint statistics accumulated on the device == oracle's (A13)
    c_dev = env.get_constraints_logs()
    assert np.allclose(c_dev, c_or, atol=1e-8 if dt == 'f64' else 2e-3)


_AWAY_CACHE = {}


def _away_reference(chart, B, T):
    """Oracle side of the second teacher-forced iiwa state set (tests/chart_cases.away_init_q), per chart."""
    if chart not in _AWAY_CACHE:
        from parity_tools import SensitivityRecorder
        from chart_cases import away_init_q
        spec = SPECS['iiwa']()
        spec.chart_mode = {'reference': 0, 'canonical': 1}[chart]
        rng = np.random.default_rng(17)
        o = ob.BatchedAtacomEnv(spec, B, init_q=away_init_q(B, rng))
        rec = SensitivityRecorder(_step_outputs, seed=5)
        states, acts, near = [], [], []
        for t in range(T):
            a = rng.uniform(-1.3, 1.3, (B, spec.n_null))
            a[: B // 8] = np.sign(a[: B // 8])
            acts.append(a)
            rec.prepare(o, (a,))
            states.append(_full_state(None, o))
            near.append((np.abs(o.q) < 0.1).any(1).mean())
            o.step(a)
        _AWAY_CACHE[chart] = (spec, rec, states, acts, float(np.mean(near)))
    return _AWAY_CACHE[chart]


@pytest.mark.mapping(name='iiwa')
@pytest.mark.parametrize('lanes', [1, 2, 4, 8])
@pytest.mark.parametrize('dt', ['f64', 'f32'])
@pytest.mark.parametrize('chart', ['reference', 'canonical'])
def test_iiwa_step_teacher_forced_away_from_the_planar_pose(chart, dt, lanes):
    """the teacher-forced states of the test above are all drawn around the reset pose, which is exactly
    planar (q1 = q3 = q5 = 0) -- the reference's most ill-conditioned regime.  Second set: joint states spread by 0.3 rad,
    corrected onto the constraint, feasible, NO joint within 0.1 rad of zero at the start (768 x 24 free-running oracle
    states from there), both charts, every mapping.  float64 1e-8 on every sample.  float32: the sensitivity rule with a
    vacuous-bound ceiling of 5 % for the canonical chart (measured 2.6 %).  For the reference chart the bound stays vacuous
    on about a quarter of these samples as well -- its tolerance regime is not a property of the planar pose: the nominal
    hitting posture keeps the striker vertical, dz/dq6 ~ 0, so the DEFAULT chart is degenerate there -- and those
    samples are compared with the oracle forced onto the device's own chart decisions (item 3b)."""
    B, T = 768, 24
    spec, rec0, states, acts, near = _away_reference(chart, B, T)
    assert near < 0.2                                           # the set stays away from zero joints while it free-runs
    rec = rec0.fresh()
    env = _env('iiwa', B, dt, lanes_per_env=lanes, chart_mode=chart)
    nq, ng = spec.dim_q, spec.n_g
    for t in range(T):
        env.set_state(states[t])
        obs, r, ab, info = env.step(acts[t])
        s_dev = env.get_state().cpu().numpy()[:, 2 * nq:2 * nq + ng]
        rec.compare(t, np.concatenate([obs.cpu().numpy(), s_dev, r.cpu().numpy()[:, None], ab.cpu().numpy()[:, None] * 1.0], 1))
    if dt == 'f64':
        assert np.max(rec.err) < 1e-8, np.max(rec.err)
        return
    print(rec.finish('iiwa away from the planar pose, %s chart, lanes %d' % (chart, lanes),
                     max_vacuous={'reference': 0.27, 'canonical': 0.05}[chart]))         # measured 22.29 % / 2.56 %
    if chart == 'reference':
        print(_followed_chart_report(rec, 'iiwa', lanes))


@pytest.mark.parametrize('lanes', [1])           # the circle family runs one environment per lane (round 6 census)
def test_circle_hold_flag_is_a_no_op(lanes):
    """CircularMotion has ONE physics sub-step per env step, so hold_q cannot change anything -- but hold_q = 1 selects
    the kernels with the first right reflector hoisted out of the (one-trip) sub-step loop (row 0 of the circle's J_c is
    its slack-free equality row), at the smallest shape the solvers see (2 x 3)."""
    B, T = 256, 30
    rng = np.random.default_rng(5)
    for dt, tol in (('f64', 1e-12), ('f32', 2e-5)):
        e0 = _env('circle', B, dt, lanes_per_env=lanes, hold_q=False)
        e1 = _env('circle', B, dt, lanes_per_env=lanes, hold_q=True)
        for t in range(T):
            a = rng.uniform(-1.2, 1.2, (B, 1))
            e1.set_state(e0.get_state())
            o0, r0, _, _ = e0.step(a)
            o1, r1, _, _ = e1.step(a)
            assert (o0 - o1).abs().max() < tol and (r0 - r1).abs().max() < tol, (dt, t)


@pytest.mark.mapping(dt='f64')
@pytest.mark.parametrize('lanes', [1, 2, 4, 8])
@pytest.mark.parametrize('name', ['planar', 'iiwa'])
def test_refresh_and_exact_bias_variants_against_oracle(name, lanes):
    """The two documented deviations-by-flag from the reference's quirks: hold_q = 0 (q, dq refreshed every sub-step
    instead of the zero-order hold, quirk Q1) and bias_mode = exact (true dJ/dt dq instead of w x v, quirk Q2)."""
    spec = SPECS[name](bias_mode='exact')
    spec.hold_q = False
    B, T = 384, 16
    env = _env(name, B, 'f64', lanes_per_env=lanes, hold_q=False, bias_mode='exact')
    nq, ng = spec.dim_q, spec.n_g
    rng = np.random.default_rng(13)
    init_q = env.get_state().cpu().numpy()[:, :nq] + rng.normal(0, 0.05, (B, nq))
    o = ob.BatchedAtacomEnv(spec, B, init_q=init_q, init_puck=np.array([0.8, 0.4, 0, 0, 0, 0.0]))
    worst = 0.0
    for t in range(T):
        a = rng.uniform(-1.2, 1.2, (B, spec.n_null))
        env.set_state(_full_state(env, o))
        obs, r, ab, _ = env.step(a)
        oo, orr, oab, _ = o.step(a)
        worst = max(worst, np.abs(obs.cpu().numpy() - oo).max(), np.abs(r.cpu().numpy() - orr).max())
    assert worst < 1e-8, worst
    # and the variants really differ from the defaults
    env_d = _env(name, B, 'f64', lanes_per_env=lanes)
    env_d.set_state(_full_state(env_d, o)); env.set_state(_full_state(env, o))
    a = rng.uniform(-1.0, 1.0, (B, spec.n_null))
    assert (env_d.step(a)[0] - env.step(a)[0]).abs().max() > 1e-6


@pytest.mark.parametrize('lanes', [1, 2, 4, 8])
@pytest.mark.parametrize('dt', ['f64', 'f32'])
@pytest.mark.parametrize('name', ['planar', 'iiwa'])
def test_puck_contact_model_against_oracle(name, dt, lanes):
    """Row N1 (this build's contact model, Bullet being unpinned): mallet impulses, rim bounces, goal mouth, has_hit
    latch, hitting reward and termination -- HIP vs oracle from identical injected states."""
    spec = SPECS[name]()
    B, T = 512, 30
    env = _env(name, B, dt, lanes_per_env=lanes)
    nq, ng = spec.dim_q, spec.n_g
    st0 = env.get_state().cpu().numpy().astype(np.float64)
    rng = np.random.default_rng(21)
    init_q = st0[:, :nq] + rng.normal(0, 0.03, (B, nq))
    mal = ob.mallet_xy_world(spec, init_q)
    puck = np.zeros((B, 6))
    third = B // 3
    ang = rng.uniform(-1.2, 1.2, B)
    puck[:, 0] = mal[:, 0] + 0.1 * np.cos(ang)            # just in front of the mallet, moving towards it
    puck[:, 1] = mal[:, 1] + 0.1 * np.sin(ang)
    puck[:, 3] = -rng.uniform(0.0, 1.0, B) * np.cos(ang)
    puck[:, 4] = -rng.uniform(0.0, 1.0, B) * np.sin(ang)
    puck[third:2 * third, 0] = rng.uniform(0.6, 0.93, third)             # fast pucks near the far end: rims + goal
    puck[third:2 * third, 1] = rng.uniform(-0.45, 0.45, third)
    puck[third:2 * third, 3] = rng.uniform(1.0, 4.0, third)
    puck[third:2 * third, 4] = rng.uniform(-3.0, 3.0, third)
    from parity_tools import SensitivityRecorder

    def contact_outputs(p, inputs):
        oo, orr, oab, _ = p.step(inputs[0])
        return np.concatenate([oo, orr[:, None], p.r_hit[:, None], p.vel_hit_x[:, None], oab[:, None] * 1.0,
                               p.has_hit[:, None] * 1.0], 1)

    o = ob.BatchedAtacomEnv(spec, B, init_q=init_q, init_puck=puck)
    rec = SensitivityRecorder(contact_outputs, seed=3)
    n_abs, n_goal, n_hit = 0, 0, 0
    for t in range(T):
        a = rng.uniform(-1.0, 1.0, (B, spec.n_null))
        a[:third, 0] = 1.0
        env.set_state(_full_state(env, o))
        obs, r, ab, info = env.step(a)
        st = env.get_state().cpu().numpy()
        dev = np.concatenate([obs.cpu().numpy(), r.cpu().numpy()[:, None], st[:, 2 * nq + ng + 7:2 * nq + ng + 9],
                              ab.cpu().numpy()[:, None] * 1.0, st[:, 2 * nq + ng + 6:2 * nq + ng + 7]], 1)
        if dt == 'f32':
            rec.record(o, (a,), dev)              # oracle outputs + sensitivity to float32-sized perturbations
        oo, orr, oab, _ = o.step(a)
        if dt == 'f64':
            ref = np.concatenate([oo, orr[:, None], o.r_hit[:, None], o.vel_hit_x[:, None], oab[:, None] * 1.0,
                                  o.has_hit[:, None] * 1.0], 1)
            assert np.abs(dev - ref).max() < 1e-8, np.abs(dev - ref).max()
        n_abs += oab.sum(); n_goal += (orr > 70).sum(); n_hit += o.has_hit.sum()
        o.reset(oab)                                     # finished episodes restart (both sides via set_state)
    assert n_abs > 0 and n_goal > 0 and n_hit > 0          # the scenario really exercises contacts, goals, hits
    if dt == 'f32':
        print(rec.finish('contact %s lanes %d' % (name, lanes)))   # every sample explained, flags included


@pytest.mark.parametrize('dt', ['f64', 'f32'])
@pytest.mark.parametrize('tag', ['E', 'T'])
def test_baseline_wrappers_reference_trajectories_through_capi(golden, tag, dt):
    """Row N3, golden set G9: the reference's CircleEnvErrorCorrection / CircleEnvTerminated, replayed through HIP."""
    g = golden('circle_baselines')
    acts, obs, rew, absb, s, init = (g[tag + '_' + k] for k in ('actions', 'obs', 'reward', 'absorbing', 's', 'init'))
    n, T = acts.shape[:2]
    env = _env('circle_ec' if tag == 'E' else 'circle_t', n, dt, horizon=300)
    assert env.dims['null'] == 2 and env.info.action_space.shape == (2,)
    spec = osc.circle_ec_spec()
    full = np.zeros((n, env.state_dim))
    worst = 0.0
    for t in range(T):
        prev = init if t == 0 else obs[:, t - 1]
        s_prev = (np.array([osc.slack_init(spec, p[:2], p[2:]) for p in init]) if t == 0 else s[:, t - 1]) if tag == 'E' else 0.0
        full[:, :4], full[:, 4:5], full[:, -1] = prev, s_prev, t
        env.set_state(full)
        o, r, ab, _ = env.step(acts[:, t])
        worst = max(worst, np.abs(o.cpu().numpy() - obs[:, t]).max(), np.abs(r.cpu().numpy() - rew[:, t]).max())
        assert (ab.cpu().numpy() == absb[:, t]).all()
    assert worst < (1e-9 if dt == 'f64' else 5e-4), worst
    from rl_on_manifold_amd import CircleEnvErrorCorrection, CircleEnvTerminated
    m = (CircleEnvErrorCorrection if tag == 'E' else CircleEnvTerminated)(horizon=300)
    st = m.reset()
    o, r, ab, info = m.step(acts[0, 0])
    assert np.allclose(st, [-1, 0, 0, 0]) and np.allclose(o, obs[0, 0], atol=1e-5) and isinstance(ab, bool)


def _policy_pair(golden, key, std=0.5, activation='relu'):
    from rl_on_manifold_amd import MlpPolicy
    from oracle.policy import MlpPolicy as OraclePolicy
    g = golden('policy_net')
    W = [g[key + '._h%d.%s' % (i, w)] for i in (1, 2, 3) for w in ('weight', 'bias')]
    n_in, n_out = W[0].shape[1], W[4].shape[0]
    rng = np.random.default_rng(3)
    shift, scale = rng.uniform(-0.5, 0.5, n_in), rng.uniform(0.5, 2.0, n_in)
    stdv = np.full(n_out, std)
    dev = MlpPolicy(*[torch.tensor(w) for w in W], std=torch.tensor(stdv), obs_shift=torch.tensor(shift),
                    obs_scale=torch.tensor(scale), activation=activation)
    ora = OraclePolicy(*W, obs_shift=shift, obs_scale=scale, std=stdv, activation=activation)
    return dev, ora


@pytest.mark.mapping(kind='mlp')
@pytest.mark.parametrize('lanes', [1, 2, 4, 8])
@pytest.mark.parametrize('dt', ['f64', 'f32'])
@pytest.mark.parametrize('name,key', [('iiwa', 'ppo_iiwa'), ('planar', 'sac_planar')])
def test_policy_rollout_against_oracle(golden, name, key, dt, lanes):
    """Row N2: actor MLP (the reference's PPONetwork / SACActorNetwork weights) + Gaussian noise + env step fused in
    one kernel, vs the oracle, one step at a time from identical injected states."""
    from oracle.policy import rollout as oracle_rollout
    spec = SPECS[name]()
    B, T = 256, 12
    env = _env(name, B, dt, lanes_per_env=lanes)
    dev, ora = _policy_pair(golden, key, activation='relu' if name == 'iiwa' else 'tanh')
    nq, ng = spec.dim_q, spec.n_g
    rng = np.random.default_rng(9)
    init_q = env.get_state().cpu().numpy().astype(np.float64)[:, :nq] + rng.normal(0, 0.05, (B, nq))
    from parity_tools import SensitivityRecorder

    def policy_outputs(p, inputs):
        ref = oracle_rollout(p, ora, 1, noise=inputs[0].reshape(1, p.B, -1), auto_reset=False)
        return np.concatenate([ref['action'][0], ref['next_obs'][0], ref['reward'][0][:, None]], 1)

    o = ob.BatchedAtacomEnv(spec, B, init_q=init_q)
    rec = SensitivityRecorder(policy_outputs, seed=4)
    for t in range(T):
        eps = rng.standard_normal((1, B, spec.n_null))
        env.set_state(_full_state(env, o))
        out = env.rollout_policy(dev, 1, noise=torch.tensor(eps))
        d = np.concatenate([out['action'][0].cpu().numpy(), out['next_obs'][0].cpu().numpy(),
                            out['reward'][0].cpu().numpy()[:, None]], 1)
        if dt == 'f32':
            rec.record(o, (eps[0],), d)
        ref = oracle_rollout(o, ora, 1, noise=eps, auto_reset=False)
        assert np.abs(out['obs'][0].cpu().numpy() - ref['obs'][0]).max() < 1e-5
        if dt == 'f64':
            r = np.concatenate([ref['action'][0], ref['next_obs'][0], ref['reward'][0][:, None]], 1)
            assert np.abs(d - r).max() < 1e-8, np.abs(d - r).max()
    if dt == 'f32':
        print(rec.finish('policy %s lanes %d' % (name, lanes)))


@pytest.mark.mapping(name='planar', kind='mlp')
@pytest.mark.parametrize('lanes', [1, 2, 4, 8])
@pytest.mark.parametrize('dt', ['f64', 'f32'])
def test_sac_style_policy_rollout_against_oracle(golden, dt, lanes):
    """The reference's default agent is SAC (examples/iiwa_air_hockey_exp.py:345): mean and log-sigma networks
    (SACActorNetwork) and a tanh-squashed sample.  Fused in the rollout kernel vs the oracle."""
    from rl_on_manifold_amd import MlpPolicy
    from oracle.policy import MlpPolicy as OraclePolicy, rollout as oracle_rollout
    g = golden('policy_net')
    W = [g['sac_planar._h%d.%s' % (i, w)] for i in (1, 2, 3) for w in ('weight', 'bias')]
    rng = np.random.default_rng(31)
    Ws = [w * 0.5 + rng.normal(0, 0.05, w.shape) for w in W]           # a different network for log sigma
    Ws[5] = Ws[5] - 1.0
    shift, scale = rng.uniform(-0.5, 0.5, 12), rng.uniform(0.5, 2.0, 12)
    dev = MlpPolicy(*[torch.tensor(w) for w in W], obs_shift=torch.tensor(shift), obs_scale=torch.tensor(scale),
                    sigma_weights=[torch.tensor(w) for w in Ws], squash=True, log_std_min=-3.0, log_std_max=0.5)
    ora = OraclePolicy(*W, obs_shift=shift, obs_scale=scale, sigma_weights=Ws, squash=True, log_std_min=-3.0,
                       log_std_max=0.5)
    spec = SPECS['planar']()
    B = 256
    env = _env('planar', B, dt, lanes_per_env=lanes)
    init_q = env.get_state().cpu().numpy().astype(np.float64)[:, :3] + rng.normal(0, 0.05, (B, 3))
    from parity_tools import SensitivityRecorder

    def sac_outputs(p, inputs):
        ref = oracle_rollout(p, ora, 1, noise=inputs[0].reshape(1, p.B, -1), auto_reset=False)
        return np.concatenate([ref['action'][0], ref['next_obs'][0]], 1)

    o = ob.BatchedAtacomEnv(spec, B, init_q=init_q)
    rec = SensitivityRecorder(sac_outputs, seed=6)
    for t in range(10):
        eps = rng.standard_normal((1, B, 3))
        env.set_state(_full_state(env, o))
        out = env.rollout_policy(dev, 1, noise=torch.tensor(eps))
        a = out['action'][0].cpu().numpy()
        assert np.abs(a).max() <= 1.0                                   # squashed
        d = np.concatenate([a, out['next_obs'][0].cpu().numpy()], 1)
        if dt == 'f32':
            rec.record(o, (eps[0],), d)
        ref = oracle_rollout(o, ora, 1, noise=eps, auto_reset=False)
        if dt == 'f64':
            assert np.abs(d - np.concatenate([ref['action'][0], ref['next_obs'][0]], 1)).max() < 1e-8
    if dt == 'f32':
        print(rec.finish('sac lanes %d' % lanes))
    # clamp really active somewhere, sigma really state dependent
    sg = ora.sigma(o.observation())
    assert sg.std() > 1e-3 and sg.min() >= np.exp(-3.0) - 1e-12 and sg.max() <= np.exp(0.5) + 1e-12


def test_policy_rollout_multi_step_consistency(golden):
    """T-step policy rollout == feeding the actions it drew to the plain rollout kernel; deterministic without noise."""
    B, T = 192, 10
    dev, _ = _policy_pair(golden, 'ppo_iiwa')
    # both T-step kernels run the same mapping under the automatic choice (8 lanes at this batch), which the two need
    # to agree to rounding
    e1 = _env('iiwa', B, 'f32', auto_reset=True, horizon=4)
    e2 = _env('iiwa', B, 'f32', auto_reset=True, horizon=4)
    assert e1.rollout_lanes_per_env == 8
    gen = torch.Generator(device=DEV); gen.manual_seed(1)
    eps = torch.randn((T, B, 5), device=DEV, generator=gen)
    o1 = e1.rollout_policy(dev, T, noise=eps)
    o2 = e2.rollout(o1['action'])
    for k in ('obs', 'next_obs', 'reward'):
        assert torch.allclose(o1[k], o2[k], rtol=1e-5, atol=1e-6), k
    assert torch.equal(o1['last'], o2['last']) and o1['last'][3].all() and not o1['last'][2].any()
    assert torch.allclose(o1['obs'][4], o1['obs'][0])          # auto-reset at the horizon inside the kernel
    e3 = _env('iiwa', B, 'f32')
    a = e3.rollout_policy(dev, 2)['action']
    e3.reset()
    assert torch.equal(a, e3.rollout_policy(dev, 2)['action'])


@pytest.mark.parametrize('dt', ['f64', 'f32'])
def test_circle_reference_trajectories_through_capi(golden, dt):
    """G4: the reference's own CircleEnvAtacom trajectories, replayed step by step through the HIP path."""
    g = golden('circle_traj')
    n = len(g['init'])
    env = _env('circle', n, dt)
    full = np.zeros((n, env.state_dim))
    worst = 0.0
    for t in range(500):
        prev = g['init'] if t == 0 else g['obs'][:, t - 1]
        s_prev = g['s0'] if t == 0 else g['s'][:, t - 1]
        full[:, :4], full[:, 4:5], full[:, -1] = prev, s_prev, t
        env.set_state(full)
        obs, r, ab, _ = env.step(g['actions'][:, t])
        s_dev = env.get_state().cpu().numpy()[:, 4:5]
        worst = max(worst, np.abs(obs.cpu().numpy() - g['obs'][:, t]).max(), np.abs(s_dev - g['s'][:, t]).max(),
                    np.abs(r.cpu().numpy() - g['reward'][:, t]).max())
        assert not ab.any()
    assert worst < (1e-9 if dt == 'f64' else 2e-4), worst
    # free-running from the fixed reset, the reference's constraint log of trajectory 0 (circle_base.py:109-115)
    env1 = _env('circle', 1, dt)
    for t in range(500):
        env1.step(g['actions'][0:1, t])
    logs = env1.get_constraints_logs()
    assert np.allclose(logs, g['logs'][0], atol=1e-8 if dt == 'f64' else 5e-3), (logs, g['logs'][0])


@pytest.mark.parametrize('lanes', [1, 2, 4, 8])
@pytest.mark.parametrize('dt', ['f64', 'f32'])
@pytest.mark.parametrize('name', ['planar', 'iiwa'])
def test_generic_wrapper_reference_trajectories_through_capi(golden, name, dt, lanes):
    """G5: the reference's generic AtacomEnvWrapper (its SVD, its rref, its zero-order hold over 4 sub-steps)
    at the planar / iiwa shapes, replayed through the HIP path."""
    g = golden('generic_traj')
    spec = SPECS[name]()
    init, acts, obs, s, s0 = (g[name + '_' + k] for k in ('init', 'actions', 'obs', 's', 's0'))
    n, T = acts.shape[:2]
    nq, ng = spec.dim_q, spec.n_g
    env = _env(name, n, dt, lanes_per_env=lanes)
    full = env.get_state().cpu().numpy().astype(np.float64)
    # the golden runs had a static puck: park this build's puck far from the arm, compare the arm columns
    full[:, 2 * nq + ng:2 * nq + ng + 6] = [0.8, 0.4, 0, 0, 0, 0]
    from parity_tools import SensitivityRecorder

    def arm_outputs(p, inputs):
        oo, _, _, _ = p.step(inputs[0])
        return np.concatenate([oo[:, 6:], p.s], 1)

    # float32: the golden outputs ARE the reference's; the oracle (== golden to 1e-9, tests/test_oracle_trajectories.py)
    # only supplies the sensitivity of each golden state to float32-sized perturbations
    orc = ob.BatchedAtacomEnv(spec, n, init_q=init[:, :nq])
    orc.puck[:] = [0.8, 0.4, 0, 0, 0, 0]
    rec = SensitivityRecorder(arm_outputs, seed=8)
    errs = []
    for t in range(T):
        q, dq, ss = (init[:, :nq], init[:, nq:], s0) if t == 0 else (obs[:, t - 1, 6:6 + nq], obs[:, t - 1, 6 + nq:], s[:, t - 1])
        full[:, :nq], full[:, nq:2 * nq], full[:, 2 * nq:2 * nq + ng], full[:, -1] = q, dq, ss, t
        env.set_state(full)
        o, r, ab, _ = env.step(acts[:, t])
        s_dev = env.get_state().cpu().numpy()[:, 2 * nq:2 * nq + ng]
        d = np.concatenate([o.cpu().numpy()[:, 6:], s_dev], 1)
        golden_out = np.concatenate([obs[:, t, 6:], s[:, t]], 1)
        errs.append(np.abs(d - golden_out).max(1))
        if dt == 'f32':
            orc.q[:], orc.dq[:], orc.s[:], orc.t[:] = q, dq, ss, t
            base = rec.prepare(orc, (acts[:, t],))
            assert np.abs(base - golden_out).max() < 1e-8          # the oracle stands on the golden values
            rec.compare(t, d)
    errs = np.array(errs)
    if dt == 'f64':
        assert errs.max() < 1e-8, errs.max()
    else:
        print(rec.finish('golden G5 %s lanes %d' % (name, lanes)))


@pytest.mark.parametrize('lanes', [1, 2, 4, 8])
@pytest.mark.parametrize('name', ['circle', 'planar', 'iiwa'])
def test_rollout_kernel_equals_step_kernel(name, lanes):
    """atacom_rollout (T steps, state in registers) == T x atacom_step (to a few ulp), incl. auto-reset."""
    B, T = 200, 2 * 7
    horizon = 5
    rng = np.random.default_rng(5)
    acts = torch.tensor(rng.uniform(-1.2, 1.2, (T, B, SHAPES[name][2])), device=DEV, dtype=torch.float32)
    e1 = _env(name, B, 'f32', auto_reset=True, horizon=horizon, lanes_per_env=lanes)
    e2 = _env(name, B, 'f32', auto_reset=True, horizon=horizon, lanes_per_env=lanes)
    out = e1.rollout(acts)
    close = lambda a, b: torch.allclose(a, b, rtol=1e-5, atol=1e-6)      # noqa: E731
    for t in range(T):
        o, r, ab, info = e2.step(acts[t])
        # two separately compiled kernels (different FMA contraction / scheduling): equal to a few ulp
        assert close(out['next_obs'][t], o) and close(out['reward'][t], r)
        assert torch.equal(out['absorbing'][t].bool(), ab) and torch.equal(out['last'][t].bool(), info['last'])
        assert bool(info['last'].all()) == ((t + 1) % horizon == 0)       # horizon reached -> last
    assert close(e1.get_state(), e2.get_state())
    assert np.allclose(e1.get_constraints_logs(), e2.get_constraints_logs(), rtol=1e-4, atol=1e-6)
    # obs[t+1] after a `last` step is the reset observation
    reset_obs = _env(name, B, 'f32').reset()
    assert torch.equal(out['obs'][horizon], reset_obs)


@pytest.mark.parametrize('dt', ['f64', 'f32'])
@pytest.mark.parametrize('name', ['circle', 'planar', 'iiwa'])
def test_non_finite_actions_are_clipped_not_propagated(name, dt):
    """A batch must survive a policy that emits NaN / Inf for some environments (include/atacom_hip.h: atacom_step): +-Inf clip
    to +-1, a NaN component acts as -1, the state stays finite, and the other environments are bit-for-bit what they are
    without the bad rows."""
    B = 256
    k = SHAPES[name][2]
    g = torch.Generator(device=DEV).manual_seed(2)
    a = (torch.rand((B, k), device=DEV, generator=g) * 2 - 1).to(DT[dt])
    bad = a.clone()
    bad[3] = float('nan'); bad[10, 0] = float('inf'); bad[17, k - 1] = float('-inf'); bad[40, 0] = float('nan')
    repl = bad.clone()
    repl[3] = -1.0; repl[10, 0] = 1.0; repl[17, k - 1] = -1.0; repl[40, 0] = -1.0
    e1, e2 = _env(name, B, dt), _env(name, B, dt)
    for t in range(3):
        o1, r1, ab1, _ = e1.step(bad)
        o2, r2, ab2, _ = e2.step(repl)
        assert torch.isfinite(o1).all() and torch.isfinite(r1).all(), t
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(ab1, ab2), t
    assert torch.isfinite(e1.get_state()).all() and torch.equal(e1.get_state(), e2.get_state())
    c = e1.get_constraints_logs()
    assert all(np.isfinite(c))


@pytest.mark.parametrize('lanes', [1, 2, 4, 8])
@pytest.mark.parametrize('B', [1, 63, 65])
def test_ragged_batches_and_masked_reset(B, lanes):
    env = _env('iiwa', B, 'f32', lanes_per_env=lanes)
    a = torch.zeros((B, 5), device=DEV)
    o0 = env.reset()
    for _ in range(3):
        env.step(a + 0.5)
    st = env.get_state()
    mask = torch.zeros(B, dtype=torch.uint8, device=DEV)
    mask[::2] = 1
    o1 = env.reset(mask=mask)
    st2 = env.get_state()
    assert torch.equal(o1[::2], o0[::2])                     # masked envs are back at the reset observation
    assert torch.equal(st2[1::2], st[1::2])                  # the others are untouched
    assert (st2[::2, -1] == 0).all() and (st[:, -1] == 3).all()
    env.set_state(st)
    assert torch.equal(env.get_state(), st)                  # state round trip


def _config4_init(B, gen):
    """q = q_init + N(0, 0.05^2), one correction step onto the equality constraint,
    rejected unless all g < 0 and |f| < 1e-3 (bench.py uses the same construction)."""
    import bench
    return bench.feasible_init('iiwa', B, torch.device(DEV), gen)[0]


def test_full_size_closed_loop_properties():
    """BASELINE config 4 at full size (8192 iiwa envs, one 120-step episode, float32): size-independent
    properties of ATACOM from FEASIBLE initial states -- the constraint residual the engine itself produces under
    random actions stays at the centimetre level, velocities stay inside their limits."""
    B, T = 8192, 120
    env = _env('iiwa', B, 'f32')
    gen = torch.Generator(device=DEV); gen.manual_seed(0)
    init = _config4_init(B, gen)
    env.reset(state=init)
    acts = torch.rand((T, B, 5), device=DEV, generator=gen) * 2 - 1
    out = env.rollout(acts)
    assert torch.isfinite(out['obs']).all() and torch.isfinite(out['reward']).all()
    c_avg, c_max, c_dq_max = env.get_constraints_logs()
    assert c_max < 0.05 and c_avg < 0.003, (c_avg, c_max)    # max |c(q)| over 8192 x 120 = 1e6 env-steps (m / rad^2)
    assert c_dq_max <= 1e-4                                   # |dq| never exceeds the velocity limit
    assert (out['last'][-1] == 1).all() and (out['last'][:-1].sum(0) == out['absorbing'][:-1].sum(0)).all()


def test_free_running_constraint_statistics_against_oracle():
    """The correctness metric of SURVEY.md section 8d -- max |c(q)| / c_avg / c_dq_max exactly as atacom.py:201-216 --
    FREE-RUNNING (no teacher forcing): 1024 config-4 environments x 120 steps on the float32 HIP engine and on the
    float64 oracle from identical initial states and actions.  The closed loop is chaotic, so
    trajectories part ways; the statistics must agree: c_max within 1.5x, c_avg within 15 %."""
    B, T = 1024, 120
    gen = torch.Generator(device=DEV); gen.manual_seed(1)
    init = _config4_init(B, gen)
    acts = torch.rand((T, B, 5), device=DEV, generator=gen) * 2 - 1
    env = _env('iiwa', B, 'f32', auto_reset=True)
    env.reset(state=init)
    env.rollout(acts)
    d_avg, d_max, d_dq = env.get_constraints_logs()
    i64 = init.double().cpu().numpy()
    o = ob.BatchedAtacomEnv(SPECS['iiwa'](), B, init_q=i64[:, :6], init_dq=i64[:, 6:12], init_puck=i64[:, 12:18])
    a64 = acts.double().cpu().numpy()
    for t in range(T):
        _, _, ab, _ = o.step(a64[t])
        last = ab | (o.t >= o.spec.horizon)
        if last.any():
            o.reset(last)
    o_avg, o_max, o_dq = o.get_constraints_logs()
    assert o_max / 1.5 <= d_max <= 1.5 * o_max, (d_max, o_max)
    assert abs(d_avg - o_avg) <= 0.15 * o_avg, (d_avg, o_avg)
    assert d_dq <= 1e-4 and o_dq <= 1e-9


@pytest.mark.parametrize('name,B,T,sub', [('planar', 8192, 120, 1024), ('circle', 4096, 500, 1024)])
def test_free_running_statistics_configs_2_and_3_at_full_size(name, B, T, sub):
    """BASELINE configs 2 (CircularMotion A, 4096 envs x 500 steps) and 3 (PlanarAirHockey H, 8192 x 120) at FULL SIZE,
    free-running, float32, initial states as bench.py builds them -- the correctness metric of atacom.py:201-216
    / circle_base.py:86-115 against the float64 oracle on identical initial states and actions (config 4
    has the test above).  The oracle runs the first `sub` environments; the device statistics it is compared with come from
    a second handle holding exactly those environments (bit-identical to their rows of the full-size run, asserted): c_max
    within 1.5x, c_avg within 15 %; the full-size run's own c_avg must agree too, its c_max can only be larger."""
    import bench
    dev = torch.device(DEV)
    gen = torch.Generator(device=DEV); gen.manual_seed(2)
    env, init, _ = bench.make_env(name, B, dev, gen)
    lanes = env.rollout_lanes_per_env
    k = env.dims['null']
    acts = torch.rand((T, B, k), device=DEV, generator=gen) * 2 - 1
    env.get_constraints_logs()
    out = env.rollout(acts)
    f_avg, f_max, f_dq = env.get_constraints_logs()
    assert torch.isfinite(out['next_obs']).all() and torch.isfinite(out['reward']).all()
    assert (out['last'].sum(0) >= 1).all()                     # horizon T: every environment ends an episode on the way
    # the same environments alone in a handle: identical rows, and the statistics of exactly the oracle's sample
    small = _env(name, sub, 'f32', auto_reset=True, lanes_per_env=lanes)
    small.reset(state=init[:sub])
    small.get_constraints_logs()
    o_s = small.rollout(acts[:, :sub].contiguous())
    for key in ('next_obs', 'reward', 'absorbing', 'last'):
        assert torch.equal(o_s[key], out[key][:, :sub]), key
    d_avg, d_max, d_dq = small.get_constraints_logs()
    i64, a64 = init[:sub].double().cpu().numpy(), acts[:, :sub].double().cpu().numpy()
    nq = env.dims['q']
    kw = dict(init_q=i64[:, :nq], init_dq=i64[:, nq:2 * nq])
    if name != 'circle':
        kw['init_puck'] = i64[:, 2 * nq:2 * nq + 6]
    o = ob.BatchedAtacomEnv(SPECS[name](), sub, **kw)
    for t in range(T):
        _, _, ab, _ = o.step(a64[t])
        last = ab | (o.t >= o.spec.horizon)
        if last.any():
            o.reset(last)
    o_avg, o_max, o_dq = o.get_constraints_logs()
    print('%s: device (full %d envs) c_avg %.3e c_max %.3e c_dq %.2e | device (first %d) %.3e %.3e %.2e | oracle f64 %.3e %.3e %.2e'
          % (name, B, f_avg, f_max, f_dq, sub, d_avg, d_max, d_dq, o_avg, o_max, o_dq))
    if o_max > 0:
        assert o_max / 1.5 <= d_max <= 1.5 * o_max, (d_max, o_max)
    else:                                    # planar: the constraints are never violated, c_max is the (negative) closest approach
        assert abs(d_max - o_max) <= 0.5 * abs(o_max) + 1e-3 and d_max <= 1e-3, (d_max, o_max)
    assert abs(d_avg - o_avg) <= 0.15 * abs(o_avg), (d_avg, o_avg)
    assert abs(f_avg - o_avg) <= 0.15 * abs(o_avg) and f_max >= d_max, (f_avg, f_max, o_avg)
    assert abs(d_dq - o_dq) <= 1e-4 + 0.15 * abs(o_dq) and f_dq >= d_dq - 1e-6, (d_dq, o_dq, f_dq)


@pytest.mark.parametrize('dt', ['f64', 'f32'])
def test_circle_time_step_quirk_through_capi(golden, dt):
    """Quirk Q4 (G4b): `time_step` of CircleEnvAtacom / CircleEnvErrorCorrection reaches only the wrapper -- the reference's
    own trajectories at time_step 0.02 / 0.004 through the HIP path (the facades pass time_step exactly as the reference
    constructors do)."""
    g = golden('circle_time_step')
    for tag, name in (('A', 'circle'), ('A2', 'circle'), ('E', 'circle_ec')):
        ts = float(g[tag + '_time_step'])
        acts, obs, ss = g[tag + '_actions'], g[tag + '_obs'], g[tag + '_s']
        n, T, _ = acts.shape
        env = _env(name, n, dt, time_step=ts, horizon=T)
        full = env.get_state().cpu().numpy().astype(np.float64)
        assert np.allclose(full[:, 4:5], g[tag + '_s0'], atol=1e-6)
        worst = 0.0
        for t in range(T):
            if t > 0:
                full[:, :4], full[:, 4:5], full[:, -1] = obs[:, t - 1], ss[:, t - 1], t
                env.set_state(full)
            o, r, ab, _ = env.step(acts[:, t])
            s_dev = env.get_state().cpu().numpy()[:, 4:5]
            worst = max(worst, np.abs(o.cpu().numpy() - obs[:, t]).max(), np.abs(s_dev - ss[:, t]).max(),
                        np.abs(r.cpu().numpy() - g[tag + '_reward'][:, t]).max())
        assert worst < (1e-9 if dt == 'f64' else 5e-4), (tag, worst)


def test_reference_facade_surface(golden):
    """Drop-in surface: constructor names / arguments, info, reset / step return types (atacom.py:93-115)."""
    from rl_on_manifold_amd import CircleEnvAtacom, AirHockeyIiwaAtacom, AirHockeyPlanarAtacom
    mdp = CircleEnvAtacom(horizon=500, gamma=0.99, Kc=100, time_step=0.01, dtype=torch.float64)
    assert mdp.info.action_space.shape == (1,) and mdp.info.observation_space.shape == (4,)
    assert mdp.info.horizon == 500 and mdp.info.gamma == 0.99
    s = mdp.reset()
    assert isinstance(s, np.ndarray) and np.allclose(s, [-1, 0, 0, 0])
    obs, r, ab, info = mdp.step(np.array([0.27392337]))
    # probed from the reference in SURVEY.md section 8c
    assert np.allclose(obs, [-1.0, 1.36961687e-4, 0.0, 2.73923375e-2], atol=1e-9)
    assert abs(r - 0.13533528260194083) < 1e-9 and ab is False and info == {}
    g = golden('circle_reset_guard')
    for st, ok in zip(g['states'], g['accepted']):
        if ok:
            mdp.reset(st)
        else:
            with pytest.raises(ValueError):
                mdp.reset(st)
    c = mdp.get_constraints_logs()
    assert len(c) == 3
    for cls, k, D in ((AirHockeyPlanarAtacom, 3, 12), (AirHockeyIiwaAtacom, 5, 18)):
        m = cls(task='H', horizon=120, gamma=0.99, random_init=True, timestep=1 / 240., n_intermediate_steps=4)
        assert m.info.action_space.shape == (k,) and m.info.observation_space.shape == (D,)
        o = m.reset()
        assert o.shape == (D,) and -0.6 - 1e-6 <= o[0] + m._engine.cfg.base_xy[0] <= -0.2 + 1e-6
        o2, r, ab, info = m.step(np.random.uniform(-1, 1, k) * 3)        # out-of-range actions are clipped
        assert o2.shape == (D,) and isinstance(r, float) and isinstance(ab, bool)
        assert len(m.get_constraints_logs()) == 3


def test_host_side_error_paths_and_multiple_handles():
    """Drop-in boundary behaviour: shape / dtype validation raises before anything reaches the device, handles are
    independent, work follows the caller's (non-default) stream, destroy is idempotent."""
    from rl_on_manifold_amd import BatchedAtacomEnv, AtacomError
    e1 = _env('planar', 96, 'f32')
    e2 = _env('planar', 96, 'f32')
    with pytest.raises(ValueError):
        e1.step(torch.zeros((96, 2), device=DEV))                 # wrong action dim
    with pytest.raises(ValueError):
        e1.reset(state=torch.zeros((95, e1.init_state_dim), device=DEV))
    with pytest.raises(ValueError):
        e1.set_state(torch.zeros((96, 3), device=DEV))
    with pytest.raises(KeyError):
        BatchedAtacomEnv('no_such_env', 4)
    with pytest.raises(AtacomError):
        BatchedAtacomEnv('planar', 0)                             # atacom_create rejects it with a message
    with pytest.raises(AtacomError):
        BatchedAtacomEnv('planar', 4, lanes_per_env=3)
    # the mapping the library picks for lanes_per_env = 0 (atacom_capi.cpp: pick_lanes), and an explicit request is kept
    for name, B, lanes in (('iiwa', 4096, 8), ('iiwa', 16384, 4), ('iiwa', 32768, 2), ('iiwa', 65536, 1),
                           ('planar', 8192, 4), ('planar', 65536, 1), ('circle', 4096, 1)):
        assert BatchedAtacomEnv(name, B, device=DEV).lanes_per_env == lanes, (name, B)
    assert BatchedAtacomEnv('iiwa', 8192, device=DEV, lanes_per_env=8).lanes_per_env == 8
    # iiwa single steps at 4096 < batch <= 8192: the STATIC policy is 8 lanes -- atacom_create launches nothing hidden and the
    # mapping (hence the bits) is the same in every process, on every box, on every rank (the cross-
    # process check is test_default_mapping_is_deterministic_across_processes in test_gpu_rollout.py)
    assert os.environ.get('ATACOM_CALIBRATE') is None
    e = BatchedAtacomEnv('iiwa', 8192, device=DEV)
    assert (e.lanes_per_env, e.rollout_lanes_per_env) == (8, 8)
    a8 = torch.full((8192, e.dims['null']), 0.25, device=DEV)
    o_first = e.step(a8)[0]
    named = BatchedAtacomEnv('iiwa', 8192, device=DEV, lanes_per_env=8)
    assert torch.equal(named.step(a8)[0], o_first)
    assert named.get_constraints_logs() == e.get_constraints_logs()
    # the timing of 8 lanes against the quad is opt-in (ATACOM_CALIBRATE=1 / verbose): whatever it picks, the timed launches
    # leave no trace in the state, and a second handle of the process takes the same answer
    os.environ['ATACOM_CALIBRATE'] = '1'
    try:
        c1 = BatchedAtacomEnv('iiwa', 8192, device=DEV)
        c2 = BatchedAtacomEnv('iiwa', 8192, device=DEV)
        assert c1.lanes_per_env in (4, 8) and c2.lanes_per_env == c1.lanes_per_env and c1.rollout_lanes_per_env == 8
        twin = BatchedAtacomEnv('iiwa', 8192, device=DEV, lanes_per_env=c1.lanes_per_env)
        assert torch.equal(c1.step(a8)[0], twin.step(a8)[0])
        assert c1.get_constraints_logs() == twin.get_constraints_logs()
    finally:
        del os.environ['ATACOM_CALIBRATE']
    e = BatchedAtacomEnv('iiwa', 16384, device=DEV)
    assert (e.lanes_per_env, e.rollout_lanes_per_env) == (4, 4)
    # float64 (round 6, solver inlined: profiles/r06_f64_lanes_inlined.log): the same rule on the float64 census -- iiwa 8 lanes up
    # to 8192 environments, 4 beyond (r06_f64_lanes_beyond_16384.log); planar 4 up to 16384
    assert BatchedAtacomEnv('iiwa', 64, device=DEV, dtype=torch.float64).lanes_per_env == 8
    assert BatchedAtacomEnv('iiwa', 16384, device=DEV, dtype=torch.float64).lanes_per_env == 4
    assert BatchedAtacomEnv('iiwa', 40000, device=DEV, dtype=torch.float64).lanes_per_env == 4
    assert BatchedAtacomEnv('planar', 64, device=DEV, dtype=torch.float64).lanes_per_env == 4
    p8 = BatchedAtacomEnv('planar', 8192, device=DEV)
    assert (p8.lanes_per_env, p8.rollout_lanes_per_env) == (4, 8)          # planar: T-step kernels on 8 lanes (round 5)
    a = torch.full((96, 3), 0.3, device=DEV)
    side = torch.cuda.Stream(device=DEV)
    with torch.cuda.stream(side):                                 # launches go to the caller's current stream
        o1 = e1.step(a)[0]
    side.synchronize()
    o2 = e2.step(a)[0]
    assert torch.equal(o1, o2)                                    # same inputs, independent handles, same result
    e1.step(a)
    assert not torch.equal(e1.get_state(), e2.get_state())        # stepping one does not touch the other
    e1.close(); e1.close()                                        # idempotent
    numpy_obs = e2.step(np.full((96, 3), 0.3))[0]                 # numpy input is accepted (host copy)
    assert numpy_obs.shape == (96, 12)


def test_kc_vector_and_time_step_overrides():
    """Constructor overrides reach the kernels: per-row Kc (atacom.py:42-45), time_step, n_intermediate_steps."""
    spec = osc.planar_spec(Kc=100.0, dt=1 / 120.0, substeps=2)
    spec.Kc = np.linspace(50, 300, 6)
    B = 128
    env = _env('planar', B, 'f64', Kc=spec.Kc, time_step=1 / 120.0, n_intermediate_steps=2)
    rng = np.random.default_rng(4)
    init_q = env.get_state().cpu().numpy()[:, :3] + rng.normal(0, 0.05, (B, 3))
    o = ob.BatchedAtacomEnv(spec, B, init_q=init_q, init_puck=np.array([0.8, 0.4, 0, 0, 0, 0.0]))
    for t in range(6):
        a = rng.uniform(-1, 1, (B, 3))
        env.set_state(_full_state(env, o))
        obs = env.step(a)[0].cpu().numpy()
        assert np.abs(obs - o.step(a)[0]).max() < 1e-9


@pytest.mark.parametrize('name', ['circle', 'planar', 'iiwa'])
def test_device_random_init_matches_oracle_generator(name):
    """A16, random branch of the reference's reset (circle_base.py:36-42, env_hitting.py:24-25) drawn ON the device with
    the counter-based generator; the oracle restates the generator, so states agree draw for draw, reset after reset."""
    spec = SPECS[name]()
    B, horizon = 300, 3
    spec.horizon = horizon
    env = _env(name, B, 'f64', random_init=True, seed=77, auto_reset=True, horizon=horizon)
    o = ob.BatchedAtacomEnv(spec, B, init_q=env.get_state().cpu().numpy()[0, :spec.dim_q] if name != 'circle' else None,
                            random_init=True, seed=77)
    o.reset(); o.episode[:] = 1; o.reset()                     # engine: create() resets once, the ctor's reset() again
    obs0 = env.reset().cpu().numpy()
    o.reset()
    assert np.allclose(obs0, o.observation(), atol=1e-12)
    if name == 'circle':
        assert np.allclose((obs0[:, :2] ** 2).sum(1), 1.0) and np.abs((obs0[:, :2] * obs0[:, 2:]).sum(1)).max() < 1e-12
        assert len(np.unique(np.round(obs0[:, 1], 6))) > 250
    else:
        px, py = obs0[:, 0] + spec.base_xy[0], obs0[:, 1] + spec.base_xy[1]
        assert (px >= -0.6).all() and (px <= -0.2).all() and (py >= -0.4).all() and (py <= 0.4).all()
        assert px.std() > 0.08 and py.std() > 0.15
    # auto-reset inside the rollout kernel re-draws with the next episode index
    rng = np.random.default_rng(2)
    acts = rng.uniform(-1, 1, (2 * horizon + 1, B, spec.action_dim))
    out = env.rollout(torch.tensor(acts))
    for t in range(acts.shape[0]):
        got, want = out['obs'][t].cpu().numpy(), o.observation()
        # the drawn part (puck / circle state at a reset) must agree exactly; the arm is free-running from the exact
        # reset pose, which sits on a discontinuity of the reference algorithm, so it is loose
        ncmp = 4 if name == 'circle' else 6
        assert np.allclose(got[:, :ncmp], want[:, :ncmp], atol=1e-6 if name == 'circle' else 1e-9), t
        assert np.abs(got - want).max() < 5e-2, t
        _, _, ab, _ = o.step(acts[t])
        last = ab | (o.t >= horizon)
        if last.any():
            o.reset(last)
    assert not np.allclose(out['obs'][horizon].cpu().numpy()[:, :2], out['obs'][0].cpu().numpy()[:, :2])


@pytest.mark.mapping(name='iiwa')
@pytest.mark.parametrize('lanes', [1, 2, 4, 8])
@pytest.mark.parametrize('dt', ['f64', 'f32'])
def test_chart_on_slack_structured_matrices(dt, lanes):
    """The chart (null basis + rref with the 0.05 tolerance) on J_c-shaped inputs [K J | diag(s)] with small and
    near-zero slack entries -- the regime where columns are skipped and pivot rows are swapped -- against the oracle.
    float64: every entry; float32: every matrix within the float64 chart's own sensitivity (tests/parity_tools.py)."""
    from rl_on_manifold_amd import nullspace
    rng = np.random.default_rng(5)
    n, M, N = 6000, 12, 17
    A = rng.normal(size=(n, M, N))
    st = np.arange(n) % 3 != 0                                   # 2/3 structured like J_c, 1/3 dense
    A[st, :, 6:] = 0
    for g in range(11):
        A[st, 1 + g, 6 + g] = rng.uniform(-0.1, 0.3, st.sum())
    A[st, :, :6] = rng.normal(size=(st.sum(), M, 6)) * rng.uniform(0.01, 1, size=(st.sum(), 1, 6))
    x, nbm = ob.bidiag_solve_null(A, rng.normal(size=(n, M)) * 0, 5)
    ref = ob.rref_tol(nbm, 0.05)
    out = nullspace('iiwa', torch.tensor(A, device=DEV, dtype=DT[dt]), None, tol=0.05, lanes_per_env=lanes)
    err = np.abs(out[2].cpu().numpy().astype(np.float64) - ref).reshape(n, -1).max(1)
    scale = np.abs(ref).reshape(n, -1).max(1) + 1.0
    if dt == 'f64':
        assert (err / scale).max() < 1e-8
    else:
        from parity_tools import assert_matrix_fn_explained
        chart = lambda M_: ob.rref_tol(ob.bidiag_solve_null(M_, np.zeros((len(M_), M)), 5)[1], 0.05)   # noqa: E731
        print(assert_matrix_fn_explained(chart, A, out[2].cpu().numpy(), 'structured chart lanes %d' % lanes))


@pytest.mark.parametrize('name,key', [('planar', 'sac_planar'), ('iiwa', 'ppo_iiwa')])
def test_policy_rollout_matrix_core_path_ragged_batch(golden, name, key):
    """float + quad mapping evaluates the policy on the matrix cores, one wavefront = 16 environments; lanes past the
    end of a batch that is not a multiple of 16 shadow the last environment with their stores masked.  Checked
    against the one-env-per-lane kernel (VALU network) on a ragged batch with device-side random resets (the puck
    positions drawn at every reset depend on the per-env episode counter, so a shadow lane committing a reset would
    show): the same step / hit bookkeeping, and the same trajectories up to float32 summation order."""
    B, T = 203, 7                                      # 203 = 12 * 16 + 11: a partial last wavefront
    dev, _ = _policy_pair(golden, key)
    k = SHAPES[name][2]
    gen = torch.Generator(device=DEV); gen.manual_seed(5)
    eps = torch.randn((T, B, k), device=DEV, generator=gen)
    outs, states = [], []
    for lanes in (4, 2, 1):
        env = _env(name, B, 'f32', lanes_per_env=lanes, auto_reset=True, horizon=3, random_init=True, seed=11)
        outs.append(env.rollout_policy(dev, T, noise=eps))
        states.append(env.get_state())
    b = outs[-1]
    for a, sa in zip(outs[:-1], states[:-1]):
        assert torch.equal(a['last'], b['last'])
        assert torch.equal(sa[:, -1], states[-1][:, -1])                  # step counters
        for kk in ('obs', 'action', 'reward'):
            err = (a[kk] - b[kk]).abs().reshape(T, B, -1).amax(-1)
            assert float(err.median()) < 2e-5, kk
            assert float((err < 5e-3).float().mean()) >= 0.97, kk       # the rest: rref tolerance flips


@pytest.mark.mapping(name='iiwa')           # (its planar cases run on the planar census: a float64 request for 8 lanes runs 4)
@pytest.mark.parametrize('lanes', [1, 2, 4, 8])
@pytest.mark.parametrize('dt', ['f64', 'f32'])
def test_rank_deficient_inputs_stay_finite(golden, dt, lanes):
    """SURVEY.md H2 / DESIGN "Rank handling": pinv_null truncates singular values (null_space_coordinate.py:12-21); the
    kernels drop bidiagonal pivots below 8 eps (M + 5) max|d| instead of dividing by them.  Defined behaviour, checked:
    (a) the reference's own rank-deficient golden case and synthetic duplicated / zero rows give finite results that
    solve the consistent system and keep an orthonormal basis inside the null space;
    (b) an environment stepped from the straight-up singular pose q = 0 (the equality row of J_c is exactly zero there)
    and from zeroed slacks never puts NaN / Inf into its state planes."""
    from rl_on_manifold_amd import nullspace
    g = golden('nullspace')
    tol = 1e-9 if dt == 'f64' else 2e-3
    rng = np.random.default_rng(2)
    cases = {'planar': [g['rankdef_Jc']], 'iiwa': []}
    A = rng.normal(size=(12, 17)); A[7] = A[3]; cases['iiwa'].append(A)                       # duplicated row
    A = rng.normal(size=(12, 17)); A[0] = 0.0; cases['iiwa'].append(A)                         # zero row
    A = rng.normal(size=(12, 17)); A[5] = 2.0 * A[4] - A[9]; cases['iiwa'].append(A)          # dependent row
    A = rng.normal(size=(6, 9)); A[2] = -A[1]; cases['planar'].append(A)
    for name, mats in cases.items():
        A = np.array(mats)
        xs = rng.normal(size=(len(A), A.shape[2]))
        rhs = np.einsum('bcn,bn->bc', A, xs)                                                  # consistent by construction
        x, nb, rr = nullspace(name, torch.tensor(A, device=DEV, dtype=DT[dt]), torch.tensor(rhs, device=DEV, dtype=DT[dt]),
                              tol=0.05, lanes_per_env=lanes)
        x, nb, rr = x.cpu().numpy().astype(np.float64), nb.cpu().numpy().astype(np.float64), rr.cpu().numpy()
        assert np.isfinite(x).all() and np.isfinite(nb).all() and np.isfinite(rr).all()
        assert np.abs(np.einsum('bcn,bn->bc', A, x) - rhs).max() < tol * 50                   # solves the system
        assert np.abs(np.einsum('bcn,bnk->bck', A, nb)).max() < tol * 50                      # inside the null space
        gram = np.einsum('bnk,bnl->bkl', nb, nb)
        assert np.abs(gram - np.eye(nb.shape[2])).max() < tol * 50                            # orthonormal
    # (b) environment level
    B = 64
    env = _env('iiwa', B, dt, lanes_per_env=lanes)
    st = env.get_state()
    st[:, :12] = 0.0                                   # q = dq = 0: straight up, J_f = 0
    st[: B // 2, 12:23] = 0.0                          # and half of them with every slack at zero as well
    env.set_state(st)
    for t in range(6):
        obs, r, ab, _ = env.step(torch.rand((B, 5), device=DEV, dtype=DT[dt]) * 2 - 1)
        assert torch.isfinite(obs).all() and torch.isfinite(r).all()
        assert torch.isfinite(env.get_state()).all()
