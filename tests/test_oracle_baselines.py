"""Row N3: the circle experiment's baseline comparators, pinned to the reference's own CircleEnvErrorCorrection ('E')
and CircleEnvTerminated ('T') trajectories (golden set G9), teacher-forced step by step."""
import numpy as np
import pytest

from oracle import atacom_scalar as osc
from oracle import atacom_batched as ob


@pytest.mark.parametrize('tag', ['E', 'T'])
def test_baseline_wrappers_match_reference(golden, tag):
    g = golden('circle_baselines')
    spec = osc.circle_ec_spec(horizon=300) if tag == 'E' else osc.circle_t_spec(horizon=300)
    acts, obs, rew, absb, s, init = (g[tag + '_' + k] for k in ('actions', 'obs', 'reward', 'absorbing', 's', 'init'))
    n, T = acts.shape[:2]
    assert spec.action_dim == 2
    benv = ob.BatchedAtacomEnv(spec, n)
    for t in range(T):
        prev = init if t == 0 else obs[:, t - 1]
        if tag == 'E':
            s_prev = np.array([osc.slack_init(spec, p[:2], p[2:]) for p in init]) if t == 0 else s[:, t - 1]
            benv.set_state(prev[:, :2], prev[:, 2:], s_prev)
        else:
            benv.set_state(prev[:, :2], prev[:, 2:])
        o, r, ab, _ = benv.step(acts[:, t])
        assert np.allclose(o, obs[:, t], atol=1e-10), (t, np.abs(o - obs[:, t]).max())
        assert np.allclose(r, rew[:, t], atol=1e-10) and (ab == absb[:, t]).all()
        if tag == 'E':
            assert np.allclose(benv.s, s[:, t], atol=1e-10)
    # scalar oracle on the first trajectory, free-running for 50 steps, incl. the constraint log format
    e = osc.ScalarAtacomEnv(spec)
    e.reset()
    for t in range(50):
        o, r, ab, _ = e.step(acts[0, t])
        assert np.allclose(o, obs[0, t], atol=1e-8) and abs(r - rew[0, t]) < 1e-8 and ab == absb[0, t]
    if tag == 'T':
        assert absb.any() and (rew[absb] == -100).all()          # the terminated baseline really terminates
