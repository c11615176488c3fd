"""Planar task 'D' on the CPU: known answers of the task logic the oracle restates (mushroom_rl's AirHockeyDefend
[upstream, restated from memory -- mushroom_rl is not in the reference tree; the reference only names it,
atacom_air_hockey.py:8,22-27, and runs it with horizon 180, examples/planar_air_hockey_exp.py:106-109]).  The values below
are the closed forms of its reward, evaluated by hand -- they pin the oracle's formulas and their branch order, not the
upstream source (parity unpinned)."""
import numpy as np

from oracle import atacom_scalar as osc
from oracle import atacom_batched as ob


def _env(B=1, **kw):
    return ob.BatchedAtacomEnv(osc.planar_spec(horizon=180, task=1), B, **kw)


def test_fixed_start_and_random_start_ranges():
    e = _env(4)
    assert np.allclose(e.puck, [[0.45, 0, 0, -1, 0, 0]] * 4)
    e = _env(4000, random_init=True, seed=9)
    p = e.puck
    assert p[:, 0].min() >= 0.25 and p[:, 0].max() <= 0.65 and np.abs(p[:, 1]).max() <= 0.4
    v = np.hypot(p[:, 3], p[:, 4])
    assert v.min() >= 1.0 and v.max() <= 2.2 and (p[:, 3] < 0).all()
    assert np.abs(np.arctan2(p[:, 4], -p[:, 3])).max() <= 0.5 and np.abs(p[:, 5]).max() <= 1.0
    assert abs(p[:, 0].mean() - 0.45) < 0.01 and abs(v.mean() - 1.6) < 0.03        # uniform draws


def test_reward_known_answers():
    e = _env(5)
    zero = np.zeros((5, 3))
    absorbing = np.zeros(5, dtype=bool)
    ee = ob.mallet_xy_world(e.spec, e.q)
    # 0: before any contact, puck at the mallet's y + 0.08: 0.3 exp(-3 |x_ee + 0.6|) + 0.7 / (2 sqrt(2 pi) 0.2)
    e.puck[0, :2] = [0.3, ee[0, 1] + 0.08]
    # 1: after a hit, puck resting at (-0.6, 0): r_x + r_y + r_vel + 1 = 1 + 3 + 5 + 1
    e.puck[1] = [-0.6, 0, 0, 0, 0, 0]; e.has_hit[1] = True
    # 2: after a hit but outside -0.8 < x < -0.4: nothing
    e.puck[2] = [-0.3, 0, 0, 0, 0, 0]; e.has_hit[2] = True
    # 3: after a bounce off the own end rim: -1, whatever else holds
    e.puck[3] = [-0.6, 0, 0, 0, 0, 0]; e.has_hit[3] = True; e.has_bounce[3] = True
    # 4: hit, moving: r_vel = 5 exp(-(5 * 0.2)^2), r_y = 3 exp(-0.3), r_x = exp(-0.5)
    e.puck[4] = [-0.5, 0.1, 0, 0.2, 0, 0]; e.has_hit[4] = True
    r = e._reward(zero, absorbing)
    want0 = 0.3 * np.exp(-3 * abs(ee[0, 0] + 0.6)) + 0.7 * 0.5 / (np.sqrt(2 * np.pi) * 0.2)
    assert np.allclose(r, [want0, 10.0, 0.0, -1.0, np.exp(-0.5) + 3 * np.exp(-0.3) + 5 * np.exp(-1.0) + 1], atol=1e-12)
    # the action penalty is on the scaled action (alpha = 10 a), as for task 'H' (env_hitting.py:68)
    assert np.isclose(e._reward(np.full((5, 3), 10.0), absorbing)[1], 10.0 - 1e-3 * np.sqrt(300.0))
    # absorbing: -50 in the agent's goal, 0 otherwise
    e.puck[0, :2] = [-0.99, 0.1]; e.puck[1, :2] = [-0.99, 0.3]; e.puck[2, :2] = [0.1, 0.0]
    r = e._reward(zero, np.ones(5, dtype=bool))
    assert np.allclose(r[:3], [-50.0, 0.0, 0.0])


def test_latches_and_termination():
    e = _env(3)
    ee = ob.mallet_xy_world(e.spec, e.q)
    # 0: a puck arriving at the mallet -> has_hit by CONTACT (not by speed as in task 'H': it is moving from the start)
    e.puck[0] = [ee[0, 0] + 0.09, ee[0, 1], 0, -1.0, 0, 0]
    # 1: a puck about to reach the agent-side end rim beside the goal mouth -> has_bounce
    e.puck[1] = [-0.94, 0.4, 0, -2.0, 0, 0]
    # 2: the same towards the goal mouth: no rim there, it leaves the table
    e.puck[2] = [-0.94, 0.0, 0, -2.0, 0, 0]
    assert not e.has_hit.any()                      # a moving puck does not latch by itself
    for _ in range(3):
        _, r, ab, _ = e.step(np.zeros((3, 3)))
    assert e.has_hit.tolist() == [True, False, False] and e.has_bounce.tolist() == [False, True, False]
    assert e.puck[0, 3] > 0 and e.puck[1, 3] > 0     # both were sent back
    assert ab.tolist() == [False, False, True] and r[2] == -50.0
    # hit or bounced, and back in the opponent's half -> absorbing, reward 0
    e.puck[0, 0] = 0.01; e.puck[1, 0] = 0.01
    assert e._is_absorbing()[:2].tolist() == [True, True]
    e.has_hit[0] = False
    assert e._is_absorbing()[:2].tolist() == [False, True]
