"""Invariants of this build's puck / mallet / rim contact model (row N1; Bullet's own solver is unpinned, so the model is
checked against physics instead -- VERDICT r1): restitution along the contact normal, untouched tangential momentum, no
energy gain at e < 1, push-out to the contact distance, goal mouth, latch.  The HIP kernels are tested bit-for-bit-level
against this oracle in tests/test_gpu_parity.py::test_puck_contact_model_against_oracle."""
import numpy as np

from oracle import atacom_scalar as osc
from oracle import atacom_batched as ob
from oracle.atacom_scalar import PUCK_RADIUS, MALLET_RADIUS, TABLE_LENGTH, TABLE_WIDTH, GOAL_WIDTH, E_MALLET, E_RIM

R = PUCK_RADIUS + MALLET_RADIUS


def _env(B):
    return ob.BatchedAtacomEnv(osc.planar_spec(), B)


def test_mallet_impact_restitution_and_energy():
    rng = np.random.default_rng(0)
    B = 4000
    env = _env(B)
    dt = env.spec.dt
    mallet = rng.uniform(-0.3, 0.3, (B, 2))
    mvel = rng.uniform(-1.0, 1.0, (B, 2)) * (np.arange(B)[:, None] % 2)          # half static, half moving mallets
    ang = rng.uniform(0, 2 * np.pi, B)
    n0 = np.stack([np.cos(ang), np.sin(ang)], -1)
    env.puck[:, :2] = mallet + n0 * rng.uniform(0.9, 1.1, (B, 1)) * R            # around the contact distance
    env.puck[:, 3:5] = rng.uniform(-2, 2, (B, 2))
    before = env.puck.copy()
    env._puck_substep(mallet, mvel)
    moved = before[:, :2] + before[:, 3:5] * dt
    d = moved - mallet
    dist = np.hypot(d[:, 0], d[:, 1])
    n = d / dist[:, None]
    t = np.stack([-n[:, 1], n[:, 0]], -1)
    vrel0 = ((before[:, 3:5] - mvel) * n).sum(-1)
    vrel1 = ((env.puck[:, 3:5] - mvel) * n).sum(-1)
    hit = dist < R
    imp = hit & (vrel0 < 0)
    assert imp.sum() > 300 and (hit & ~imp).sum() > 100 and (~hit).sum() > 300
    # approaching contacts: the normal relative velocity is reversed and scaled by the restitution, the tangential one kept
    assert np.allclose(vrel1[imp], -E_MALLET * vrel0[imp], atol=1e-12)
    tr0 = ((before[:, 3:5] - mvel) * t).sum(-1)
    tr1 = ((env.puck[:, 3:5] - mvel) * t).sum(-1)
    assert np.allclose(tr1, tr0, atol=1e-12)
    # separating contacts and misses: velocity untouched
    assert np.array_equal(env.puck[~imp, 3:5], before[~imp, 3:5])
    # no energy gain in the mallet's frame (e < 1), exact loss (1 - e^2) of the normal part
    ke0 = ((before[:, 3:5] - mvel) ** 2).sum(-1)
    ke1 = ((env.puck[:, 3:5] - mvel) ** 2).sum(-1)
    assert (ke1 <= ke0 + 1e-12).all()
    assert np.allclose((ke0 - ke1)[imp], (1 - E_MALLET ** 2) * vrel0[imp] ** 2, atol=1e-12)
    # push-out: after a contact the puck sits exactly at the contact distance, along the same normal
    d1 = env.puck[:, :2] - mallet
    assert np.allclose(np.hypot(d1[:, 0], d1[:, 1])[hit], R, atol=1e-12)
    assert np.allclose((d1 / R)[hit], n[hit], atol=1e-12)
    assert np.allclose(env.puck[~hit, :2], moved[~hit], atol=0)


def test_rims_goal_mouth_and_latch():
    rng = np.random.default_rng(1)
    B = 4000
    env = _env(B)
    dt = env.spec.dt
    far = np.full((B, 2), 5.0)                                                  # mallet out of the way
    ylim, xlim = TABLE_WIDTH / 2 - PUCK_RADIUS, TABLE_LENGTH / 2 - PUCK_RADIUS
    env.puck[:, 0] = rng.uniform(-0.9, 0.9, B)
    env.puck[:, 1] = np.sign(rng.uniform(-1, 1, B)) * (ylim - rng.uniform(-0.004, 0.004, B))
    env.puck[:, 3:5] = rng.uniform(-2, 2, (B, 2))
    before = env.puck.copy()
    env._puck_substep(far, np.zeros((B, 2)))
    moved_y = before[:, 1] + before[:, 4] * dt
    out = np.abs(moved_y) > ylim
    outward = out & (before[:, 4] * np.sign(moved_y) > 0)
    assert out.sum() > 500 and outward.sum() > 300
    assert (np.abs(env.puck[:, 1]) <= ylim + 1e-12).all()                       # reflected back inside
    assert np.allclose(env.puck[outward, 4], -E_RIM * before[outward, 4])        # normal velocity reversed, e = 0.8
    assert np.array_equal(env.puck[:, 3], before[:, 3])                          # tangential velocity untouched (frictionless)
    assert (np.abs(env.puck[:, 4]) <= np.abs(before[:, 4]) + 1e-15).all()        # no energy gain
    # end rims: reflect outside the goal mouth, let the puck through inside it
    env2 = _env(B)
    env2.puck[:, 0] = xlim - rng.uniform(-0.003, 0.003, B)
    env2.puck[:, 1] = rng.uniform(-0.45, 0.45, B)
    env2.puck[:, 3] = rng.uniform(0.5, 3, B)
    env2.puck[:, 4] = 0.0
    b2 = env2.puck.copy()
    env2._puck_substep(far, np.zeros((B, 2)))
    past = (b2[:, 0] + b2[:, 3] * dt) > xlim
    mouth = np.abs(env2.puck[:, 1]) < GOAL_WIDTH
    assert (past & mouth).sum() > 200 and (past & ~mouth).sum() > 200
    assert np.array_equal(env2.puck[past & mouth, 3], b2[past & mouth, 3])       # through the goal mouth: untouched
    assert np.allclose(env2.puck[past & ~mouth, 3], -E_RIM * b2[past & ~mouth, 3])
    assert (env2.puck[past & ~mouth, 0] <= xlim + 1e-12).all()
    # the has_hit latch (env_hitting.py:80-85): set once the puck moves faster than 0.1, with the velocity at that moment
    assert env2.has_hit.all() and np.allclose(env2.vel_hit_x, env2.puck[:, 3])
    env3 = _env(8)
    env3.puck[:, 3] = 0.05
    env3._puck_substep(np.full((8, 2), 5.0), np.zeros((8, 2)))
    assert not env3.has_hit.any()
