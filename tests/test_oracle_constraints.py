"""Oracle pinned to the reference: constraint algebra (G3), acc_truncation / slack tables (G6), log
aggregation (G7)."""
import numpy as np

from oracle import atacom_scalar as osc
from oracle import atacom_batched as ob


def test_circle_constraint_algebra(golden):
    g = golden('constraints_circle')
    spec = osc.circle_spec()
    for i, (q, dq) in enumerate(zip(g['q'], g['dq'])):
        fun, J, b = osc.constraint_terms(spec, q, dq)
        Jdq = J @ dq
        c = fun + spec.K * Jdq                     # ViabilityConstraint.fun, constraints.py:33-37
        KJ = np.diag(spec.K) @ J                   # .K_J :39-40
        bb = Jdq + spec.K * b                      # .b :42-43
        assert np.allclose(c, [g['f_fun'][i][0], g['g_fun'][i][0]], atol=1e-13)
        assert np.allclose(fun, [g['f_fun_origin'][i][0], g['g_fun_origin'][i][0]], atol=1e-13)
        assert np.allclose(KJ, np.vstack([g['f_KJ'][i], g['g_KJ'][i]]), atol=1e-13)
        assert np.allclose(bb, [g['f_b'][i][0], g['g_b'][i][0]], atol=1e-13)
    fb, Jb, bb_ = ob.constraint_terms(spec, g['q'], g['dq'])
    assert np.allclose(fb[:, 0], g['f_fun_origin'][:, 0]) and np.allclose(fb[:, 1], g['g_fun_origin'][:, 0])


def test_truncation_and_slack_tables(golden):
    g = golden('tables')
    for name, spec in (('planar', osc.planar_spec()), ('iiwa', osc.iiwa_spec())):
        out = np.array([osc.acc_truncation(spec, a, b) for a, b in zip(g[name + '_trunc_dq'], g[name + '_trunc_ddq'])])
        assert np.allclose(out, g[name + '_trunc_out'], atol=1e-13)
        env = ob.BatchedAtacomEnv(spec, len(out))
        assert np.allclose(env.acc_truncation(g[name + '_trunc_dq'], g[name + '_trunc_ddq']), g[name + '_trunc_out'], atol=1e-13)
        s = np.array([osc.slack_init(spec, q, d) for q, d in zip(g[name + '_slack_q'], g[name + '_slack_dq'])])
        assert np.allclose(s, g[name + '_slack_s'], atol=1e-12)
        assert np.allclose(env.slack_init(g[name + '_slack_q'], g[name + '_slack_dq']), g[name + '_slack_s'], atol=1e-12)
        assert (g[name + '_slack_s'] == 0).any() and (g[name + '_slack_s'] > 0).any()   # both branches of max(., 0)


def test_log_aggregation(golden):
    g = golden('tables')
    env = osc.ScalarAtacomEnv(osc.planar_spec())
    env.logs = [row for row in g['agg_logs']]
    out = env.get_constraints_logs()
    assert np.allclose(out, g['agg_out'], atol=1e-14)
    assert env.logs == []                                   # cleared, atacom.py:213


def test_device_generator_restatement_statistics():
    """The counter-based generator used for on-device random resets: uniform, decorrelated across env / episode / draw."""
    u = ob.device_uniform(3, np.arange(200000), 0, 0)
    assert abs(u.mean() - 0.5) < 3e-3 and abs(u.std() - 12 ** -0.5) < 3e-3 and u.min() >= 0 and u.max() < 1
    v = ob.device_uniform(3, np.arange(200000), 1, 0)
    w = ob.device_uniform(3, np.arange(200000), 0, 1)
    assert abs(np.corrcoef(u, v)[0, 1]) < 0.01 and abs(np.corrcoef(u, w)[0, 1]) < 0.01
    env = ob.BatchedAtacomEnv(osc.circle_spec(), 64, random_init=True, seed=1)
    assert np.allclose((env.q ** 2).sum(1), 1) and np.abs((env.q * env.dq).sum(1)).max() < 1e-12
    assert (env.q[:, 1] >= -0.5).all() and (np.linalg.norm(env.dq, axis=1) <= 1).all()
