"""Calibration run behind tests/parity_tools.py (not a test): teacher-forced float32 env steps against the float64 oracle,
with the oracle's own sensitivity to float32-sized perturbations (inputs + unstructured J_c noise) and its decision
margins.  Output is pasted into profiles/r02_parity_sensitivity.md.   python profiles/tools/gpu_sens_probe.py [lanes] [B] [T]
(MB_CHART=canonical: the opt-in chart, kernel and oracle both -- profiles/r03_sens_soak_canonical_l*.log)"""
import dataclasses
import os
import sys
import numpy as np
import torch
HERE = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests")      # parity_tools etc. live in tests/ (these probes lived there until round 6)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from oracle import atacom_scalar as osc, atacom_batched as ob
from rl_on_manifold_amd import BatchedAtacomEnv
from test_gpu_parity import _full_state, _step_outputs
from parity_tools import SensitivityRecorder, C_SENS, FLOOR

lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
T = int(sys.argv[3]) if len(sys.argv) > 3 else 40
CHART = os.environ.get('MB_CHART', 'reference')
DTYPE = torch.float64 if os.environ.get('MB_DTYPE') == 'f64' else torch.float32     # f64: the errors alone are the result
for name, spec in (('circle', osc.circle_spec()), ('planar', osc.planar_spec()), ('iiwa', osc.iiwa_spec())):
    spec = dataclasses.replace(spec, chart_mode=1 if CHART == 'canonical' else 0)
    env = BatchedAtacomEnv(name, B, device='cuda:0', dtype=DTYPE, lanes_per_env=lanes, chart_mode=CHART)
    nq, ng = spec.dim_q, spec.n_g
    st0 = env.get_state().cpu().numpy().astype(np.float64)
    rng = np.random.default_rng(int(os.environ.get('MB_SEED', '11')))      # 11: the seed of every committed soak but the *_seed23 ones
    o = ob.BatchedAtacomEnv(spec, B, init_q=st0[:, :nq] + (rng.normal(0, 0.05, (B, nq)) if name != 'circle' else 0.0))
    o.track_margins()
    rec = SensitivityRecorder(_step_outputs, seed=5)
    M, K = [], []
    for t in range(T):
        a = rng.uniform(-1.3, 1.3, (B, spec.n_null))
        a[: B // 8] = np.sign(a[: B // 8])
        env.set_state(_full_state(env, o))
        obs, r, ab, info = env.step(a)
        s_dev = env.get_state().cpu().numpy()[:, 2 * nq:2 * nq + ng]
        dev = np.concatenate([obs.cpu().numpy(), s_dev, r.cpu().numpy()[:, None], ab.cpu().numpy()[:, None] * 1.0], 1)
        rec.record(o, (a,), dev)
        res = o.step(a)
        M.append(o.decision_margin.copy()); K.append(o.cond_number.copy())
    E, S = np.array(rec.err), np.array(rec.sens)
    M, K = np.array(M), np.array(K)
    ratio = E / (C_SENS * S + FLOOR)
    print('== %s, %s chart, %d lanes per env, %s: %d env-steps' % (name, CHART, lanes, str(DTYPE).split('.')[-1], E.size))
    print('   err: median %.2e  p99 %.2e  p99.9 %.2e  max %.2e' % (np.median(E), np.quantile(E, .99), np.quantile(E, .999), E.max()))
    print('   quick sens (6 draws): median %.2e  p99 %.2e  max %.2e' % (np.median(S), np.quantile(S, .99), S.max()))
    print('   err / (C sens + floor): median %.3f  p99 %.3f  p99.9 %.3f  max %.3f ; above 1 (go to the deep probe): %d'
          % (np.median(ratio), np.quantile(ratio, .99), np.quantile(ratio, .999), ratio.max(), (ratio > 1).sum()))
    if name != 'circle':
        print('   oracle pivot margin < 1e-4: %.4f of env-steps, < 1e-3: %.4f ; cond(J_c) median %.0f max %.0f'
              % ((M < 1e-4).mean(), (M < 1e-3).mean(), np.median(K), K.max()))
        for lo, hi in ((0, 1e-5), (1e-5, 1e-4), (1e-4, 1e-3), (1e-3, 1e-2), (1e-2, np.inf)):
            m = (M >= lo) & (M < hi)
            if m.any():
                print('     margin [%.0e, %.0e): %7d samples, err max %.2e p99 %.2e' % (lo, hi, m.sum(), E[m].max(), np.quantile(E[m], .99)))
    if DTYPE == torch.float64:
        env.close()
        continue
    print('   ' + rec.finish('verdict'))
    env.close()
