"""Diagnostic (not a test): where does the float32 canonical chart differ from its float64 specification?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import torch
from chart_cases import rollout_systems
from oracle import canonical_chart as cc
from rl_on_manifold_amd import canonical_mu

name = sys.argv[1] if len(sys.argv) > 1 else 'iiwa'
sy = rollout_systems(name)
spec = sy['spec']
A, s, y = sy['A'][:6000], sy['s'][:6000], sy['y'][:6000]
n, k, nf = len(A), spec.n_null, spec.n_f
rng = np.random.default_rng(4)
alpha = rng.uniform(-10, 10, (n, k))
margin = np.full(n, np.inf)
info = {}
ref = cc.canonical_mu(A, s, y, alpha, 0.05, nf, margin=margin, info=info)
t = lambda x, dt: torch.tensor(x, device='cuda:0', dtype=dt)
d32 = canonical_mu(name, t(A, torch.float32), t(s, torch.float32), t(y, torch.float32), t(alpha, torch.float32)).double().cpu().numpy()
d64 = canonical_mu(name, t(A, torch.float64), t(s, torch.float64), t(y, torch.float64), t(alpha, torch.float64)).cpu().numpy()
sc = np.maximum(1, np.abs(ref).max(1))
e32 = np.abs(d32 - ref).max(1) / sc
e64 = np.abs(d64 - ref).max(1) / sc
print('f64 max', e64.max(), 'f32 median', np.median(e32), 'p90', np.quantile(e32, .9), 'p99', np.quantile(e32, .99), 'max', e32.max())
arow = np.abs(A[:, nf:, :]).max(2)
rel_s = (np.abs(s) / arow).min(1)
nslack = info['n_slack']
for lo, hi in ((0, 1e-5), (1e-5, 1e-4), (1e-4, 1e-3), (1e-3, 1e-2), (1e-2, 1e9)):
    m = (e32 >= lo) & (e32 < hi)
    if m.any():
        print('err in [%g, %g): %5d  slack-chart frac %.2f  default frac %.2f  median margin %.2e  median min|s|/arow %.3f  max|mu| median %.1f' % (
            lo, hi, m.sum(), (nslack[m] > 0).mean(), info['default'][m].mean(), np.median(margin[m]), np.median(rel_s[m]), np.median(sc[m])))
w = np.argsort(-e32)[:6]
for i in w:
    j = np.abs(d32[i] - ref[i]).argmax()
    print('sample', i, 'err', e32[i], 'at', j, 'fcol', info['fcol'][i], 'margin', margin[i], 'ref', ref[i].round(3), 'd32', d32[i].round(3))
