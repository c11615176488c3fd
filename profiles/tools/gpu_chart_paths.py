"""Kernel-tuning helper (not a pytest file): what do the data-dependent paths of the canonical chart cost?

Times the k_chart primitive (one system per lane, 64 per wavefront) on batches in which every wavefront holds the same
mix: plain systems only / one system with a stiff row / several different stiff rows / one system one free coordinate
short (stage B) / the natural mix of a rollout."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import torch
from chart_cases import rollout_systems
from oracle import canonical_chart as cc
from rl_on_manifold_amd import canonical_mu

name = sys.argv[1] if len(sys.argv) > 1 else 'iiwa'
sy = rollout_systems(name, B=256, T=40, seed=0, stride=1)
spec = sy['spec']
A, s, y = sy['A'], sy['s'], sy['y']
n, k, nf = len(A), spec.n_null, spec.n_f
info = {}
cc.canonical_mu(A, s, y, np.zeros((n, k)), spec.rref_tol, nf, info=info)
arow = np.abs(A[:, nf:, :]).max(2)
stiff = np.abs(s) < cc.THETA * arow
plain = np.nonzero(~stiff.any(1) & (info['n_slack'] == 0))[0]
only_b = np.nonzero(~stiff.any(1) & (info['n_slack'] == 1))[0]
only_stiff = np.nonzero((stiff.sum(1) == 1) & (info['n_slack'] == 0))[0]
rng = np.random.default_rng(0)
W, NW = 64, 4096


def batch(special):
    """NW wavefronts of W systems: plain ones, with special[i] (arrays of system indices, one row per wavefront) in front"""
    idx = rng.choice(plain, (NW, W))
    if special is not None:
        idx[:, :special.shape[1]] = special
    return idx.reshape(-1)


def timeit(label, idx):
    t = lambda x: torch.tensor(x[idx], device='cuda:0', dtype=torch.float32)
    a, ss, yy = t(A), t(s), t(y)
    al = torch.zeros((len(idx), k), device='cuda:0')
    for _ in range(3):
        canonical_mu(name, a, ss, yy, al)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        canonical_mu(name, a, ss, yy, al)
    e1.record(); torch.cuda.synchronize()
    print('%-70s %7.1f us per launch of %d systems' % (label, e0.elapsed_time(e1) / 20 * 1e3, len(idx)), flush=True)


timeit('plain systems only', batch(None))
timeit('one system per wavefront with a stiff row', batch(rng.choice(only_stiff, (NW, 1))))
for m in (2, 4, 8):
    timeit('%d systems per wavefront with a stiff row (different rows)' % m, batch(rng.choice(only_stiff, (NW, m))))
timeit('one system per wavefront one coordinate short (stage B)', batch(rng.choice(only_b, (NW, 1))))
timeit('8 systems per wavefront one coordinate short (stage B)', batch(rng.choice(only_b, (NW, 8))))
timeit('natural mix of the rollout', rng.choice(n, NW * W))
print('shares in the rollout: stiff row %.3f, one short %.3f, plain %.3f' % (stiff.any(1).mean(), (info['n_slack'] == 1).mean(), len(plain) / n))
