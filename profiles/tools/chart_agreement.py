"""How often does the canonical chart (oracle/canonical_chart.py) pick the free coordinates the REFERENCE's
rref(null(Jc), tol = 0.05) picks?  CPU only; systems sampled from oracle rollouts of the three tasks
(tests/chart_cases.py: 256 environments x 40 steps, perturbed reset poses, random actions).

    python profiles/tools/chart_agreement.py > profiles/r03_chart_agreement.md
"""
import os
import sys
from collections import Counter

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from chart_cases import rollout_systems, jc_of          # noqa: E402
from oracle import atacom_batched as ob                  # noqa: E402
from oracle import canonical_chart as cc                 # noqa: E402


def ref_pivots(N, tol):
    """pivot columns of the reference's rref (null_space_coordinate.py:40-79), per sample"""
    V = np.swapaxes(N, 1, 2).copy()
    B, m, n = V.shape
    piv = np.full((B, m), -1)
    i = np.zeros(B, int)
    ar = np.arange(B)
    rows = np.arange(m)[None, :]
    for j in range(n):
        act = i < m
        if not act.any():
            break
        col = np.where(rows >= i[:, None], np.abs(V[:, :, j]), -1.0)
        kk = col.argmax(1)
        p = col[ar, kk]
        pv = act & (p > tol)
        V[:, :, j] = np.where((act & ~pv)[:, None] & (rows >= i[:, None]), 0.0, V[:, :, j])
        b = ar[pv]
        if len(b):
            ib, kb = i[pv], kk[pv]
            tmp = V[b, ib, :].copy(); V[b, ib, :] = V[b, kb, :]; V[b, kb, :] = tmp
            prow = V[b, ib, :] / V[b, ib, j][:, None]
            colj = V[b, :, j].copy()
            V[b, :, :] -= colj[:, :, None] * prow[:, None, :]
            V[b, ib, :] = prow
            piv[b, ib] = j
            i[pv] += 1
    return piv


print('# Canonical chart vs the reference\'s rref(tol = 0.05): which coordinates are free?\n')
print('Systems: every 3rd (Jc, rhs) factorised during oracle rollouts (tests/chart_cases.py); random alpha in [-10, 10].')
print('"clear" = the reference\'s rref never took its tolerance branch (its N_c is then the exact reduced echelon basis).\n')
print('| task | systems | reference clear | canonical default chart | both | same free set (all systems) | max rel. diff of mu where both default | reference leak max abs(Jc N_c) | canonical leak max abs(Jc N) |')
print('|---|---|---|---|---|---|---|---|---|')
tops = {}
for name in ('circle', 'planar', 'iiwa'):
    sy = rollout_systems(name)
    spec = sy['spec']
    k, nf = spec.n_null, spec.n_f
    _, N = ob.bidiag_solve_null(sy['Jc'], sy['y'], k)
    piv = ref_pivots(N, spec.rref_tol)
    rng = np.random.default_rng(2)
    alpha = rng.uniform(-10, 10, (len(N), k))
    x, _ = ob.bidiag_solve_null(sy['Jc'], sy['y'], k)
    mu_ref = -x + np.einsum('bnk,bk->bn', sy['Nr'], alpha)
    info = {}
    mu = cc.canonical_mu(sy['A'], sy['s'], sy['y'], alpha, spec.rref_tol, nf, info=info)
    clear = ~sy['skipped']
    both = clear & info['default']
    same = (info['fcol'] == piv).all(1)
    err = np.abs(mu - mu_ref).max(1) / np.maximum(1.0, np.abs(mu_ref).max(1))
    Nc, _ = cc.null_basis(sy['A'], sy['s'], spec.rref_tol, nf)
    leak_ref = np.abs(np.einsum('bcn,bnk->bck', sy['Jc'], sy['Nr'])).max()
    leak_can = np.abs(np.einsum('bcn,bnk->bck', sy['Jc'], Nc)).max()
    print('| %s | %d | %.3f | %.3f | %.3f | %.4f | %.1e | %.2g | %.1e |' % (
        name, len(N), clear.mean(), info['default'].mean(), both.mean(), same.mean(), err[both].max(), leak_ref, leak_can))
    tops[name] = (Counter(map(tuple, piv.tolist())).most_common(6), Counter(map(tuple, info['fcol'].tolist())).most_common(6), len(N))
print()
for name, (a, b, n) in tops.items():
    print('%s -- most frequent free sets (column indices; joints first, then slacks), share of the systems:' % name)
    print('  reference: ' + ', '.join('%s %.3f' % (str(c), m / n) for c, m in a))
    print('  canonical: ' + ', '.join('%s %.3f' % (str(c), m / n) for c, m in b))
    print()
