"""The canonical chart's specification (oracle/canonical_chart.py) against the REFERENCE's chart and against its own
invariants -- CPU only.  The reference side is the oracle's restatement of pinv_null + rref(tol = 0.05), which is pinned to
the reference's golden vectors (tests/test_oracle_nullspace.py)."""
import numpy as np
import pytest

from chart_cases import rollout_systems, degenerate_systems, jc_of
from oracle import atacom_batched as ob
from oracle import canonical_chart as cc


@pytest.mark.parametrize('name', ['circle', 'planar', 'iiwa'])
def test_canonical_equals_reference_where_rref_takes_no_tolerance_branch(name):
    """Parity definition of the opt-in mode: on EVERY system where the reference's rref never skips a column, the
    canonical chart is the default chart as well -- unless its own decision is within the stated band of the tolerance --
    and mu is the reference's mu to 1e-9 (relative to max(1, |mu|))."""
    sy = rollout_systems(name)
    spec = sy['spec']
    k = spec.n_null
    rng = np.random.default_rng(2)
    alpha = rng.uniform(-10, 10, (len(sy['A']), k))
    x, _ = ob.bidiag_solve_null(sy['Jc'], sy['y'], k)
    mu_ref = -x + np.einsum('bnk,bk->bn', sy['Nr'], alpha)
    info = {}
    margin = np.full(len(alpha), np.inf)
    mu = cc.canonical_mu(sy['A'], sy['s'], sy['y'], alpha, spec.rref_tol, spec.n_f, margin=margin, info=info)
    clear = ~sy['skipped']
    agree = clear & info['default']
    err = np.abs(mu - mu_ref).max(1) / np.maximum(1.0, np.abs(mu_ref).max(1))
    assert err[agree].max() < 1e-9, err[agree].max()
    # where the two disagree on "default chart or not" the canonical decision sat close to its threshold: the tests
    # compare ||P_S e_j||^2 with tol^2 = 2.5e-3, the reference a max-norm of a non-orthonormal basis of the same space
    dis = clear & ~info['default']
    frac_clear, frac_dis = clear.mean(), dis.mean()
    print('%s: %d systems, reference clear %.3f, of those canonical-default %.4f; same free set overall: see '
          'profiles/r03_chart_agreement.md' % (name, len(alpha), frac_clear, agree.sum() / max(clear.sum(), 1)))
    assert frac_dis < 0.02, frac_dis
    assert frac_clear > {'circle': 0.02, 'planar': 0.8, 'iiwa': 0.5}[name]      # circle: the rollouts stay near (-1, 0), where y is the free coordinate


@pytest.mark.parametrize('name', ['circle', 'planar', 'iiwa'])
def test_canonical_chart_invariants_on_every_system(name):
    """Jc N = 0 and Jc mu + y = 0 on ALL systems (the reference's N_c leaks up to O(10) in its tolerance branch), N has an
    identity block on its free coordinates, which are sorted by column."""
    # tolerance: rounding on the systems the rollouts visit; on the hand-made ones 2e-5, because a row whose slack is below
    # 1e-6 of its Jacobian row is treated as an equality (w_g := target or 0), which neglects a motion of relative size s
    for sy, tol_inv in ((rollout_systems(name), 1e-9), (degenerate_systems(name), 2e-5)):
        spec = sy['spec']
        nf, k = spec.n_f, spec.n_null
        A, s, y = sy['A'], sy['s'], sy['y']
        Jc = jc_of(A, s, nf)
        rng = np.random.default_rng(3)
        alpha = rng.uniform(-10, 10, (len(A), k))
        info = {}
        mu = cc.canonical_mu(A, s, y, alpha, spec.rref_tol, nf, info=info)
        assert np.isfinite(mu).all()
        N, fcol = cc.null_basis(A, s, spec.rref_tol, nf)
        assert np.isfinite(N).all()
        scale = np.maximum(1.0, np.abs(N).max((1, 2)))
        assert (np.abs(np.einsum('bcn,bnk->bck', Jc, N)).max((1, 2)) / scale).max() < tol_inv
        # rows the algorithm must drop (a vanishing equality row; an all-zero row with a zero slack) are excluded from the
        # residual: the reference's pinv drops them too (singular value 0)
        live = (np.abs(Jc).max(2) > 0)
        if 'kind' in sy:            # two slacks at zero can make Jc mu = -y inconsistent (rows that are exact negatives of
            live &= (sy['kind'] != 5)[:, None]     # each other: least squares there, like the reference's pinv)
        res = np.abs(np.einsum('bcn,bn->bc', Jc, mu) + y) * live
        assert (res.max(1) / np.maximum(1.0, np.abs(y).max(1))).max() < 10 * tol_inv
        ok = (fcol >= 0).all(1)
        # (hand-made cases with two or three slacks at / near zero: fewer columns, see the module docstring)
        assert ok.mean() > (0.999 if tol_inv < 1e-8 else 0.5)
        ident = np.take_along_axis(N[ok], fcol[ok][:, :, None].repeat(k, 2), 1)
        assert np.allclose(ident, np.eye(k)[None], atol=1e-9)
        assert (np.diff(fcol[ok], axis=1) > 0).all()
        # mu = mu_min_norm + N alpha: the action part is exactly the chart's basis
        mu0 = cc.canonical_mu(A, s, y, np.zeros_like(alpha), spec.rref_tol, nf)
        assert np.allclose(mu[ok] - mu0[ok], np.einsum('bnk,bk->bn', N[ok], alpha[ok]), atol=1e-7 * scale[ok, None])


@pytest.mark.parametrize('name', ['planar', 'iiwa'])
def test_minimum_norm_part_is_the_pseudo_inverse(name):
    """mu(alpha = 0) = -Jc^+ y: checked against numpy's SVD pseudo-inverse wherever Jc is well conditioned."""
    sy = rollout_systems(name)
    spec = sy['spec']
    A, s, y = sy['A'][:600], sy['s'][:600], sy['y'][:600]
    Jc = jc_of(A, s, spec.n_f)
    mu0 = cc.canonical_mu(A, s, y, np.zeros((len(A), spec.n_null)), spec.rref_tol, spec.n_f)
    ref = -np.einsum('bnc,bc->bn', np.linalg.pinv(Jc), y)
    cond = np.linalg.cond(Jc)
    good = cond < 1e6
    assert good.mean() > 0.9
    err = np.abs(mu0 - ref).max(1) / np.maximum(1.0, np.abs(ref).max(1))
    assert err[good].max() < 1e-7, err[good].max()


def test_closed_loop_constraint_statistics_not_worse_than_the_reference_chart():
    """Free-running oracle, iiwa, perturbed reset poses: the canonical chart keeps the constraints at least as well as the
    reference's (whose tolerance branch leaks out of the null space and relies on the error correction)."""
    import copy
    from chart_cases import init_q, SPECS
    B, T = 192, 60
    rng = np.random.default_rng(5)
    q0 = init_q('iiwa', B, rng, sigma=0.03)
    acts = rng.uniform(-1, 1, (T, B, 5))
    stats = []
    for chart in (0, 1):
        sp = SPECS['iiwa']()
        sp.chart_mode = chart
        env = ob.BatchedAtacomEnv(sp, B, init_q=q0)
        for t in range(T):
            env.step(acts[t])
        stats.append(env.get_constraints_logs())
    (a0, m0, d0), (a1, m1, d1) = stats
    assert m1 <= 1.05 * m0 + 1e-6 and a1 <= 1.1 * a0 + 1e-6 and d1 <= 1e-6, stats


@pytest.mark.parametrize('name', ['circle', 'planar', 'iiwa'])
def test_specification_against_the_references_own_outputs(name):
    """Golden set G12: the imported reference's pinv_null + rref(tol = 0.05) + atacom.py:127-133 on J_c systems from
    rollouts.  Wherever the reference zeroed nothing (its N is an exact null basis) and its reduced echelon basis sits on
    the free coordinates the canonical chart picks, mu is the same vector -- the reduced echelon basis of a null space on
    given free coordinates and the minimum-norm solution are unique."""
    from chart_cases import reference_golden
    g = reference_golden(name)
    spec = g['spec']
    info = {}
    mu = cc.canonical_mu(g['A'], g['s'], g['y'], g['alpha'], spec.rref_tol, spec.n_f, info=info)
    same = g['exact'] & (g['free'] == info['fcol']).all(1)
    # (circle: near (-1, 0) the reference lives in its tolerance branch -- x is skipped, y becomes the free coordinate and
    # the skipped entry is zeroed: exact on 7 % of the rollout systems only)
    share = {'circle': 0.03, 'planar': 0.9, 'iiwa': 0.5}[name]
    print('%s: reference exact on %.3f of the systems, same free coordinates on %.3f, both %.3f' % (
        name, g['exact'].mean(), (g['free'] == info['fcol']).all(1).mean(), same.mean()))
    assert same.mean() > share, same.mean()
    err = np.abs(mu - g['mu']).max(1) / np.maximum(1.0, np.abs(g['mu']).max(1))
    assert err[same].max() < 1e-8, err[same].max()
    # and where the reference did zero entries its own N leaves the null space: the case the opt-in chart exists for
    if name == 'iiwa':
        assert (~g['exact']).mean() > 0.05
    N, _ = cc.null_basis(g['A'], g['s'], spec.rref_tol, spec.n_f)
    assert np.abs(np.einsum('bcn,bnk->bck', g['Jc'], N)).max() < 1e-9
