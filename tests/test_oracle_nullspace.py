"""Oracle pinned to the reference: pinv_null / rref golden vectors (G1, G2) and the closed form of
LAPACK's null basis that the HIP kernels implement."""
import numpy as np
import pytest
from scipy import linalg

from oracle import nullspace as ns
from oracle import atacom_batched as ob

SHAPES = {'circle': (2, 3, 1), 'planar': (6, 9, 3), 'iiwa': (12, 17, 5)}


@pytest.mark.parametrize('name', list(SHAPES))
def test_pinv_null_matches_reference(golden, name):
    g = golden('nullspace')
    for Jc, B, Q, R in zip(g[name + '_Jc'], g[name + '_pinv'], g[name + '_null'], g[name + '_rref']):
        b, q = ns.pinv_null(Jc)
        assert np.allclose(b, B, rtol=0, atol=1e-12 * max(1.0, np.abs(B).max()))
        assert np.allclose(q, Q, rtol=0, atol=1e-12)
        r = ns.rref(q[:, :SHAPES[name][2]], row_vectors=False, tol=0.05)
        assert np.allclose(r, R, rtol=0, atol=1e-10 * max(1.0, np.abs(R).max()))


def test_rank_deficient_reference_behaviour(golden):
    g = golden('nullspace')
    b, q = ns.pinv_null(g['rankdef_Jc'])
    assert q.shape == g['rankdef_null'].shape == (9, 4) and int(g['rankdef_rank']) == 5
    assert np.allclose(b, g['rankdef_pinv'], atol=1e-12)


def test_rref_default_tolerance_equals_sympy_golden(golden):
    # the reference's own rref_test (null_space_coordinate.py:172-179): rref == sympy's rref
    g = golden('nullspace')
    for A, R in zip(g['rref_default_in'], g['rref_default_out']):
        m = int((~np.isnan(A[:, 0])).sum())
        n = int((~np.isnan(A[0, :])).sum())
        assert np.allclose(ns.rref(A[:m, :n]), R[:m, :n], atol=1e-9)


def test_rref_tolerance_branch(golden):
    g = golden('nullspace')
    for V, R in zip(g['rref_tol_in'], g['rref_tol_out']):
        assert np.allclose(ns.rref(V, row_vectors=False, tol=0.05), R, atol=1e-12)
    # batched restatement == scalar restatement, including the zeroing branch
    out = ob.rref_tol(g['rref_tol_in'], 0.05)
    assert np.allclose(out, g['rref_tol_out'], atol=1e-12)


@pytest.mark.parametrize('name', ['planar', 'iiwa', 'circle'])
def test_bidiagonal_null_basis_is_lapacks(golden, name):
    """vh[M:] of LAPACK's dgesdd == last N-M columns of the Householder bidiagonalisation's P."""
    g = golden('nullspace')
    c, n, k = SHAPES[name]
    for Jc, Q, B in zip(g[name + '_Jc'], g[name + '_null'], g[name + '_pinv']):
        nb = ns.bidiag_null(Jc)
        if k == 1:   # 2x3 goes down dgesdd's LQ path; a 1-d null space is unique up to sign anyway
            nb = nb * np.sign((nb * Q).sum())
        assert np.allclose(nb, Q, atol=1e-11)
        rhs = np.arange(1, c + 1, dtype=float)
        assert np.allclose(ns.bidiag_pinv_apply(Jc, rhs), B @ rhs, atol=1e-10 * max(1, np.abs(B).max()))
    rng = np.random.default_rng(0)
    for _ in range(50):
        A = rng.standard_normal((c, n))
        vh = linalg.svd(A, full_matrices=True)[2]
        nb = ns.bidiag_null(A)
        if k == 1:
            nb = nb * np.sign((nb[:, 0] * vh[c]).sum())
        assert np.allclose(nb, vh[c:].T, atol=1e-11)


@pytest.mark.parametrize('name', ['circle', 'planar', 'iiwa'])
def test_batched_solver_matches_golden(golden, name):
    g = golden('nullspace')
    c, n, k = SHAPES[name]
    Jc = g[name + '_Jc']
    rhs = np.tile(np.arange(1, c + 1, dtype=float), (len(Jc), 1))
    x, N = ob.bidiag_solve_null(Jc, rhs, k)
    xr = np.einsum('bnc,bc->bn', g[name + '_pinv'], rhs)
    assert np.allclose(x, xr, atol=1e-9 * max(1, np.abs(xr).max()))
    if k > 1:
        assert np.allclose(N, g[name + '_null'], atol=1e-11)
    R = ob.rref_tol(N, 0.05)
    assert np.allclose(R, g[name + '_rref'], atol=1e-9 * max(1, np.abs(g[name + '_rref']).max()))
    # invariants: Jc Jc^+ = I always; Jc N = 0 for the orthonormal basis
    assert np.abs(np.einsum('bcn,bnk->bck', Jc, N)).max() < 1e-12


def test_rref_on_forced_decisions_and_their_decoding():
    """tests/parity_tools.skip_pattern_of_rref reads the pivot / skip decisions off an rref'd basis, and
    rref_tol(forced_skip=...) replays them: on iiwa J_c matrices around the reset pose (40 % of which take the tolerance
    branch) decoding the oracle's own output and forcing it reproduces that output exactly; forcing the no-skip chart
    changes exactly the matrices that had skipped."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from parity_tools import skip_pattern_of_rref
    from oracle import atacom_scalar as osc, atacom_batched as ob
    rng = np.random.default_rng(0)
    spec = osc.iiwa_spec()
    B = 600
    q = np.array([0.0, 0.7135214629060707, 0.0, -0.5024756033561426, 0.0, 1.9256631778550268]) + rng.normal(0, 0.05, (B, 6))
    o = ob.BatchedAtacomEnv(spec, B, init_q=q)
    fun, J, _ = ob.constraint_terms(spec, o.q, o.dq)
    Jc = np.zeros((B, 12, 17))
    Jc[:, :, :6] = spec.K[None, :, None] * J + 0.0
    idx = np.arange(11)
    Jc[:, 1 + idx, 6 + idx] = o.s
    _, N = ob.bidiag_solve_null(Jc, np.zeros((B, 12)), 5)
    skipped = np.zeros(B, bool)
    Nr = ob.rref_tol(N, 0.05, None, skipped)
    pat = skip_pattern_of_rref(Nr)
    assert 0.2 < skipped.mean() < 0.6 and np.array_equal(pat.any(1), skipped)
    assert np.abs(ob.rref_tol(N, 0.05, forced_skip=pat) - Nr).max() == 0.0
    other = ob.rref_tol(N, 0.05, forced_skip=np.zeros_like(pat))
    assert np.array_equal(np.abs(other - Nr).max((1, 2)) > 1e-9, skipped)
