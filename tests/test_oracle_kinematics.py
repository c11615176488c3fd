"""The kinematics restatement is PARITY-UNPINNED against Pinocchio (absent here); it is pinned by URDF known
answers, finite differences and the CLIK reset-pose property."""
import numpy as np

from oracle import robots as R
from oracle import atacom_scalar as osc

IIWA_INIT_Q = np.array([0.0, 0.7135214629060707, 0.0, -0.5024756033561426, 0.0, 1.9256631778550268])


def test_iiwa_known_answers():
    cases = [(np.zeros(7), (0, 0, 1.846), (0, 0, 0.780), (0, 0, 1.261)),
             (np.array([0, 0.5, 0, -1.0, 0, 0.8, 0]), (1.09700, 0, 0.31314), (0.20136, 0, 0.72858), (0.66076, 0, 0.70291)),
             (np.array([0.3, -0.4, 0.2, 1.1, -0.5, 0.9, 0.7]), (-0.74536, -0.59483, 1.28448),
              (-0.15625, -0.04833, 0.74685), (-0.53884, -0.27610, 0.83952))]
    for q, ee, l4, l7 in cases:
        assert np.allclose(R.iiwa_frame(q, 'ee')[0], ee, atol=1e-5)
        assert np.allclose(R.iiwa_frame(q, 'link_4')[0], l4, atol=1e-5)
        assert np.allclose(R.iiwa_frame(q, 'link_7')[0], l7, atol=1e-5)
    assert np.allclose(R.iiwa_frame(np.zeros(6), 'ee')[1], np.eye(3))
    # joint 7 does not move the tip point (quirk Q3)
    q = np.array([0.3, -0.4, 0.2, 1.1, -0.5, 0.9, 0.0])
    q2 = q.copy(); q2[6] = 1.3
    assert np.allclose(R.iiwa_frame(q, 'ee')[0], R.iiwa_frame(q2, 'ee')[0], atol=1e-14)


def test_jacobians_and_bias_by_finite_differences():
    rng = np.random.default_rng(0)
    for _ in range(5):
        q, dq = rng.uniform(-1, 1, 6), rng.uniform(-1, 1, 6)
        for fr in ('ee', 'link_4', 'link_7'):
            J = R.iiwa_frame_jacobian(q, fr)
            Jn = np.zeros((3, 6))
            for i in range(6):
                e = np.zeros(6); e[i] = 1e-6
                Jn[:, i] = (R.iiwa_frame(q + e, fr)[0] - R.iiwa_frame(q - e, fr)[0]) / 2e-6
            assert np.abs(J[:3] - Jn).max() < 1e-8
            h = 1e-6
            fd = (R.iiwa_frame_jacobian(q + h * dq, fr)[:3] - R.iiwa_frame_jacobian(q - h * dq, fr)[:3]) @ dq / (2 * h)
            assert np.abs(R.iiwa_frame_bias(q, dq, fr, 'exact') - fd).max() < 1e-7
            # reference-mode bias = w x v of the frame (quirk Q2); equals the exact one for single-joint motion
            w = J[3:] @ dq
            v = J[:3] @ dq
            assert np.allclose(R.iiwa_frame_bias(q, dq, fr, 'reference'), np.cross(w, v), atol=1e-13)
        d1 = np.zeros(6); d1[1] = 0.7
        assert np.allclose(R.iiwa_frame_bias(q, d1, 'ee', 'reference'), R.iiwa_frame_bias(q, d1, 'ee', 'exact'), atol=1e-12)
        q3, dq3 = rng.uniform(-1, 1, 3), rng.uniform(-1, 1, 3)
        Jp = R.planar_jacobian(q3)
        Jpn = np.stack([(R.planar_fk(q3 + e)[0] - R.planar_fk(q3 - e)[0]) / 2e-6 for e in np.eye(3) * 1e-6], -1)
        assert np.abs(Jp - Jpn).max() < 1e-8
        fd = (R.planar_jacobian(q3 + 1e-6 * dq3) - R.planar_jacobian(q3 - 1e-6 * dq3)) @ dq3 / 2e-6
        assert np.abs(R.planar_bias(q3, dq3, 'exact') - fd).max() < 1e-7


def test_reset_pose_property():
    """CLIK target (0.65, 0, 0.1505), R = diag(-1, 1, -1) (env_single.py:39-44): pinned as a property."""
    ok, q = R.iiwa_clik(np.array([0.65, 0.0, 0.1505]), np.diag([-1.0, 1.0, -1.0]), np.zeros(7))
    assert ok and np.allclose(q[:6], IIWA_INIT_Q, atol=1e-12)
    p, rot = R.iiwa_frame(q, 'ee')
    assert np.linalg.norm(p - [0.65, 0, 0.1505]) < 1e-4 and np.abs(rot - np.diag([-1.0, 1, -1])).max() < 1e-3
    assert np.all(np.abs(q) < R.IIWA_POS_LIMIT)
    spec = osc.iiwa_spec()
    fun, _, _ = osc.constraint_terms(spec, q[:6], np.zeros(6))
    assert abs(fun[0]) < 1e-4 and np.all(fun[1:] < 0)            # on the table plane, inside every inequality
    pp, _ = R.planar_fk(R.PLANAR_INIT_Q)
    assert np.allclose(pp + R.PLANAR_BASE_XYZ[:2], [-0.74, 0.0], atol=1e-4)
    fun, _, _ = osc.constraint_terms(osc.planar_spec(), R.PLANAR_INIT_Q, np.zeros(3))
    assert np.all(fun < 0)


def test_batched_terms_equal_scalar():
    from oracle import atacom_batched as ob
    rng = np.random.default_rng(1)
    for spec in (osc.planar_spec(), osc.iiwa_spec(), osc.iiwa_spec(bias_mode='exact'), osc.circle_spec()):
        q = rng.uniform(-1, 1, (7, spec.dim_q)); dq = rng.uniform(-1, 1, (7, spec.dim_q))
        fb, Jb, bb = ob.constraint_terms(spec, q, dq)
        for i in range(7):
            f, J, b = osc.constraint_terms(spec, q[i], dq[i])
            assert np.allclose(f, fb[i], atol=1e-13) and np.allclose(J, Jb[i], atol=1e-13) and np.allclose(b, bb[i], atol=1e-13)
