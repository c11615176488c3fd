"""Golden set G10 / G11: the reference's own iiwa_1.urdf, evaluated by the generic URDF tree evaluator
(oracle/urdf_model.py via oracle/gen_golden.py urdf), pins the hand-unrolled kinematics of oracle/robots.py and the
constraint callables built on it.  The evaluator itself is pinned to textbook closed forms on a synthetic URDF."""
import numpy as np

from oracle import robots as R
from oracle import atacom_scalar as osc
from oracle.urdf_model import UrdfModel

PENDULUM = """<robot name="p2">
  <link name="base"/>
  <link name="l1"><inertial><origin xyz="0.5 0 0" rpy="0 0 0"/><mass value="2.0"/>
    <inertia ixx="0.01" ixy="0" ixz="0" iyy="0.02" iyz="0" izz="0.03"/></inertial></link>
  <link name="l2"><inertial><origin xyz="0.4 0 0" rpy="0 0 0"/><mass value="1.5"/>
    <inertia ixx="0.01" ixy="0" ixz="0" iyy="0.02" iyz="0" izz="0.05"/></inertial></link>
  <link name="tip"/>
  <joint name="j1" type="revolute"><parent link="base"/><child link="l1"/><origin xyz="0 0 0" rpy="0 0 0"/>
    <axis xyz="0 0 1"/><limit lower="-3" upper="3" velocity="2" effort="10"/></joint>
  <joint name="j2" type="revolute"><parent link="l1"/><child link="l2"/><origin xyz="1.0 0 0" rpy="0 0 0"/>
    <axis xyz="0 0 1"/><limit lower="-3" upper="3" velocity="2" effort="10"/></joint>
  <joint name="jt" type="fixed"><parent link="l2"/><child link="tip"/><origin xyz="0.8 0 0" rpy="0 0 0"/></joint>
</robot>"""


def test_evaluator_against_planar_double_pendulum_closed_forms():
    """2R arm in the xy plane, gravity along -y: textbook M(q), gravity vector, tip Jacobian and acceleration."""
    m = UrdfModel(PENDULUM)
    assert m.nq == 2
    m1, m2, l1, c1, c2, I1, I2 = 2.0, 1.5, 1.0, 0.5, 0.4, 0.03, 0.05
    rng = np.random.default_rng(0)
    for _ in range(10):
        q, dq, ddq = rng.uniform(-2, 2, 2), rng.uniform(-2, 2, 2), rng.uniform(-2, 2, 2)
        c, s12, c1_, c12 = np.cos(q[1]), np.sin(q[0] + q[1]), np.cos(q[0]), np.cos(q[0] + q[1])
        M = np.array([[I1 + I2 + m1 * c1 ** 2 + m2 * (l1 ** 2 + c2 ** 2 + 2 * l1 * c2 * c), I2 + m2 * (c2 ** 2 + l1 * c2 * c)],
                      [I2 + m2 * (c2 ** 2 + l1 * c2 * c), I2 + m2 * c2 ** 2]])
        assert np.allclose(m.mass_matrix(q), M, atol=1e-13)
        g = 9.81
        G = np.array([(m1 * c1 + m2 * l1) * g * c1_ + m2 * c2 * g * c12, m2 * c2 * g * c12])
        assert np.allclose(m.rnea(q, np.zeros(2), np.zeros(2), gravity=(0, -g, 0)), G, atol=1e-12)
        h = m2 * l1 * c2 * np.sin(q[1])
        C = np.array([-h * (2 * dq[0] * dq[1] + dq[1] ** 2), h * dq[0] ** 2])
        assert np.allclose(m.rnea(q, dq, ddq, gravity=(0, -g, 0)), M @ ddq + C + G, atol=1e-12)
        # tip = l2 + 0.8 x
        p, _ = m.frame(q, 'tip')
        assert np.allclose(p, [l1 * c1_ + 0.8 * c12, l1 * np.sin(q[0]) + 0.8 * s12, 0], atol=1e-14)
        J = m.frame_jacobian(q, 'tip')
        assert np.allclose(J[:2], [[-l1 * np.sin(q[0]) - 0.8 * s12, -0.8 * s12], [l1 * c1_ + 0.8 * c12, 0.8 * c12]], atol=1e-14)
        mo = m.frame_motion(q, dq, 'tip')
        w1, w12 = dq[0], dq[0] + dq[1]
        a = np.array([-l1 * c1_ * w1 ** 2 - 0.8 * c12 * w12 ** 2, -l1 * np.sin(q[0]) * w1 ** 2 - 0.8 * s12 * w12 ** 2, 0])
        assert np.allclose(mo['a_classical'], a, atol=1e-13)
        assert np.allclose(mo['v'], J[:3] @ dq, atol=1e-14) and np.allclose(mo['w_cross_v'], np.cross([0, 0, w12], mo['v']))


def test_robots_py_against_the_reference_urdf(golden):
    """oracle/robots.py (hand-transcribed chain) == the reference's URDF file under the generic evaluator."""
    g = golden('iiwa_urdf')
    q, dq = g['q'], g['dq']
    assert np.allclose(g['pos_upper'][:7], R.IIWA_POS_LIMIT, atol=0) and np.allclose(g['vel_limit'][:7], R.IIWA_VEL_LIMIT, atol=0)
    for fr in ('ee', 'link_4', 'link_7'):
        assert np.abs(R.iiwa_frame(q, fr)[0] - g[fr + '_pos']).max() < 1e-13
        assert np.abs(R.iiwa_frame_jacobian(q, fr) - g[fr + '_J']).max() < 1e-13
        assert np.abs(R.iiwa_frame_bias(q, dq, fr, 'reference') - g[fr + '_wxv']).max() < 1e-12
        assert np.abs(R.iiwa_frame_bias(q, dq, fr, 'exact') - g[fr + '_jdotqdot']).max() < 1e-12
    # SURVEY.md section 8c known answers hold for the file too
    assert np.allclose(g['ee_pos'][0], [0, 0, 1.846]) and np.allclose(g['link_4_pos'][0], [0, 0, 0.78])
    assert np.allclose(g['link_7_pos'][0], [0, 0, 1.261])
    assert abs(g['ee_pos'][1][2] - osc.UNIVERSAL_HEIGHT) < 1e-4 and abs(g['ee_pos'][1][0] - 0.65) < 1e-4     # the reset pose


def expected_terms(g, bias='reference'):
    """The reference's constraint callables (iiwa_hit_atacom.py:70-139) composed from the URDF-derived frames."""
    q, dq = g['q'], g['dq']
    n = len(q)
    pe, p4, p7 = g['ee_pos'], g['link_4_pos'], g['link_7_pos']
    Je, J4, J7 = g['ee_J'], g['link_4_J'], g['link_7_J']
    key = '_wxv' if bias == 'reference' else '_jdotqdot'
    ae, a4, a7 = g['ee' + key], g['link_4' + key], g['link_7' + key]
    lim = g['pos_upper'][:6]
    xw, yw = pe[:, 0] - 1.51, pe[:, 1]                                     # env_base.py:50
    fun = np.concatenate([np.stack([pe[:, 2] - 0.1505, -xw - 0.93, -yw - 0.46, yw - 0.46, -p4[:, 2] + 0.36,
                                    -p7[:, 2] + 0.25], -1), q ** 2 - lim ** 2], -1)
    J = np.zeros((n, 12, 6))
    J[:, 0], J[:, 1], J[:, 2], J[:, 3] = Je[:, 2], -Je[:, 0], -Je[:, 1], Je[:, 1]
    J[:, 4], J[:, 5] = -J4[:, 2], -J7[:, 2]
    J[:, 6 + np.arange(6), np.arange(6)] = 2 * q
    b = np.concatenate([np.stack([ae[:, 2], -ae[:, 0], -ae[:, 1], ae[:, 1], -a4[:, 2], -a7[:, 2]], -1), 2 * dq ** 2], -1)
    return fun, J, b


def test_oracle_constraint_terms_against_the_reference_urdf(golden):
    from oracle import atacom_batched as ob
    g = golden('iiwa_urdf')
    for bias in ('reference', 'exact'):
        fun, J, b = expected_terms(g, bias)
        fo, Jo, bo = ob.constraint_terms(osc.iiwa_spec(bias_mode=bias), g['q'], g['dq'])
        assert np.abs(fo - fun).max() < 1e-13 and np.abs(Jo - J).max() < 1e-13 and np.abs(bo - b).max() < 1e-12


def test_oracle_dynamics_against_the_reference_urdf(golden):
    """Row N4, golden set G11: inverse dynamics, mass matrix and energies of the nine-joint chain computed by the generic
    evaluator from the reference's URDF, link by link, vs oracle/dynamics.py on the merged-body constants."""
    from oracle import dynamics as D
    g = golden('iiwa_urdf')
    q, dq, ddq = g['dyn_q'], g['dyn_dq'], g['dyn_ddq']
    assert np.abs(D.rnea(q, dq, ddq) - g['dyn_tau']).max() < 1e-11
    assert np.abs(D.rnea(q, 0 * dq, 0 * dq) - g['dyn_gravity']).max() < 1e-11
    M = D.mass_matrix(q)
    assert np.abs(M - g['dyn_M']).max() < 1e-12
    assert np.abs(M - np.swapaxes(M, 1, 2)).max() < 1e-15 and np.linalg.eigvalsh(M).min() > 1e-4     # symmetric positive definite
    kin, pot = D.energy(q, dq)
    assert np.abs(kin - g['dyn_energy'][:, 0]).max() < 1e-12
    assert np.ptp(pot - g['dyn_energy'][:, 1]) < 1e-11          # up to the constant of the fixed base link (link_0)
    assert np.allclose(g['damping'], D.II.DAMPING)
    # inverse dynamics followed by forward dynamics is the identity (no damping, servo joints at rest)
    dd = np.concatenate([ddq[:, :6], np.zeros((len(q), 3))], 1)
    tau = D.rnea(q, dq, dd)
    back = D.forward_dynamics(q, dq, tau[:, :6], np.zeros((len(q), 3)), damping=np.zeros(9))
    assert np.abs(back - ddq[:, :6]).max() < 1e-10
    # damping only ever removes energy: with tau = gravity compensation the kinetic energy cannot grow
    rng = np.random.default_rng(0)
    qq, dd_ = q[:8].copy(), dq[:8].copy()
    dd_[:, 6:] = 0.0
    e0 = D.energy(qq, dd_)[0]
    for _ in range(50):
        grav = D.rnea(qq, 0 * dd_, 0 * dd_)[:, :6]
        coriolis = D.rnea(qq, dd_, 0 * dd_)[:, :6] - grav
        acc = D.forward_dynamics(qq, dd_, grav + coriolis * 0, np.zeros((8, 3)))
        dd_[:, :6] += acc / 240.0
        qq[:, :6] += dd_[:, :6] / 240.0
    assert (D.energy(qq, dd_)[0] <= e0 * (1 + 1e-3) + 1e-9).all()


def test_servo_targets():
    """env_single.py:137-185: at the reset pose (tip pointing straight down, R = diag(-1, 1, -1)) both servos rest at 0;
    tilting the last link tilts the universal joint by the same angle."""
    from oracle import dynamics as D
    q0 = np.array([[0.0, 0.7135214629060707, 0.0, -0.5024756033561426, 0.0, 1.9256631778550268, 0.0]])
    assert np.abs(D.joint7_target(q0[:, :6], np.zeros(1))).max() < 1e-3
    assert np.abs(D.universal_joint_target(q0)).max() < 1e-3
    q1 = q0.copy(); q1[0, 5] += 0.3
    u = D.universal_joint_target(q1)
    assert abs(abs(u[0, 0]) - 0.3) < 2e-3 and u[0, 1] == 0.0
    q2 = q0.copy(); q2[0, 4] = 0.7                       # rolling joint 5 turns the striker's y axis: joint 7 compensates
    t7 = D.joint7_target(q2[:, :6], np.zeros(1))
    assert 0.05 < abs(t7[0]) < np.pi / 2
