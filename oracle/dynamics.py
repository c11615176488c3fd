"""Rigid-body dynamics of the iiwa + striker chain, float64 numpy, batched (ORACLE: test infrastructure only) -- row N4.

The reference gets its joint torques from PyBullet's inverse dynamics on the nine movable joints of urdf/iiwa_1.urdf
(iiwa_hit_atacom.py:58-63: calculateInverseDynamics(q, dq, ddq padded with zeros)) and then lets Bullet integrate them
(stepSimulation; joint damping from the URDF, position-controlled joint 7 and universal joint, env_base.py:62-70,
env_single.py:137-185).  Bullet is not installed here, so this module restates the textbook algorithms on the constants
of oracle/iiwa_inertial.py (generated from the reference's URDF) and is pinned to golden set G11 -- the same URDF
evaluated link by link by the generic evaluator (tests/test_oracle_urdf.py):

  rnea(q, dq, ddq)        M(q) ddq + C(q, dq) dq + g(q)              recursive Newton-Euler, world coordinates
  mass_matrix(q)          M(q)                                       composite rigid bodies (spatial inertias about the
                                                                     world origin simply add up along the chain)
  forward_dynamics(...)   ddq of the six controlled joints for given torques, the three servo joints' accelerations
                          prescribed (hybrid dynamics): M_aa ddq_a = tau_a - rnea_a(q, dq, [0; ddq_b]) - D_a dq_a
  joint7_target / universal_joint_target    the servo set-points of env_single.py:137-185

All functions take q, dq, ... of shape [B, 9] = joints 1..7, striker_joint_1, striker_joint_2.
"""
import numpy as np

from . import robots
from . import iiwa_inertial as II

N_BODY = 9


# URDF effort limits of the six controlled joints (urdf/iiwa_1.urdf:74,112,149,186,223,260)
EFFORT_LIMIT = np.array([320.0, 320.0, 176.0, 176.0, 110.0, 40.0])


def _rot(axis, q):
    """Rotation by q about a coordinate axis ('x' / 'y' / 'z'), batched: [B, 3, 3]."""
    c, s = np.cos(q), np.sin(q)
    z, o = np.zeros_like(c), np.ones_like(c)
    rows = {'z': [[c, -s, z], [s, c, z], [z, z, o]], 'y': [[c, z, s], [z, o, z], [-s, z, c]],
            'x': [[o, z, z], [z, c, -s], [z, s, c]]}[axis]
    return np.stack([np.stack(r, -1) for r in rows], -2)


def chain9(q):
    """Frames of the nine moving bodies in the robot base frame: R [B, 9, 3, 3], origins o [B, 9, 3], joint axes a [B, 9, 3]
    (world).  Joints 1..7: oracle/robots.py; striker_joint_1 (axis y) and striker_joint_2 (axis x) sit at the tip point,
    0.585 m up the z axis of link_7 (iiwa_1.urdf:328-343,380-399)."""
    q = np.asarray(q, dtype=np.float64)
    R7, o7 = robots.iiwa_chain(q[:, :7])
    tip = o7[:, 6] + R7[:, 6] @ II.STRIKER_OFFSET
    R8 = R7[:, 6] @ _rot('y', q[:, 7])
    R9 = R8 @ _rot('x', q[:, 8])
    R = np.concatenate([R7, R8[:, None], R9[:, None]], 1)
    o = np.concatenate([o7, tip[:, None], tip[:, None]], 1)
    a = np.concatenate([R7[:, :, :, 2], (R7[:, 6] @ II.STRIKER_AXES[0])[:, None], (R8 @ II.STRIKER_AXES[1])[:, None]], 1)
    return R, o, a


def _world_inertials(R, o):
    c = o + np.einsum('bkij,kj->bki', R, II.COM)                          # centres of mass
    Iw = np.einsum('bkij,kjl,bkml->bkim', R, II.INERTIA, R)               # R I R^T
    return c, Iw


def rnea(q, dq, ddq, gravity=II.GRAVITY):
    """tau = M(q) ddq + C(q, dq) dq + g(q)  (no damping), [B, 9]."""
    q, dq, ddq = (np.asarray(x, dtype=np.float64) for x in (q, dq, ddq))
    B = q.shape[0]
    R, o, a = chain9(q)
    c, Iw = _world_inertials(R, o)
    w = np.zeros((B, 3)); al = np.zeros((B, 3))
    ao = np.broadcast_to(-np.asarray(gravity, dtype=np.float64), (B, 3)).copy()     # base "accelerates upward"
    op = np.zeros((B, 3))
    F, Nm = [], []
    for k in range(N_BODY):
        r = o[:, k] - op
        ao = ao + np.cross(al, r) + np.cross(w, np.cross(w, r))
        al = al + a[:, k] * ddq[:, k:k + 1] + np.cross(w, a[:, k] * dq[:, k:k + 1])
        w = w + a[:, k] * dq[:, k:k + 1]
        op = o[:, k]
        rc = c[:, k] - o[:, k]
        ac = ao + np.cross(al, rc) + np.cross(w, np.cross(w, rc))
        F.append(II.MASS[k] * ac)
        Nm.append(np.einsum('bij,bj->bi', Iw[:, k], al) + np.cross(w, np.einsum('bij,bj->bi', Iw[:, k], w)))
    tau = np.zeros((B, N_BODY))
    f = np.zeros((B, 3)); n = np.zeros((B, 3))            # wrench of the sub-chain, moment about the current joint origin
    for k in range(N_BODY - 1, -1, -1):
        if k < N_BODY - 1:
            n = n + np.cross(o[:, k + 1] - o[:, k], f)    # shift the child's moment to this joint's origin
        f = f + F[k]
        n = n + Nm[k] + np.cross(c[:, k] - o[:, k], F[k])
        tau[:, k] = (n * a[:, k]).sum(-1)
    return tau


def mass_matrix(q):
    """M(q) [B, 9, 9] by composite rigid bodies: the spatial inertia of body k about the WORLD origin is
    (m, h = m c, I_O = I_c + m ([c.c] 1 - c c^T)); composites are plain sums from the tip down."""
    q = np.asarray(q, dtype=np.float64)
    B = q.shape[0]
    R, o, a = chain9(q)
    c, Iw = _world_inertials(R, o)
    m = np.zeros(B); h = np.zeros((B, 3)); Io = np.zeros((B, 3, 3))
    M = np.zeros((B, N_BODY, N_BODY))
    eye = np.eye(3)
    for i in range(N_BODY - 1, -1, -1):
        ck = c[:, i]
        m = m + II.MASS[i]
        h = h + II.MASS[i] * ck
        Io = Io + Iw[:, i] + II.MASS[i] * ((ck * ck).sum(-1)[:, None, None] * eye - ck[:, :, None] * ck[:, None, :])
        vi = np.cross(o[:, i], a[:, i])                                     # velocity of the point at the world origin
        p = m[:, None] * vi + np.cross(a[:, i], h)                          # linear momentum of the composite
        L = np.einsum('bij,bj->bi', Io, a[:, i]) + np.cross(h, vi)          # angular momentum about the world origin
        for j in range(i + 1):
            vj = np.cross(o[:, j], a[:, j])
            M[:, i, j] = M[:, j, i] = (a[:, j] * L).sum(-1) + (vj * p).sum(-1)
    return M


def forward_dynamics(q, dq, tau_a, ddq_b, damping=II.DAMPING, n_ctrl=6):
    """Hybrid dynamics: accelerations of the controlled joints [B, n_ctrl] for given torques tau_a, the remaining joints
    (joint 7, universal joint) following prescribed accelerations ddq_b (position servos):
        M_aa ddq_a = tau_a - rnea_a(q, dq, [0; ddq_b]) - D_a dq_a."""
    q, dq = np.asarray(q, dtype=np.float64), np.asarray(dq, dtype=np.float64)
    B = q.shape[0]
    dd = np.zeros((B, N_BODY))
    dd[:, n_ctrl:] = ddq_b
    bias = rnea(q, dq, dd)
    rhs = tau_a - bias[:, :n_ctrl] - damping[:n_ctrl] * dq[:, :n_ctrl]
    Maa = mass_matrix(q)[:, :n_ctrl, :n_ctrl]
    return np.linalg.solve(Maa, rhs[:, :, None])[:, :, 0]


def energy(q, dq, gravity=II.GRAVITY):
    q, dq = np.asarray(q, dtype=np.float64), np.asarray(dq, dtype=np.float64)
    R, o, _ = chain9(q)
    c, _ = _world_inertials(R, o)
    pot = -(II.MASS[None, :, None] * c * np.asarray(gravity)).sum((1, 2))
    kin = 0.5 * np.einsum('bi,bij,bj->b', dq, mass_matrix(q), dq)
    return kin, pot


# ------------------------------------------------------------------ servo set-points (env_single.py:137-185)
def joint7_target(q7arm, q7_cur):
    """_compute_joint_7 (env_single.py:137-167): the joint-7 angle that keeps the striker's y axis horizontal.
    q7arm [B, 6] controlled joints (joint 7 taken as 0 for the forward kinematics, :139-142), q7_cur [B]."""
    qq = np.zeros((q7arm.shape[0], 7))
    qq[:, :6] = q7arm
    _, Rt = robots.iiwa_frame(qq, 'ee')
    zt, yt = Rt[:, :, 2], Rt[:, :, 1]
    z_axis = np.array([0.0, 0.0, -1.0])
    y_des = np.cross(z_axis, zt)
    nrm = np.linalg.norm(y_des, axis=-1)
    y_des = np.where((nrm > 1e-2)[:, None], y_des / np.where(nrm > 1e-2, nrm, 1.0)[:, None], zt)       # :146-150
    target = np.arccos(np.clip((yt * y_des).sum(-1), -1.0, 1.0))                                       # :152
    axis = np.cross(yt, y_des)
    an = np.linalg.norm(axis, axis=-1)
    axis = np.where((an > 1e-2)[:, None], axis / np.where(an > 1e-2, an, 1.0)[:, None], np.array([0.0, 0.0, 1.0]))
    target = target * (axis * zt).sum(-1)                                                              # :161
    target = np.where(target - q7_cur > np.pi / 2, target - np.pi, target)                             # :163-166
    target = np.where(target - q7_cur < -np.pi / 2, target + np.pi, target)
    return target


def universal_joint_target(q7full):
    """_compute_universal_joint (env_single.py:169-185): tilt of the link_ee z axis against the table normal.
    q7full [B, 7] (the link_ee frame has link_7's orientation)."""
    R, _ = robots.iiwa_chain(q7full)
    Rz, Ry = R[:, 6, :, 2], R[:, 6, :, 1]
    down = np.array([0.0, 0.0, -1.0])
    q1 = np.arccos(np.clip((Rz * down).sum(-1), -1.0, 1.0))
    axis = np.cross(Rz, down)
    an = np.linalg.norm(axis, axis=-1)
    axis = np.where((an > 1e-2)[:, None], axis / np.where(an > 1e-2, an, 1.0)[:, None], np.array([0.0, 0.0, 1.0]))
    q1 = q1 * (axis * Ry).sum(-1)
    return np.stack([q1, np.zeros_like(q1)], -1)
