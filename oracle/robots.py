"""Kinematics of the two manipulators, float64 numpy (oracle; test infrastructure only).

The reference obtains these quantities from Pinocchio, which is neither vendored in
/root/reference nor installed in this image, so this file restates the published
algorithms (product-of-exponentials forward kinematics, geometric Jacobian in the
LOCAL_WORLD_ALIGNED convention, "classical acceleration") from the kinematic data the
reference does ship:

* iiwa chain: /root/reference/atacom/environments/iiwa_air_hockey/urdf/iiwa_1.urdf:69-301
  (joint origins xyz/rpy and axes), tip frame ``striker_rod_tip`` = joint-7 frame + (0,0,0.585)
  (env_base.py:147-151), frames 12 / 18 = link_4 / link_7 body frames (iiwa_hit_atacom.py:45-46).
* planar 3R arm: the URDF lives in MushroomRL (not in the tree); link lengths and
  limits below are the documented configuration of this build.

PARITY UNPINNED against Pinocchio itself; pinned by URDF known answers and finite differences
(tests/test_oracle_kinematics.py).

All functions broadcast over leading batch dimensions of ``q``.
"""
import numpy as np

# ----------------------------------------------------------------------------- iiwa chain
# (xyz, rpy) of joint_1 .. joint_7 origins, iiwa_1.urdf:72,110,147,184,221,258,295; all axes are +z
# (iiwa_1.urdf:73,111,148,185,222,259,296).
_HP = np.pi / 2
IIWA_JOINT_XYZ = np.array([
    [0.0, 0.0, 0.1575],
    [0.0, 0.0, 0.2025],
    [0.0, 0.2045, 0.0],
    [0.0, 0.0, 0.2155],
    [0.0, 0.1845, 0.0],
    [0.0, 0.0, 0.2155],
    [0.0, 0.081, 0.0],
])
IIWA_JOINT_RPY = np.array([
    [0.0, 0.0, 0.0],
    [_HP, 0.0, np.pi],
    [_HP, 0.0, np.pi],
    [_HP, 0.0, 0.0],
    [-_HP, np.pi, 0.0],
    [_HP, 0.0, 0.0],
    [-_HP, np.pi, 0.0],
])
IIWA_TIP_OFFSET = np.array([0.0, 0.0, 0.585])          # env_base.py:148
# upper position limits of joints 1..7 (lower = -upper), iiwa_1.urdf:74,112,149,186,223,260,297
IIWA_POS_LIMIT = np.array([2.9670597283903604, 2.0943951023931953, 2.9670597283903604,
                           2.0943951023931953, 2.9670597283903604, 2.0943951023931953,
                           3.0543261909900763])
# velocity limits, same lines
IIWA_VEL_LIMIT = np.array([1.4835298641951802, 1.4835298641951802, 1.7453292519943295,
                           1.3089969389957472, 2.2689280275926285, 2.356194490192345,
                           2.356194490192345])
IIWA_BASE_XYZ = np.array([-1.51, 0.0, -0.1])           # env_base.py:50
IIWA_N_CTRL = 6                                        # env_single.py:17-18 (isolated joint 7)


def _rpy_matrix(rpy):
    """URDF fixed-axis roll-pitch-yaw: R = Rz(yaw) Ry(pitch) Rx(roll), entries snapped to the
    exact signed permutation they are for this robot (all rpy are multiples of pi/2)."""
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    R = np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                  [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                  [-sp, cp * sr, cp * cr]])
    Rr = np.round(R)
    assert np.abs(R - Rr).max() < 1e-12
    return Rr + 0.0          # +0.0 turns -0.0 into +0.0


IIWA_JOINT_ROT = np.stack([_rpy_matrix(rpy) for rpy in IIWA_JOINT_RPY])


def _rotz(q):
    c, s = np.cos(q), np.sin(q)
    z, o = np.zeros_like(c), np.ones_like(c)
    return np.stack([np.stack([c, -s, z], -1), np.stack([s, c, z], -1), np.stack([z, z, o], -1)], -2)


def iiwa_chain(q, n_joints=7):
    """Joint frames of the iiwa chain expressed in the robot base frame.

    q: (..., nq) with nq <= 7; missing trailing joints are taken as 0 -- the reference pads the
    6 controlled joints with zeros for joint 7 and the two striker joints
    (iiwa_hit_atacom.py:65-68).
    Returns (R, o): R (..., 7, 3, 3) rotation of link_i frames, o (..., 7, 3) their origins
    (= joint origins; the joint axis is the frame's z axis).
    """
    q = np.asarray(q, dtype=np.float64)
    batch = q.shape[:-1]
    qq = np.zeros(batch + (7,))
    qq[..., :q.shape[-1]] = q
    R_prev = np.broadcast_to(np.eye(3), batch + (3, 3))
    o_prev = np.zeros(batch + (3,))
    Rs, os_ = [], []
    for i in range(n_joints):
        o_i = o_prev + R_prev @ IIWA_JOINT_XYZ[i]
        R_i = R_prev @ IIWA_JOINT_ROT[i] @ _rotz(qq[..., i])
        Rs.append(R_i)
        os_.append(o_i)
        R_prev, o_prev = R_i, o_i
    return np.stack(Rs, -3), np.stack(os_, -2)


# frame name -> (index of last supporting joint (0-based), offset in that joint frame)
IIWA_FRAMES = {
    'link_4': (3, np.zeros(3)),          # pinocchio frame 12 (iiwa_hit_atacom.py:45)
    'link_7': (6, np.zeros(3)),          # pinocchio frame 18 (iiwa_hit_atacom.py:46)
    'ee': (6, IIWA_TIP_OFFSET),          # 'striker_rod_tip', env_base.py:148-151
}


def iiwa_frame(q, frame):
    """Position (..., 3) and rotation (..., 3, 3) of a frame in the robot base frame
    (pinocchio framesForwardKinematics + oMf[...], iiwa_hit_atacom.py:72-73,95-96,102-103)."""
    j, off = IIWA_FRAMES[frame]
    R, o = iiwa_chain(q)
    return o[..., j, :] + R[..., j, :, :] @ off, R[..., j, :, :]


def iiwa_frame_jacobian(q, frame, n_cols=IIWA_N_CTRL):
    """6 x n_cols geometric Jacobian, LOCAL_WORLD_ALIGNED (linear rows 0-2 = d p / d q in base axes,
    angular rows 3-5 = joint axes), as pinocchio.computeFrameJacobian / getFrameJacobian return it
    (iiwa_hit_atacom.py:78-81,110-116).  Columns of joints past the frame's supporting joint are 0."""
    j, off = IIWA_FRAMES[frame]
    R, o = iiwa_chain(q)
    p = o[..., j, :] + R[..., j, :, :] @ off
    batch = p.shape[:-1]
    J = np.zeros(batch + (6, n_cols))
    for i in range(min(j + 1, n_cols)):
        z = R[..., i, :, 2]
        J[..., 0:3, i] = np.cross(z, p - o[..., i, :])
        J[..., 3:6, i] = z
    return J


def iiwa_frame_bias(q, dq, frame, mode='reference'):
    """Linear "classical acceleration" of a frame for zero joint acceleration, base axes.

    mode='reference': what the reference actually evaluates:
      pinocchio.forwardKinematics(model, data, q, dq) is the first-order call, so data.a stays at
      its zero initial value and getFrameClassicalAcceleration returns a_spatial(=0) + w x v,
      i.e. (angular velocity of the frame) x (linear velocity of the frame origin)
      (iiwa_hit_atacom.py:87-90,122-130; atacom_air_hockey.py:94-96).
    mode='exact': the true dJ/dt * dq of the linear Jacobian.
    """
    j, off = IIWA_FRAMES[frame]
    dq = np.asarray(dq, dtype=np.float64)
    R, o = iiwa_chain(q)
    p = o[..., j, :] + R[..., j, :, :] @ off
    batch = p.shape[:-1]
    dqq = np.zeros(batch + (7,))
    dqq[..., :dq.shape[-1]] = dq
    if mode == 'reference':
        w = np.zeros(batch + (3,))
        v = np.zeros(batch + (3,))
        for i in range(j + 1):
            z = R[..., i, :, 2]
            w = w + z * dqq[..., i:i + 1]
            v = v + np.cross(z, p - o[..., i, :]) * dqq[..., i:i + 1]
        return np.cross(w, v)
    if mode == 'exact':
        # d/dt sum_i z_i x (p - o_i) dq_i with dz_i/dt = w_i x z_i, d(p - o_i)/dt = v_p - v_oi
        w_i = np.zeros(batch + (3,))          # angular velocity of link i (after joint i)
        ws, vos = [], []
        for i in range(j + 1):
            z = R[..., i, :, 2]
            vo = np.zeros(batch + (3,))
            for k in range(i):
                zk = R[..., k, :, 2]
                vo = vo + np.cross(zk, o[..., i, :] - o[..., k, :]) * dqq[..., k:k + 1]
            w_i = w_i + z * dqq[..., i:i + 1]
            ws.append(w_i)
            vos.append(vo)
        vp = np.zeros(batch + (3,))
        for i in range(j + 1):
            vp = vp + np.cross(R[..., i, :, 2], p - o[..., i, :]) * dqq[..., i:i + 1]
        a = np.zeros(batch + (3,))
        for i in range(j + 1):
            z = R[..., i, :, 2]
            zdot = np.cross(ws[i], z)
            a = a + (np.cross(zdot, p - o[..., i, :]) + np.cross(z, vp - vos[i])) * dqq[..., i:i + 1]
        return a
    raise ValueError(mode)


# ----------------------------------------------------------------------------- planar 3R arm
# MushroomRL planar_robot_1.urdf is not in the reference tree; documented config:
PLANAR_LINK = np.array([0.55, 0.44, 0.44])
PLANAR_BASE_XYZ = np.array([-1.51, 0.0, -0.189])
PLANAR_POS_LIMIT = np.array([2.9670597283903604, 2.0943951023931953, 2.0943951023931953])
PLANAR_VEL_LIMIT = np.array([np.pi / 2, np.pi / 2, 2 * np.pi / 3])
PLANAR_INIT_Q = np.array([-0.9273, 0.9273, np.pi / 2])


def planar_fk(q):
    """End-effector xy in the robot base frame and the cumulative joint angles."""
    q = np.asarray(q, dtype=np.float64)
    th = np.cumsum(q, axis=-1)
    x = (PLANAR_LINK * np.cos(th)).sum(-1)
    y = (PLANAR_LINK * np.sin(th)).sum(-1)
    return np.stack([x, y], -1), th


def planar_jacobian(q):
    """2 x 3 linear Jacobian of the end effector (rows x, y), the [:2] block the reference takes
    from pinocchio.computeFrameJacobian(..., LOCAL_WORLD_ALIGNED) (atacom_air_hockey.py:88-89)."""
    q = np.asarray(q, dtype=np.float64)
    th = np.cumsum(q, axis=-1)
    sx = PLANAR_LINK * np.sin(th)
    cx = PLANAR_LINK * np.cos(th)
    J = np.zeros(q.shape[:-1] + (2, 3))
    for i in range(3):
        J[..., 0, i] = -sx[..., i:].sum(-1)
        J[..., 1, i] = cx[..., i:].sum(-1)
    return J


def planar_bias(q, dq, mode='reference'):
    """xy of the end effector's classical acceleration at zero joint acceleration
    (atacom_air_hockey.py:94-96).  'reference' = w x v (quirk Q2), 'exact' = dJ/dt dq."""
    q = np.asarray(q, dtype=np.float64)
    dq = np.asarray(dq, dtype=np.float64)
    J = planar_jacobian(q)
    v = np.einsum('...ij,...j->...i', J, dq)
    w = dq.sum(-1)
    if mode == 'reference':
        # (0,0,w) x (vx,vy,0) = (-w vy, w vx, 0)
        return np.stack([-w * v[..., 1], w * v[..., 0]], -1)
    th = np.cumsum(q, axis=-1)
    thd = np.cumsum(dq, axis=-1)
    ax = -(PLANAR_LINK * np.cos(th) * thd ** 2).sum(-1)
    ay = -(PLANAR_LINK * np.sin(th) * thd ** 2).sum(-1)
    return np.stack([ax, ay], -1)


# ----------------------------------------------------------------------------- CLIK (reset pose)
def _log3(R):
    """SO(3) logarithm (pinocchio.log3) for a single matrix."""
    tr = np.clip((np.trace(R) - 1.0) / 2.0, -1.0, 1.0)
    th = np.arccos(tr)
    if th < 1e-12:
        return np.zeros(3)
    if np.pi - th < 1e-6:
        # near pi: axis from the symmetric part
        A = (R + np.eye(3)) / 2.0
        ax = np.sqrt(np.maximum(np.diag(A), 0.0))
        k = int(np.argmax(ax))
        ax = A[:, k] / ax[k]
        return th * ax / np.linalg.norm(ax)
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return th / (2.0 * np.sin(th)) * w


def iiwa_clik(target_pos, target_rot, q0, it_max=1000, eps=1e-4, dt=1e-1, damp=1e-12):
    """Closed-loop inverse kinematics for the tip frame over the 7 arm joints, following
    kinematics.py:11-38 (error in the desired frame: -dMi.translation, log3(dMi.rotation), damped
    least-squares step on the LOCAL_WORLD_ALIGNED Jacobian, Euler integration, 2*pi wrap, limit check).
    Returns (success, q[7]).  Note: the reference starts from q = 0, a kinematic
    singularity, where its damped solve amplifies rounding noise; its converged branch therefore
    depends on Pinocchio's rounding and is unpinned.  Callers here pass an off-singular q0."""
    q = np.array(q0, dtype=np.float64).copy()
    success = False
    for _ in range(it_max + 1):
        p, R = iiwa_frame(q, 'ee')
        dR = target_rot.T @ R
        dp = target_rot.T @ (p - target_pos)
        err = np.concatenate([-dp, _log3(dR)])
        if np.linalg.norm(err) < eps:
            success = True
            break
        J = iiwa_frame_jacobian(q, 'ee', n_cols=7)
        v = -J.T @ np.linalg.solve(J @ J.T + damp * np.eye(6), err)
        q = q + v * dt
    hi = q > IIWA_POS_LIMIT
    q[hi] -= 2 * np.pi
    lo = q < -IIWA_POS_LIMIT
    q[lo] += 2 * np.pi
    if not (np.all(-IIWA_POS_LIMIT < q) and np.all(q < IIWA_POS_LIMIT)):
        return False, q
    return success, q
