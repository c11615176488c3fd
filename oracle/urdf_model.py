"""A generic URDF kinematic-tree evaluator in float64 numpy (ORACLE code: test infrastructure only).

Purpose: turn a URDF *file* -- in particular the one the reference ships,
/root/reference/atacom/environments/iiwa_air_hockey/urdf/iiwa_1.urdf -- into forward kinematics, frame Jacobians,
frame accelerations and rigid-body dynamics WITHOUT any hand-transcribed constant, so that
oracle/robots.py (hand-unrolled chain) and the HIP kernels can be pinned to a reference-held file
(oracle/gen_golden.py, golden sets G10 / G11).  The reference gets these quantities from Pinocchio and
PyBullet, which are not installed here; this module restates the textbook algorithms they implement
(Featherstone, "Rigid Body Dynamics Algorithms", 2008: recursive Newton-Euler ch. 5, composite rigid body ch. 6)
with the conventions of the reference's call sites:

  * frames / Jacobians in pinocchio.LOCAL_WORLD_ALIGNED (origin at the frame, axes of the world):
    iiwa_hit_atacom.py:78-81,110-116;
  * getFrameClassicalAcceleration = spatial acceleration + w x v (iiwa_hit_atacom.py:87-90,122-130); the
    reference calls the first-order forwardKinematics(q, dq), so its spatial part is zero (quirk Q2);
  * PyBullet calculateInverseDynamics(q, dq, ddq) = M(q) ddq + C(q, dq) dq + g(q), gravity (0, 0, -9.81),
    joint damping NOT included (iiwa_hit_atacom.py:58-63).

Everything is written for ONE configuration at a time with plain loops over the tree (clarity over speed:
it only generates fixtures and checks the vectorised restatement).
"""
import xml.etree.ElementTree as ET

import numpy as np


def rpy_matrix(rpy):
    """URDF fixed-axis roll-pitch-yaw: R = Rz(yaw) Ry(pitch) Rx(roll)."""
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def axis_angle_matrix(axis, angle):
    """Rodrigues: rotation by `angle` about the unit vector `axis`."""
    a = np.asarray(axis, dtype=np.float64)
    a = a / np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


def _vec(s, default=(0.0, 0.0, 0.0)):
    return np.array([float(x) for x in s.split()]) if s is not None else np.array(default, dtype=np.float64)


class UrdfModel:
    """Kinematic tree of a URDF: `joints` in document order, movable joints (revolute / continuous / prismatic)
    numbered in that order -- for a serial chain this is Pinocchio's joint order (names[1:])."""

    def __init__(self, path_or_xml):
        text = path_or_xml if path_or_xml.lstrip().startswith('<') else open(path_or_xml).read()
        root = ET.fromstring(text)
        self.links = {}
        for ln in root.findall('link'):
            ine = ln.find('inertial')
            if ine is None:
                self.links[ln.get('name')] = None
                continue
            org = ine.find('origin')
            mass = float(ine.find('mass').get('value'))
            it = ine.find('inertia')
            g = lambda k: float(it.get(k, 0.0))          # noqa: E731
            I = np.array([[g('ixx'), g('ixy'), g('ixz')], [g('ixy'), g('iyy'), g('iyz')], [g('ixz'), g('iyz'), g('izz')]])
            self.links[ln.get('name')] = {
                'mass': mass, 'com': _vec(org.get('xyz') if org is not None else None),
                'rpy': _vec(org.get('rpy') if org is not None else None), 'I': I}
        self.joints = []
        children = set()
        for jn in root.findall('joint'):
            org = jn.find('origin')
            ax = jn.find('axis')
            lim = jn.find('limit')
            dyn = jn.find('dynamics')
            j = {'name': jn.get('name'), 'type': jn.get('type'), 'parent': jn.find('parent').get('link'),
                 'child': jn.find('child').get('link'),
                 'xyz': _vec(org.get('xyz') if org is not None else None),
                 'rpy': _vec(org.get('rpy') if org is not None else None),
                 'axis': _vec(ax.get('xyz') if ax is not None else None, (1.0, 0.0, 0.0)),
                 'lower': float(lim.get('lower', 0)) if lim is not None else None,
                 'upper': float(lim.get('upper', 0)) if lim is not None else None,
                 'velocity': float(lim.get('velocity', 0)) if lim is not None else None,
                 'effort': float(lim.get('effort', 0)) if lim is not None else None,
                 'damping': float(dyn.get('damping', 0)) if dyn is not None else 0.0,
                 'friction': float(dyn.get('friction', 0)) if dyn is not None else 0.0}
            self.joints.append(j)
            children.add(j['child'])
        roots = [n for n in self.links if n not in children]
        assert len(roots) == 1, roots
        self.root = roots[0]
        self.joint_of_child = {j['child']: j for j in self.joints}
        self.movable = [j for j in self.joints if j['type'] in ('revolute', 'continuous', 'prismatic')]
        for i, j in enumerate(self.movable):
            j['index'] = i
        self.nq = len(self.movable)
        # topological order of links from the root
        self.order = [self.root]
        k = 0
        while k < len(self.order):
            self.order += [j['child'] for j in self.joints if j['parent'] == self.order[k]]
            k += 1

    # ------------------------------------------------------------------ kinematics
    def _q(self, q):
        qq = np.zeros(self.nq)
        q = np.asarray(q, dtype=np.float64)
        qq[:len(q)] = q                                  # the reference pads with zeros (iiwa_hit_atacom.py:65-68)
        return qq

    def link_frames(self, q):
        """World placement (R, p) of every link frame."""
        q = self._q(q)
        T = {self.root: (np.eye(3), np.zeros(3))}
        for ln in self.order[1:]:
            j = self.joint_of_child[ln]
            Rp, pp = T[j['parent']]
            R = Rp @ rpy_matrix(j['rpy'])
            p = pp + Rp @ j['xyz']
            if j['type'] in ('revolute', 'continuous'):
                R = R @ axis_angle_matrix(j['axis'], q[j['index']])
            elif j['type'] == 'prismatic':
                p = p + R @ (j['axis'] * q[j['index']])
            T[ln] = (R, p)
        return T

    def chain(self, link):
        """Movable joints on the path root -> link, root first."""
        out = []
        while link != self.root:
            j = self.joint_of_child[link]
            if 'index' in j:
                out.append(j)
            link = j['parent']
        return out[::-1]

    def frame(self, q, link, offset=(0.0, 0.0, 0.0)):
        """Position and rotation of the frame `link` + local translation `offset` (a Pinocchio body frame /
        addBodyFrame placement, env_base.py:147-151)."""
        R, p = self.link_frames(q)[link]
        return p + R @ np.asarray(offset, dtype=np.float64), R

    def frame_jacobian(self, q, link, offset=(0.0, 0.0, 0.0)):
        """6 x nq Jacobian, LOCAL_WORLD_ALIGNED: rows 0-2 velocity of the frame origin, rows 3-5 angular velocity,
        both in world axes (pinocchio.getFrameJacobian / computeFrameJacobian)."""
        T = self.link_frames(q)
        R, p = T[link]
        p = p + R @ np.asarray(offset, dtype=np.float64)
        J = np.zeros((6, self.nq))
        for j in self.chain(link):
            Rj, oj = T[j['child']]
            a = Rj @ (j['axis'] / np.linalg.norm(j['axis']))
            if j['type'] == 'prismatic':
                J[0:3, j['index']] = a
            else:
                J[0:3, j['index']] = np.cross(a, p - oj)
                J[3:6, j['index']] = a
        return J

    def frame_motion(self, q, dq, link, offset=(0.0, 0.0, 0.0), ddq=None):
        """Velocity / acceleration of the frame by propagating link motion down the tree (NOT via the Jacobian):
        returns dict(v, w: linear / angular velocity of the frame origin; a_classical: d/dt v for the given ddq
        (default 0), i.e. the true dJ/dt dq + J ddq; w_cross_v: what getFrameClassicalAcceleration returns when the
        spatial acceleration is zero -- the reference's bias term, quirk Q2)."""
        q, dq = self._q(q), self._q(dq)
        ddq = np.zeros(self.nq) if ddq is None else self._q(ddq)
        T = self.link_frames(q)
        # per link: angular velocity w, velocity of the link-frame origin v, angular acceleration al, acceleration of
        # the link-frame origin a (all world axes, classical = material derivative of the point's velocity)
        mot = {self.root: (np.zeros(3), np.zeros(3), np.zeros(3), np.zeros(3))}
        for ln in self.order[1:]:
            j = self.joint_of_child[ln]
            wp, vp, alp, ap = mot[j['parent']]
            Rp, pp = T[j['parent']]
            R, p = T[ln]
            r = p - pp
            v = vp + np.cross(wp, r)
            a = ap + np.cross(alp, r) + np.cross(wp, np.cross(wp, r))
            w, al = wp, alp
            if 'index' in j:
                ax = R @ (j['axis'] / np.linalg.norm(j['axis']))
                qd, qdd = dq[j['index']], ddq[j['index']]
                if j['type'] == 'prismatic':
                    # the link-frame origin slides along ax
                    v = v + ax * qd
                    a = a + ax * qdd + 2 * np.cross(wp, ax * qd)
                else:
                    w = wp + ax * qd
                    al = alp + ax * qdd + np.cross(wp, ax * qd)
            mot[ln] = (w, v, al, a)
        w, v, al, a = mot[link]
        R, p = T[link]
        r = R @ np.asarray(offset, dtype=np.float64)
        vf = v + np.cross(w, r)
        af = a + np.cross(al, r) + np.cross(w, np.cross(w, r))
        return {'v': vf, 'w': w, 'a_classical': af, 'w_cross_v': np.cross(w, vf)}

    # ------------------------------------------------------------------ dynamics
    def _inertials(self, T):
        """Per link with mass: (mass, world COM, world inertia about the COM)."""
        out = {}
        for ln, ine in self.links.items():
            if ine is None or ine['mass'] == 0.0:
                continue
            R, p = T[ln]
            Ri = R @ rpy_matrix(ine['rpy'])
            out[ln] = (ine['mass'], p + R @ ine['com'], Ri @ ine['I'] @ Ri.T)
        return out

    def _moving_link(self, link):
        """The link whose motion a (possibly fixed-attached) link shares: nearest ancestor-or-self that is the child of
        a movable joint (or the root)."""
        while link != self.root and 'index' not in self.joint_of_child[link]:
            link = self.joint_of_child[link]['parent']
        return link

    def rnea(self, q, dq, ddq, gravity=(0.0, 0.0, -9.81)):
        """Inverse dynamics tau = M(q) ddq + C(q, dq) dq + g(q) by the recursive Newton-Euler algorithm written in
        world coordinates about each body's centre of mass (what PyBullet's calculateInverseDynamics evaluates;
        joint damping / friction are not part of it)."""
        q, dq, ddq = self._q(q), self._q(dq), self._q(ddq)
        g = np.asarray(gravity, dtype=np.float64)
        T = self.link_frames(q)
        ine = self._inertials(T)
        # forward pass: motion of every link frame
        mot = {self.root: (np.zeros(3), np.zeros(3), np.zeros(3), -g)}     # base accelerates upward by -g
        for ln in self.order[1:]:
            j = self.joint_of_child[ln]
            wp, vp, alp, ap = mot[j['parent']]
            r = T[ln][1] - T[j['parent']][1]
            v = vp + np.cross(wp, r)
            a = ap + np.cross(alp, r) + np.cross(wp, np.cross(wp, r))
            w, al = wp, alp
            if 'index' in j:
                ax = T[ln][0] @ (j['axis'] / np.linalg.norm(j['axis']))
                qd, qdd = dq[j['index']], ddq[j['index']]
                if j['type'] == 'prismatic':
                    v = v + ax * qd
                    a = a + ax * qdd + 2 * np.cross(wp, ax * qd)
                else:
                    w = wp + ax * qd
                    al = alp + ax * qdd + np.cross(wp, ax * qd)
            mot[ln] = (w, v, al, a)
        # per-body net force / moment about the COM
        F, N = {}, {}
        for ln, (m, c, I) in ine.items():
            w, v, al, a = mot[ln]
            rc = c - T[ln][1]
            ac = a + np.cross(al, rc) + np.cross(w, np.cross(w, rc))
            F[ln] = m * ac
            N[ln] = I @ al + np.cross(w, I @ w)
        # backward pass: accumulate wrenches (force f, moment n about the LINK-FRAME origin) towards the root
        f = {ln: np.zeros(3) for ln in self.order}
        n = {ln: np.zeros(3) for ln in self.order}
        tau = np.zeros(self.nq)
        for ln in self.order[::-1]:
            if ln in F:
                f[ln] = f[ln] + F[ln]
                n[ln] = n[ln] + N[ln] + np.cross(ine[ln][1] - T[ln][1], F[ln])
            if ln == self.root:
                continue
            j = self.joint_of_child[ln]
            if 'index' in j:
                ax = T[ln][0] @ (j['axis'] / np.linalg.norm(j['axis']))
                tau[j['index']] = (f[ln] if j['type'] == 'prismatic' else n[ln]) @ ax
            par = j['parent']
            f[par] = f[par] + f[ln]
            n[par] = n[par] + n[ln] + np.cross(T[ln][1] - T[par][1], f[ln])
        return tau

    def mass_matrix(self, q):
        """Joint-space inertia M(q) = sum over bodies of  m Jv^T Jv + Jw^T I Jw  (COM Jacobians) -- an independent
        route from rnea (tests check rnea(q, 0, e_i, g = 0) == M[:, i])."""
        q = self._q(q)
        T = self.link_frames(q)
        M = np.zeros((self.nq, self.nq))
        for ln, (m, c, I) in self._inertials(T).items():
            Jv = np.zeros((3, self.nq))
            Jw = np.zeros((3, self.nq))
            for j in self.chain(ln):
                Rj, oj = T[j['child']]
                a = Rj @ (j['axis'] / np.linalg.norm(j['axis']))
                if j['type'] == 'prismatic':
                    Jv[:, j['index']] = a
                else:
                    Jv[:, j['index']] = np.cross(a, c - oj)
                    Jw[:, j['index']] = a
            M += m * Jv.T @ Jv + Jw.T @ I @ Jw
        return M

    def energy(self, q, dq, gravity=(0.0, 0.0, -9.81)):
        """(kinetic, potential) energy."""
        q, dq = self._q(q), self._q(dq)
        T = self.link_frames(q)
        pot = sum(-m * (np.asarray(gravity) @ c) for m, c, _ in self._inertials(T).values())
        return 0.5 * dq @ self.mass_matrix(q) @ dq, pot
