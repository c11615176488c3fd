"""One-environment-at-a-time float64 restatement of the ATACOM step (oracle; test infrastructure).

Follows, function by function,
  /root/reference/atacom/atacom.py            (AtacomEnvWrapper: step, step_action_function, ...)
  /root/reference/atacom/constraints.py       (ViabilityConstraint / ConstraintsSet algebra)
  /root/reference/atacom/environments/circular_motion/circle_base.py + circle_atacom.py
  /root/reference/atacom/environments/planar_air_hockey/atacom_air_hockey.py:28-107
  /root/reference/atacom/environments/iiwa_air_hockey/iiwa_hit_atacom.py:23-139,
      env_base.py:155-194, env_single.py:69-120, env_hitting.py:8-85
and keeps the reference's algorithmic shape (one SVD + one RREF per env per physics sub-step), so
it also serves as the "port" CPU baseline of bench.py.

Where the reference delegates to PyBullet (rigid-body stepping, inverse dynamics) this build defines
its own dynamics -- inverse dynamics followed by forward dynamics is the identity on the controlled
joints, integrated with semi-implicit Euler at the physics rate -- see DESIGN.md "Dynamics model".
"""
from dataclasses import dataclass, field
import numpy as np

from . import robots
from .nullspace import pinv_null, rref

ENV_CIRCLE, ENV_PLANAR, ENV_IIWA = 0, 1, 2
MODE_ATACOM, MODE_ERROR_CORRECTION, MODE_TERMINATED = 0, 1, 2     # wrapper variants

# env_base.py:155-159 (iiwa) -- the MushroomRL planar env uses the same table
TABLE_LENGTH, TABLE_WIDTH, GOAL_WIDTH = 1.96, 1.02, 0.25
PUCK_RADIUS, MALLET_RADIUS, UNIVERSAL_HEIGHT = 0.03165, 0.05, 0.1505
HIT_RANGE = np.array([[-0.6, -0.2], [-0.4, 0.4]])      # env_hitting.py:11
GOAL_POS = np.array([0.98, 0.0])                        # env_hitting.py:12
DEFEND_START_RANGE = np.array([[0.25, 0.65], [-0.4, 0.4]])   # mushroom_rl AirHockeyDefend.start_range [upstream]
# contact model of this build (row N1; Bullet's is unpinned): restitution of mallet / rim contacts
E_MALLET, E_RIM = 0.8, 0.8


@dataclass
class EnvSpec:
    env_id: int
    dim_q: int
    n_f: int
    n_g: int
    K: np.ndarray            # per constraint row (f rows first): ViabilityConstraint.K
    Kc: np.ndarray           # per row, atacom.py:42-45
    vel_max: np.ndarray
    acc_max: np.ndarray
    Kq: np.ndarray
    dt: float
    substeps: int
    horizon: int
    gamma: float = 0.99
    obs_dim: int = 0
    hold_q: bool = True      # quirk Q1: q, dq frozen over the sub-steps of one env step
    bias_mode: str = 'reference'
    rref_tol: float = 0.05   # atacom.py:128
    action_penalty: float = 1e-3
    mode: int = 0            # MODE_*
    term_tol: float = 0.1    # circle_terminated.py:13
    base_xy: np.ndarray = field(default_factory=lambda: np.zeros(2))
    base_dt: float = 0.0     # time step of the BASE env when it differs from the wrapper's (quirk Q4: CircleEnvAtacom /
                             # CircleEnvErrorCorrection hand time_step to the wrapper only, circle_atacom.py:7-18 -- the
                             # base CircularMotion keeps its default 0.01); 0 = same as dt
    chart_mode: int = 0      # 0: the reference's chart (LAPACK null basis + rref with tolerance); 1: the canonical chart
                             # (oracle/canonical_chart.py; opt-in) -- batched oracle only
    task: int = 0            # planar: 0 = hitting ('H'), 1 = defending ('D', atacom_air_hockey.py:22-27 -> mushroom_rl's
                             # AirHockeyDefend [upstream, restated from memory]) -- batched oracle only
    dynamics_mode: int = 0   # 0: inverse o forward dynamics = identity; 1: rigid body (row N4, iiwa,
                             # oracle/dynamics.py -- implemented by the batched oracle only)
    # domain randomisation of the air-hockey base envs (constructor kwargs of iiwa_hit_atacom.py:11-13 /
    # atacom_air_hockey.py:12-14, all default False) -- batched oracle only:
    obs_noise: bool = False  # env_single.py:105-107: puck pose (x, y, yaw) of every observation += N(0, 0.001^2)
    obs_delay: bool = False  # env_single.py:114-117: puck and joint velocities of every observation low-passed, alpha = 0.5
    env_noise: bool = False  # env_base.py:176-180: a random planar force 0.0005 N(0, 1) on the puck in every physics sub-step
    puck_mass: float = 0.01  # kg; MushroomRL's puck.urdf [upstream, from memory -- not in the reference tree]: scales env_noise

    @property
    def dt_base(self):
        return self.base_dt if self.base_dt > 0 else self.dt

    @property
    def n_c(self):
        return self.n_f + self.n_g

    @property
    def n_null(self):
        return self.dim_q - self.n_f          # atacom.py:39

    @property
    def action_dim(self):
        return self.n_null if self.mode == MODE_ATACOM else self.dim_q       # error_correction_wrapper.py:48

    @property
    def alpha_max(self):
        return float(np.max(self.acc_max))    # atacom.py:71


def circle_spec(horizon=500, gamma=0.99, Kc=100.0, dt=0.01):
    """circle_atacom.py:7-18.  `dt` is the WRAPPER's time_step; the base CircularMotion is built without it and keeps
    its default 0.01 (quirk Q4)."""
    return EnvSpec(ENV_CIRCLE, 2, 1, 1, K=np.array([0.1, 2.0]), Kc=np.full(2, float(Kc)),
                   vel_max=np.ones(2), acc_max=np.full(2, 10.0), Kq=np.full(2, 20.0), dt=dt,
                   substeps=1, horizon=horizon, gamma=gamma, obs_dim=4, hold_q=False, base_dt=0.01)


def circle_ec_spec(horizon=500, gamma=0.99, Kc=100.0, dt=0.01):
    """CircleEnvErrorCorrection, circle_error_correction.py:7-21 (same constraints / gains, wrapper 'E')."""
    sp = circle_spec(horizon, gamma, Kc, dt)
    sp.mode = MODE_ERROR_CORRECTION
    return sp


def circle_t_spec(horizon=500, gamma=0.99, dt=0.01, tol=0.1):
    """CircleEnvTerminated, circle_terminated.py:8-29 (no wrapper at all)."""
    sp = circle_spec(horizon, gamma, 100.0, dt)
    sp.mode = MODE_TERMINATED
    sp.base_dt = dt                 # here time_step does reach the base env (circle_terminated.py:13-14)
    sp.term_tol = tol
    return sp


def planar_spec(horizon=120, gamma=0.99, Kc=240.0, dt=1 / 240.0, substeps=4, bias_mode='reference', task=0):
    """atacom_air_hockey.py:28-43 (Kq = 2 acc_max / vel_max)."""
    acc = np.full(3, 10.0)
    vel = robots.PLANAR_VEL_LIMIT.copy()
    return EnvSpec(ENV_PLANAR, 3, 0, 6, K=np.array([0.5] * 3 + [1.0] * 3), Kc=np.full(6, float(Kc)),
                   vel_max=vel, acc_max=acc, Kq=2 * acc / vel, dt=dt, substeps=substeps,
                   horizon=horizon, gamma=gamma, obs_dim=12, bias_mode=bias_mode,
                   base_xy=robots.PLANAR_BASE_XYZ[:2].copy(), task=task)


def iiwa_spec(horizon=120, gamma=0.99, Kc=240.0, dt=1 / 240.0, substeps=4, bias_mode='reference', dynamics_mode=0):
    """iiwa_hit_atacom.py:23-40 (Kq = 4 acc_max / vel_max)."""
    acc = np.full(6, 10.0)
    vel = robots.IIWA_VEL_LIMIT[:6].copy()
    return EnvSpec(ENV_IIWA, 6, 1, 11, K=np.array([0.1] + [0.5] * 5 + [1.0] * 6),
                   Kc=np.full(12, float(Kc)), vel_max=vel, acc_max=acc, Kq=4 * acc / vel, dt=dt,
                   substeps=substeps, horizon=horizon, gamma=gamma, obs_dim=18, bias_mode=bias_mode,
                   base_xy=robots.IIWA_BASE_XYZ[:2].copy(), dynamics_mode=dynamics_mode)


def make_spec(env_id, **kw):
    return {ENV_CIRCLE: circle_spec, ENV_PLANAR: planar_spec, ENV_IIWA: iiwa_spec}[env_id](**kw)


# ------------------------------------------------------------------ constraint callables (A9-A11)
def constraint_terms(spec, q, dq):
    """(fun_origin[c], J[c, q], b_state[c]) with the f rows first -- the three callables every
    ViabilityConstraint is built from (constraints.py:12-31)."""
    if spec.env_id == ENV_CIRCLE:
        # circle_atacom.py:47-70
        fun = np.array([q[0] ** 2 + q[1] ** 2 - 1.0, -q[1] - 0.5])
        J = np.array([[2 * q[0], 2 * q[1]], [0.0, -1.0]])
        b = np.array([2 * dq[0] ** 2 + 2 * dq[1] ** 2, 0.0])
        return fun, J, b
    bx = TABLE_LENGTH / 2 - MALLET_RADIUS      # 0.93
    by = TABLE_WIDTH / 2 - MALLET_RADIUS       # 0.46
    if spec.env_id == ENV_PLANAR:
        # atacom_air_hockey.py:78-107
        p, _ = robots.planar_fk(q)
        pw = p + spec.base_xy
        Je = robots.planar_jacobian(q)
        acc = robots.planar_bias(q, dq, spec.bias_mode)
        sel = np.array([[-1.0, 0.0], [0.0, -1.0], [0.0, 1.0]])          # :90,:97
        lim = robots.PLANAR_POS_LIMIT
        fun = np.concatenate([[-pw[0] - bx, -pw[1] - by, pw[1] - by], q ** 2 - lim ** 2])
        J = np.vstack([sel @ Je, 2 * np.diag(q)])
        b = np.concatenate([sel @ acc, 2 * dq ** 2])
        return fun, J, b
    if spec.env_id == ENV_IIWA:
        # iiwa_hit_atacom.py:70-139
        pe, _ = robots.iiwa_frame(q, 'ee')
        p4, _ = robots.iiwa_frame(q, 'link_4')
        p7, _ = robots.iiwa_frame(q, 'link_7')
        Je = robots.iiwa_frame_jacobian(q, 'ee')
        J4 = robots.iiwa_frame_jacobian(q, 'link_4')
        J7 = robots.iiwa_frame_jacobian(q, 'link_7')
        ae = robots.iiwa_frame_bias(q, dq, 'ee', spec.bias_mode)
        a4 = robots.iiwa_frame_bias(q, dq, 'link_4', spec.bias_mode)
        a7 = robots.iiwa_frame_bias(q, dq, 'link_7', spec.bias_mode)
        xw = pe[0] + spec.base_xy[0]
        yw = pe[1] + spec.base_xy[1]
        lim = robots.IIWA_POS_LIMIT[:6]
        fun = np.concatenate([[pe[2] - UNIVERSAL_HEIGHT],                          # :70-74
                              [-xw - bx, -yw - by, yw - by, -p4[2] + 0.36, -p7[2] + 0.25],  # :93-106
                              q ** 2 - lim ** 2])                                  # :132-133
        J = np.vstack([Je[2], -Je[0], -Je[1], Je[1], -J4[2], -J7[2], 2 * np.diag(q)])   # :76-82,:108-117,:135
        b = np.concatenate([[ae[2]], [-ae[0], -ae[1], ae[1], -a4[2], -a7[2]], 2 * dq ** 2])  # :84-91,:119-130,:138
        return fun, J, b
    raise ValueError(spec.env_id)


def mallet_xy_world(spec, q):
    """World xy of the mallet tip (LINK_POS of striker_mallet_tip, env_base.py:189-191): the
    universal joint keeps the mallet below the rod tip, so xy equals the tip frame's."""
    if spec.env_id == ENV_PLANAR:
        p, _ = robots.planar_fk(q)
        return p + spec.base_xy
    p, _ = robots.iiwa_frame(q, 'ee')
    return p[:2] + spec.base_xy


# ------------------------------------------------------------------ ATACOM core (A2-A8, A12)
def acc_truncation(spec, dq, ddq):
    """atacom.py:117-121."""
    up = np.maximum(np.minimum(spec.acc_max, -spec.Kq * (dq - spec.vel_max)), -spec.acc_max)
    lo = np.minimum(np.maximum(-spec.acc_max, -spec.Kq * (dq + spec.vel_max)), spec.acc_max)
    return np.clip(ddq, lo, up)


def slack_init(spec, q, dq):
    """atacom.py:145-149."""
    fun, J, _ = constraint_terms(spec, q, dq)
    g = (fun + spec.K * (J @ dq))[spec.n_f:]
    return np.sqrt(np.maximum(-2.0 * g, 0.0))


def tangent_space_accel(spec, q, dq, s, alpha, return_parts=False):
    """atacom.py:123-133: mu = -Jc^+ psi + Nc alpha - Jc^+ (Kc * c)  for the current (q, dq, s)."""
    nq, nf, ng, nc = spec.dim_q, spec.n_f, spec.n_g, spec.n_c
    fun, J, bst = constraint_terms(spec, q, dq)
    Jdq = J @ dq
    Jc = np.zeros((nc, nq + ng))                                  # :151-165
    Jc[:, :nq] = np.diag(spec.K) @ J       # constraints.py:39-40; the matmul also turns -0.0 into +0.0,
    #                                        which matters: LAPACK's reflectors branch on sign(alpha)
    Jc[nf:, nq:] = np.diag(s)
    psi = Jdq + spec.K * bst                                      # constraints.py:42-43
    Jc_inv, Nc = pinv_null(Jc)                                    # :127
    Nc = rref(Nc[:, :spec.n_null], row_vectors=False, tol=spec.rref_tol)   # :128
    c = fun + spec.K * Jdq                                        # constraints.py:33-37, :183-196
    c[nf:] += 0.5 * s ** 2
    act_a = -Jc_inv @ psi                                         # :130
    act_b = Nc @ alpha                                            # :131
    act_err = -Jc_inv @ (spec.Kc * c)                             # :132,:181
    mu = act_a + act_b + act_err                                  # :133
    if return_parts:
        return mu, dict(Jc=Jc, psi=psi, Jc_inv=Jc_inv, Nc=Nc, c=c, act_a=act_a, act_b=act_b,
                        act_err=act_err)
    return mu


def error_correction_accel(spec, q, dq, s, alpha):
    """error_correction_wrapper.py:117-130: mu = [alpha; 0] - Jc^+ (Kc * c); no drift term, no null space."""
    nq, nf, ng, nc = spec.dim_q, spec.n_f, spec.n_g, spec.n_c
    fun, J, _ = constraint_terms(spec, q, dq)
    Jc = np.zeros((nc, nq + ng))
    Jc[:, :nq] = np.diag(spec.K) @ J
    Jc[nf:, nq:] = np.diag(s)
    Jc_inv, _ = pinv_null(Jc)
    c = fun + spec.K * (J @ dq)
    c[nf:] += 0.5 * s ** 2
    return np.concatenate([alpha, np.zeros(ng)]) - Jc_inv @ (spec.Kc * c)


def origin_constraints(spec, q):
    """c with origin_constr=True and s = 0, |.| on the equality rows (atacom.py:201-203)."""
    fun, _, _ = constraint_terms(spec, q, np.zeros(spec.dim_q))
    c = fun.copy()
    c[:spec.n_f] = np.abs(c[:spec.n_f])
    return c


# ------------------------------------------------------------------ the environment (A1, A13-A16)
class ScalarAtacomEnv:
    """One environment; mirrors the reference's step()/reset()/get_constraints_logs() surface."""

    def __init__(self, spec, init_q=None, init_dq=None, puck=None):
        self.spec = spec
        nq = spec.dim_q
        if init_q is None:
            init_q = {ENV_CIRCLE: np.array([-1.0, 0.0]), ENV_PLANAR: robots.PLANAR_INIT_Q,
                      ENV_IIWA: np.zeros(6)}[spec.env_id]
        self.init_q = np.array(init_q, dtype=np.float64)
        self.init_dq = np.zeros(nq) if init_dq is None else np.array(init_dq, dtype=np.float64)
        # puck (x, y, yaw, vx, vy, wz) in world axes; default = centre of the hit range (env_hitting.py:27)
        self.init_puck = np.array([HIT_RANGE[0].mean(), HIT_RANGE[1].mean(), 0, 0, 0, 0.0]) \
            if puck is None else np.array(puck, dtype=np.float64)
        self.logs = []
        self.reset()

    # -- reset (A16, A12)
    def reset(self, q=None, dq=None, puck=None):
        sp = self.spec
        self.q = (self.init_q if q is None else np.array(q, dtype=np.float64)).copy()
        self.dq = (self.init_dq if dq is None else np.array(dq, dtype=np.float64)).copy()
        if sp.env_id == ENV_CIRCLE and q is not None:
            # circle_base.py:46-49
            if not (abs(self.q[0] ** 2 + self.q[1] ** 2 - 1) < 1e-6
                    and abs(self.q[0] * self.dq[0] - self.q[1] * self.dq[1]) < 1e-6):
                raise ValueError("Can not reset to the state: ", np.concatenate([self.q, self.dq]))
        self.puck = (self.init_puck if puck is None else np.array(puck, dtype=np.float64)).copy()
        self.has_hit, self.r_hit, self.vel_hit_x = False, 0.0, 0.0      # env_hitting.py:35-37
        self.t = 0
        self.s = slack_init(sp, self.q, self.dq)
        return self.observation()

    def observation(self):
        sp = self.spec
        if sp.env_id == ENV_CIRCLE:
            return np.concatenate([self.q, self.dq])                    # circle_base.py:83-84
        pk = self.puck
        # env_single.py:82-120: puck pose / velocity in the robot frame (pure translation), q, dq
        return np.concatenate([[pk[0] - sp.base_xy[0], pk[1] - sp.base_xy[1], pk[2]], pk[3:6],
                               self.q, self.dq])

    # -- one env step (A1, A2, A14, A15)
    def step(self, action, return_debug=False):
        sp = self.spec
        nq = sp.dim_q
        act = np.clip(np.asarray(action, dtype=np.float64), -1.0, 1.0)
        if sp.mode == MODE_ATACOM:
            alpha = act * sp.alpha_max                                                    # atacom.py:107-108
        elif sp.mode == MODE_ERROR_CORRECTION:
            alpha = act * sp.acc_max                                                      # error_correction_wrapper.py:106-107
        else:
            alpha = act
        dbg = []
        if sp.env_id == ENV_CIRCLE and sp.mode == MODE_TERMINATED:
            # circle_terminated.py:17-29 over circle_base.py:53-67
            c_pre = np.array([abs(self.q[0] ** 2 + self.q[1] ** 2 - 1), -self.q[1] - 0.5,
                              abs(self.dq[0]) - 1, abs(self.dq[1]) - 1])
            self.logs.append(c_pre)
            a = alpha * 10.0
            self.q = self.q + (self.dq * sp.dt_base + a * sp.dt_base ** 2 / 2)
            self.dq = self.dq + a * sp.dt_base
            reward = float(np.exp(-np.linalg.norm(np.array([1.0, 0.0]) - self.q)))
            absorbing = bool(np.any(c_pre > sp.term_tol))
            if absorbing:
                reward = -100.0
        elif sp.env_id == ENV_CIRCLE:
            # circle_base.py:53-67 with the ATACOM callback of atacom.py:123-139
            c_pre = np.array([abs(self.q[0] ** 2 + self.q[1] ** 2 - 1), -self.q[1] - 0.5,
                              abs(self.dq[0]) - 1, abs(self.dq[1]) - 1])            # :86-107
            self.logs.append(c_pre)
            if sp.mode == MODE_ERROR_CORRECTION:
                mu = error_correction_accel(sp, self.q, self.dq, self.s, alpha)
            else:
                mu = tangent_space_accel(sp, self.q, self.dq, self.s, alpha)
            self.s = self.s + mu[nq:] * sp.dt                                       # atacom.py:135
            ddq = acc_truncation(sp, self.dq, mu[:nq])                              # :137
            ctrl = ddq / sp.acc_max                                                 # circle_atacom.py:26-27
            a = np.clip(ctrl, -1.0, 1.0) * 10.0                                     # circle_base.py:59-60
            self.q = self.q + (self.dq * sp.dt_base + a * sp.dt_base ** 2 / 2)                  # :62
            self.dq = self.dq + a * sp.dt_base                                           # :63
            reward = float(np.exp(-np.linalg.norm(np.array([1.0, 0.0]) - self.q)))  # :65
            absorbing = False
            dbg.append(mu)
        else:
            q_ctl, dq_ctl = self.q.copy(), self.dq.copy()       # held copies (quirk Q1)
            q_sim, dq_sim = self.q.copy(), self.dq.copy()
            m0 = mallet_xy_world(sp, self.q)                     # mallet at the start of the env step
            for _ in range(sp.substeps):
                if not sp.hold_q:
                    q_ctl, dq_ctl = q_sim.copy(), dq_sim.copy()
                mu = tangent_space_accel(sp, q_ctl, dq_ctl, self.s, alpha)
                self.s = self.s + mu[nq:] * sp.dt
                ddq = acc_truncation(sp, dq_ctl, mu[:nq])
                # own dynamics: ID o FD = identity, semi-implicit Euler, Bullet's maxJointVelocity
                # clamp at 1.5 x the URDF limit (iiwa_hit_atacom.py:48-50, atacom_air_hockey.py:49-54)
                dq_sim = np.clip(dq_sim + ddq * sp.dt, -1.5 * sp.vel_max, 1.5 * sp.vel_max)
                q_sim = q_sim + dq_sim * sp.dt
                dbg.append(mu)
            self.q, self.dq = q_sim, dq_sim                      # atacom.py:111-112
            # puck: the arm is kinematic w.r.t. the puck (no reaction), so the sub-steps of the puck run after
            # the arm's, against a mallet moving uniformly from m0 to m1 over the env step
            m1 = mallet_xy_world(sp, self.q)
            for k in range(sp.substeps):
                self._puck_substep(m0 + (m1 - m0) * ((k + 1) / sp.substeps), (m1 - m0) / (sp.substeps * sp.dt))
            absorbing = self._is_absorbing()
            reward = self._reward(alpha, absorbing)
            c_i = origin_constraints(sp, self.q)                 # atacom.py:201-205
            self.logs.append(np.array([np.max(c_i), np.max(np.abs(self.dq) - sp.vel_max)]))
        self.t += 1
        out = (self.observation(), reward, absorbing, {})
        return out + (dbg,) if return_debug else out

    # -- puck: 2-D disc on a frictionless table, kinematic mallet, elastic rims with a goal mouth (row N1).
    #    PyBullet's contact solver is unpinned; this is the model of this build.
    def _puck_substep(self, mallet, mallet_vel):
        sp = self.spec
        pk = self.puck
        pk[0:3] = pk[0:3] + pk[3:6] * sp.dt
        # mallet contact: impulse along the centre line if approaching, then push the puck out of the overlap
        d = pk[0:2] - mallet
        dist = np.hypot(d[0], d[1])
        R = PUCK_RADIUS + MALLET_RADIUS
        if dist < R:
            n = d / dist if dist > 0 else np.array([1.0, 0.0])
            vrel = (pk[3:5] - mallet_vel) @ n
            if vrel < 0:
                pk[3:5] = pk[3:5] - (1 + E_MALLET) * vrel * n
            pk[0:2] = mallet + n * R
        # side rims (y)
        ylim = TABLE_WIDTH / 2 - PUCK_RADIUS
        if abs(pk[1]) > ylim:
            sgn = np.sign(pk[1])
            pk[1] = sgn * (2 * ylim - abs(pk[1]))
            if pk[4] * sgn > 0:
                pk[4] = -E_RIM * pk[4]
        # end rims (x) except in the goal mouth |y| < goal half-width (env_hitting.py:44-45)
        xlim = TABLE_LENGTH / 2 - PUCK_RADIUS
        if abs(pk[0]) > xlim and abs(pk[1]) >= GOAL_WIDTH:
            sgn = np.sign(pk[0])
            pk[0] = sgn * (2 * xlim - abs(pk[0]))
            if pk[3] * sgn > 0:
                pk[3] = -E_RIM * pk[3]
        if not self.has_hit:                                     # env_hitting.py:80-85
            v = np.hypot(pk[3], pk[4])
            if v > 0.1:
                self.has_hit = True
                self.vel_hit_x = pk[3]

    def _is_absorbing(self):
        # env_base.py:182-194 + env_hitting.py:71-78
        sp = self.spec
        bnd = np.array([TABLE_LENGTH, TABLE_WIDTH]) / 2
        if np.any(np.abs(self.puck[:2]) > bnd):
            return True
        if np.any(np.abs(mallet_xy_world(sp, self.q)) - bnd > 0.02):
            return True
        if self.has_hit and np.hypot(self.puck[3], self.puck[4]) < 0.01:
            return True
        return False

    def _reward(self, alpha, absorbing):
        # env_hitting.py:39-69
        sp = self.spec
        r = 0.0
        pp = self.puck[:2]
        if absorbing:
            if pp[0] - TABLE_LENGTH / 2 > 0 and abs(pp[1]) - GOAL_WIDTH < 0:
                r = 80.0
        elif not self.has_hit:
            ee = mallet_xy_world(sp, self.q)
            dist = np.linalg.norm(pp - ee)
            v1 = (pp - ee) / dist
            v2 = (GOAL_POS - pp) / np.linalg.norm(GOAL_POS - pp)
            r = np.exp(-8 * (dist - 0.08)) * np.clip(v2 @ v1, 0, 1)
            self.r_hit = r
        else:
            r = 1 + self.r_hit + self.vel_hit_x * 0.1
        return float(r - sp.action_penalty * np.linalg.norm(alpha))

    # -- constraint statistics (A13)
    def get_constraints_logs(self):
        logs = np.array(self.logs)
        if self.spec.env_id == ENV_CIRCLE:        # circle_base.py:109-115
            out = (np.mean(np.max(logs[:, :2], axis=1)), np.max(logs[:, :2]), np.max(logs[:, 2:]))
        else:                                      # atacom.py:207-216
            out = (np.mean(logs[:, 0]), np.max(logs[:, 0]), np.max(logs[:, 1]))
        self.logs.clear()
        return tuple(float(x) for x in out)
