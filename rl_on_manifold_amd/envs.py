"""Batch-1 facades with the reference's constructor signatures and numpy in / numpy out, so that existing
MushroomRL-style scripts (examples/*_exp.py: `mdp.info`, `mdp.reset()`, `mdp.step(a)`,
`mdp.get_constraints_logs()`) run unchanged on the HIP engine:

  CircleEnvAtacom        /root/reference/atacom/environments/circular_motion/circle_atacom.py:6-18
  AirHockeyPlanarAtacom  /root/reference/atacom/environments/planar_air_hockey/atacom_air_hockey.py:11-14
  AirHockeyIiwaAtacom    /root/reference/atacom/environments/iiwa_air_hockey/iiwa_hit_atacom.py:10-13

They are thin: one BatchedAtacomEnv with batch = 1 (or `n_envs` > 1 for a vectorised agent).
"""
import numpy as np
import torch

from .engine import BatchedAtacomEnv

DEFEND_START_RANGE = np.array([[0.25, 0.65], [-0.4, 0.4]])   # mushroom_rl AirHockeyDefend.start_range [upstream]
HIT_RANGE = np.array([[-0.6, -0.2], [-0.4, 0.4]])      # env_hitting.py:11


class _Facade:
    _env_name = None

    def _make(self, **kw):
        self._engine = BatchedAtacomEnv(self._env_name, batch=1, **kw)
        self.dims = self._engine.dims
        self.state = self._engine.reset()[0].cpu().numpy().astype(np.float64)

    @property
    def info(self):
        return self._engine.info

    def seed(self, seed):
        np.random.seed(seed)          # the reference's random_init draws from numpy's global generator
        self._engine.seed(seed)

    def render(self):
        pass

    def stop(self):
        self._engine.stop()

    def set_logger(self, logger):
        self._engine.set_logger(logger)

    def step(self, action):
        """atacom.py:106-115: returns (state copy, float reward, bool absorbing, {})."""
        a = np.asarray(action, dtype=np.float64).reshape(1, -1)
        obs, r, ab, _ = self._engine.step(a)
        # one device -> host transfer (and one synchronisation) per step: observation, reward and flag travel together
        host = torch.cat([obs[0], r, ab.to(obs.dtype)]).cpu().numpy().astype(np.float64)
        self.state = host[:-2].copy()
        return self.state.copy(), float(host[-2]), bool(host[-1] != 0.0), {}

    def get_constraints_logs(self):
        return self._engine.get_constraints_logs()


class CircleEnvAtacom(_Facade):
    _env_name = 'circle'

    def __init__(self, horizon=500, gamma=0.99, random_init=False, Kc=100, time_step=0.01, device='cuda:0',
                 dtype=torch.float32):
        self.random_init = random_init
        self._make(horizon=horizon, gamma=gamma, Kc=Kc, time_step=time_step, device=device, dtype=dtype)

    def reset(self, state=None):
        if state is None:
            if self.random_init:                       # circle_base.py:36-42
                y = np.random.uniform(-0.5, 1)
                x = np.sqrt(1 - y ** 2) * np.sign(np.random.uniform(-1, 1))
                dx = np.random.uniform(-1, 1)
                dy = -x * dx / y
                v = np.array([dx, dy])
                v = v / np.linalg.norm(v) * np.random.uniform(0, 1)
                state = np.array([x, y, v[0], v[1]])
            else:
                state = np.array([-1.0, 0.0, 0.0, 0.0])   # :44
        else:
            state = np.asarray(state, dtype=np.float64)
            # the reference's guard, verbatim in behaviour (circle_base.py:46-49)
            if not (abs(state[0] ** 2 + state[1] ** 2 - 1) < 1e-6
                    and abs(state[0] * state[2] - state[1] * state[3]) < 1e-6):
                raise ValueError("Can not reset to the state: ", state)
        self.state = self._engine.reset(state=state.reshape(1, 4))[0].cpu().numpy().astype(np.float64)
        return self.state


class CircleEnvErrorCorrection(CircleEnvAtacom):
    """Baseline 'E' (circle_error_correction.py:7-21): the action is the 2-d acceleration, only the error-correction
    term -Jc^+ Kc c is added.  Same constructor as the reference."""
    _env_name = 'circle_ec'

    def __init__(self, horizon=500, gamma=0.99, random_init=False, Kc=100., time_step=0.01, device='cuda:0',
                 dtype=torch.float32):
        super().__init__(horizon=horizon, gamma=gamma, random_init=random_init, Kc=Kc, time_step=time_step,
                         device=device, dtype=dtype)


class CircleEnvTerminated(CircleEnvAtacom):
    """Baseline 'T' (circle_terminated.py:8-29): unconstrained 2-d acceleration control, the episode ends with reward
    -100 as soon as a constraint value exceeds `tol`.  Same constructor as the reference."""
    _env_name = 'circle_t'

    def __init__(self, time_step=0.01, horizon=500, gamma=0.99, random_init=False, tol=0.1, device='cuda:0',
                 dtype=torch.float32):
        self.random_init = random_init
        self._make(horizon=horizon, gamma=gamma, time_step=time_step, term_tol=tol, device=device, dtype=dtype)


class _AirHockeyFacade(_Facade):
    def _init_common(self, task, gamma, horizon, timestep, n_intermediate_steps, env_noise, obs_noise, obs_delay,
                     Kc, random_init, action_penalty, device, dtype, seed=0):
        if task not in ('H', 'D'):
            raise ValueError("task must be 'H' or 'D', got %r" % (task,))
        if task == 'D' and self._env_name != 'planar':
            raise NotImplementedError       # as the reference does for the iiwa wrapper (iiwa_hit_atacom.py:20-21)
        self.task = task
        self.random_init = random_init
        # env_noise / obs_noise / obs_delay (env_base.py:176-180, env_single.py:105-107,114-117): drawn on the device from the
        # counter-based generator keyed by `seed`; mdp.seed(s) re-keys it (atacom_set_seed)
        self._make(horizon=horizon, gamma=gamma, Kc=Kc, time_step=timestep,
                   n_intermediate_steps=n_intermediate_steps, action_penalty=action_penalty, device=device,
                   dtype=dtype, task=task, env_noise=env_noise, obs_noise=obs_noise, obs_delay=obs_delay, seed=seed)
        st = self._engine.get_state()[0].cpu().numpy()
        nq = self.dims['q']
        self._init_q = st[:nq].astype(np.float64)

    def reset(self, state=None):
        nq = self.dims['q']
        if state is not None:
            state = np.asarray(state, dtype=np.float64)
        else:
            if self.task == 'D':                       # mushroom_rl AirHockeyDefend.setup [upstream, from memory]
                if self.random_init:
                    puck_pos = np.random.rand(2) * (DEFEND_START_RANGE[:, 1] - DEFEND_START_RANGE[:, 0]) \
                        + DEFEND_START_RANGE[:, 0]
                    lin_vel = np.random.uniform(1.0, 2.2)
                    angle = np.random.uniform(-0.5, 0.5)
                    puck = [puck_pos[0], puck_pos[1], 0.0, -np.cos(angle) * lin_vel, np.sin(angle) * lin_vel,
                            np.random.uniform(-1, 1)]
                else:
                    puck = [DEFEND_START_RANGE[0].mean(), 0.0, 0.0, -1.0, 0.0, 0.0]
                state = np.concatenate([self._init_q, np.zeros(nq), puck])
            else:
                if self.random_init:                   # env_hitting.py:24-25
                    puck_pos = np.random.rand(2) * (HIT_RANGE[:, 1] - HIT_RANGE[:, 0]) + HIT_RANGE[:, 0]
                else:                                  # :27
                    puck_pos = np.mean(HIT_RANGE, axis=1)
                state = np.concatenate([self._init_q, np.zeros(nq), puck_pos, np.zeros(4)])
        self.state = self._engine.reset(state=state.reshape(1, -1))[0].cpu().numpy().astype(np.float64)
        return self.state


class AirHockeyPlanarAtacom(_AirHockeyFacade):
    _env_name = 'planar'

    def __init__(self, task='H', gamma=0.99, horizon=120, timestep=1 / 240., n_intermediate_steps=4,
                 debug_gui=False, env_noise=False, obs_noise=False, obs_delay=False, Kc=240., random_init=False,
                 action_penalty=1e-3, device='cuda:0', dtype=torch.float32, seed=0):
        self._init_common(task, gamma, horizon, timestep, n_intermediate_steps, env_noise, obs_noise, obs_delay,
                          Kc, random_init, action_penalty, device, dtype, seed)


class AirHockeyIiwaAtacom(_AirHockeyFacade):
    _env_name = 'iiwa'

    def __init__(self, task='H', gamma=0.99, horizon=120, timestep=1 / 240., n_intermediate_steps=4,
                 debug_gui=False, env_noise=False, obs_noise=False, obs_delay=False, Kc=240., random_init=False,
                 action_penalty=1e-3, device='cuda:0', dtype=torch.float32, seed=0):
        self._init_common(task, gamma, horizon, timestep, n_intermediate_steps, env_noise, obs_noise, obs_delay,
                          Kc, random_init, action_penalty, device, dtype, seed)


class VectorizedAtacomEnv:
    """`n_envs` ATACOM environments behind ONE surface, for a vectorised Core: the method names and argument order of
    MushroomRL 2's `VectorizedEnvironment` (`reset_all(env_mask, state)`, `step_all(env_mask, action)`, `number`,
    `info`), torch tensors on the device in and out -- no per-step host synchronisation, unlike the batch-1 facades
    above, which exist for unchanged MushroomRL-1 scripts (SURVEY.md section 8b "batched surface").

      env = VectorizedAtacomEnv('iiwa', n_envs=8192)
      obs, _ = env.reset_all(mask_all)
      obs, reward, absorbing, info = env.step_all(mask_all, actions)     # info['last'] marks finished episodes

    Environments whose mask entry is False sit the call out ON THE DEVICE (`atacom_step_masked`): their state, step
    counter, servo joints and constraint statistics are untouched, they report their current observation, reward 0 and
    False flags.  No state save / restore, no host synchronisation: a `step_all` is one kernel launch whatever the mask,
    and is capturable in a HIP graph."""

    def __init__(self, env, n_envs, device='cuda:0', dtype=torch.float32, **engine_kwargs):
        self._engine = BatchedAtacomEnv(env, n_envs, device=device, dtype=dtype, **engine_kwargs)
        self.number = int(n_envs)
        self.dims = self._engine.dims
        self._obs = self._engine.reset()

    @property
    def info(self):
        return self._engine.info

    @property
    def engine(self):
        return self._engine

    def seed(self, seed):
        self._engine.seed(seed)

    def render_all(self, env_mask=None, record=False):
        pass

    def stop(self):
        self._engine.stop()

    def _mask(self, env_mask):
        """uint8 [n_envs] on the device, or None for "all" -- decided from the argument alone (None), never by looking at
        the mask's values (that would be a device -> host synchronisation per step)."""
        if env_mask is None:
            return None
        m = env_mask if isinstance(env_mask, torch.Tensor) else torch.as_tensor(env_mask)
        m = m.to(device=self._engine.device)
        if m.dtype == torch.bool:
            m = m.contiguous().view(torch.uint8)
        elif m.dtype != torch.uint8:
            m = (m != 0).view(torch.uint8)
        if tuple(m.shape) != (self.number,):
            raise ValueError("env_mask must have shape (%d,), got %s" % (self.number, tuple(m.shape)))
        return m.contiguous()

    def reset_all(self, env_mask=None, state=None):
        """Reset the masked environments (all if None); returns (observations [n_envs, D], {})."""
        self._obs = self._engine.reset(mask=self._mask(env_mask), state=state)
        return self._obs, {}

    def step_all(self, env_mask, action):
        """One env step of the masked environments.  action [n_envs, k] (rows of masked-out environments are ignored)."""
        obs, r, ab, info = self._engine.step(action, mask=self._mask(env_mask))
        self._obs = obs
        return obs, r, ab, info

    def get_constraints_logs(self):
        return self._engine.get_constraints_logs()
