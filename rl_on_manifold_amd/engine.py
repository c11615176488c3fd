"""Batched ATACOM environments on one MI355X: the host-side mirror of the reference's
`AtacomEnvWrapper` surface (/root/reference/atacom/atacom.py:90-115,141-143,207-216) with a leading
batch dimension.  All arithmetic happens in libatacom_hip.so (hand-written HIP, gfx950); this file
only moves pointers: torch ROCm tensors supply device memory and the current HIP stream.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .spaces import Box, MDPInfo

_ENV_IDS = {'circle': _lib.ENV_CIRCLE, 'A': _lib.ENV_CIRCLE, 'planar': _lib.ENV_PLANAR, 'H': _lib.ENV_PLANAR,
            'iiwa': _lib.ENV_IIWA, '7H': _lib.ENV_IIWA, 'circle_ec': _lib.ENV_CIRCLE_EC, 'E': _lib.ENV_CIRCLE_EC,
            'circle_t': _lib.ENV_CIRCLE_T, 'T': _lib.ENV_CIRCLE_T}


def _ptr(t):
    # a plain int (or None) converts to void* through the argtypes of _lib.py; no c_void_p object per call
    return t.data_ptr() if t is not None else None


# torch.cuda.current_stream(device).cuda_stream builds two Python objects per call (~2 us -- more than the circle
# kernel runs); the raw accessor PyTorch keeps for extension launchers returns the hipStream_t as an int directly
_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


class BatchedAtacomEnv:
    """`batch` independent ATACOM environments stepped by one HIP kernel launch per call.

    Same method names as the reference wrapper, batched:
      reset(mask=None, state=None) -> obs[B, D]
      step(actions[B, k])          -> obs[B, D], reward[B], absorbing[B] (bool), {'last': bool[B]}
      rollout(actions[T, B, k])    -> dict of time-major tensors, T steps in ONE launch
      get_constraints_logs()       -> (c_avg, c_max, c_dq_max), clears the log   (atacom.py:207-216)
    Tensors are torch tensors on `device` (zero-copy); numpy inputs are accepted and copied.
    """

    _warned_rigid_body = False

    def __init__(self, env, batch, device='cuda:0', dtype=torch.float32, horizon=None, gamma=None, Kc=None,
                 time_step=None, n_intermediate_steps=None, action_penalty=None, auto_reset=False,
                 hold_q=None, bias_mode='reference', rref_tol=None, lanes_per_env=0, term_tol=None, random_init=False, seed=0,
                 dynamics_mode='kinematic', chart_mode='reference', task='H', obs_noise=False, obs_delay=False,
                 env_noise=False, puck_mass=None):
        lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.AtacomError("BatchedAtacomEnv needs a ROCm GPU (torch.cuda.is_available() is False); "
                                   "there is no CPU fallback")
        self.env_id = _ENV_IDS[env] if isinstance(env, str) else int(env)
        dev = torch.device(device)
        if dev.type != 'cuda':
            raise _lib.AtacomError("device must be a ROCm GPU ('cuda:N')")
        # normalised once: torch.device('cuda') != torch.device('cuda:0'), and every comparison below is on self.device
        self._dev_index = dev.index if dev.index is not None else torch.cuda.current_device()
        self.device = torch.device('cuda', self._dev_index)
        self.dtype = dtype
        cfg = _lib.default_config(self.env_id)
        cfg.batch = int(batch)
        cfg.dtype = {torch.float32: _lib.F32, torch.float64: _lib.F64}[dtype]
        if horizon is not None:
            cfg.horizon = int(horizon)
        if gamma is not None:
            cfg.gamma = float(gamma)
        if time_step is not None:
            cfg.dt = float(time_step)
            if self.env_id == _lib.ENV_CIRCLE_T:
                cfg.dt_base = float(time_step)      # CircularMotion itself takes time_step (circle_terminated.py:13-14);
            # CircleEnvAtacom / CircleEnvErrorCorrection keep the base env at its default 0.01 (quirk Q4, dt_base untouched)
        if n_intermediate_steps is not None:
            cfg.substeps = int(n_intermediate_steps)
        if action_penalty is not None:
            cfg.action_penalty = float(action_penalty)
        if rref_tol is not None:
            cfg.rref_tol = float(rref_tol)
        if hold_q is not None:
            cfg.hold_q = int(bool(hold_q))
        cfg.bias_mode = {'reference': 0, 'exact': 1}[bias_mode]
        cfg.auto_reset = int(bool(auto_reset))
        cfg.lanes_per_env = int(lanes_per_env)      # 0 auto; 1, 2, 4 lanes per env (lane / pair / quad kernels)
        if term_tol is not None:
            cfg.term_tol = float(term_tol)
        cfg.random_init = int(bool(random_init))     # device-side random reset (circle_base.py:36-42, env_hitting.py:24-25)
        cfg.seed = int(seed) & 0x7fffffff
        # 'kinematic': q'' integrates directly (default); 'rigid_body' (iiwa, row N4): inverse dynamics -> torque ->
        # forward dynamics with the URDF inertias / damping and the joint-7 / universal-joint servos;
        # 'rigid_body_ff': the same with the servo joints' reaction fed forward into the inverse dynamics
        modes = {'kinematic': 0, 'rigid_body': 1, 'rigid_body_ff': 2}
        if dynamics_mode not in modes:
            raise ValueError("dynamics_mode must be one of %s, got %r" % (sorted(modes), dynamics_mode))
        cfg.dynamics_mode = modes[dynamics_mode]
        if dynamics_mode == 'rigid_body' and not BatchedAtacomEnv._warned_rigid_body:
            # steer users to the mode that keeps ATACOM's velocity guarantee
            BatchedAtacomEnv._warned_rigid_body = True
            import warnings
            warnings.warn("dynamics_mode='rigid_body' reproduces the reference's torque path literally (inverse dynamics with "
                          "ZERO accelerations for joint 7 and the striker's universal joint, iiwa_hit_atacom.py:58-63): near "
                          "q6 = 0 the joint-7 set-point of env_single.py:137-170 chatters and the servo's reaction drives joint 5 "
                          "past its velocity limit (c_dq_max up to +1.1, the 1.5 v_max clamp).  Use "
                          "dynamics_mode='rigid_body_ff' (servo accelerations fed forward: c_dq_max <= 0) unless that regime is "
                          "what you want to study.", stacklevel=2)
        # 'reference': LAPACK's null basis + rref(tol = 0.05), the reference's chart decision by decision (default);
        # 'canonical' (opt-in): basis-independent chart, exact null basis on every state, ~3x cheaper
        charts = {'reference': 0, 'canonical': 1}
        if chart_mode not in charts:
            raise ValueError("chart_mode must be one of %s, got %r" % (sorted(charts), chart_mode))
        cfg.chart_mode = charts[chart_mode]
        # planar only: 'H' hitting (default), 'D' defending (atacom_air_hockey.py:14-27); the reference's iiwa wrapper has no 'D'
        tasks = {'H': 0, 'D': 1}
        if task not in tasks:
            raise ValueError("task must be 'H' or 'D', got %r" % (task,))
        if task == 'D' and self.env_id != _lib.ENV_PLANAR:
            raise NotImplementedError("task 'D' exists for the planar environment only (iiwa_hit_atacom.py:20-21 raises too)")
        cfg.task = tasks[task]
        # domain randomisation of the air-hockey base envs (iiwa_hit_atacom.py:11-13, atacom_air_hockey.py:12-14): noise on
        # the observed puck pose, a low-pass on the observed velocities, a random force on the puck -- drawn on the device
        # from the counter-based generator keyed by `seed` (include/atacom_hip.h)
        cfg.obs_noise, cfg.obs_delay, cfg.env_noise = int(bool(obs_noise)), int(bool(obs_delay)), int(bool(env_noise))
        if puck_mass is not None:
            cfg.puck_mass = float(puck_mass)
        d = _lib.get_dims(self.env_id)
        self.dims = {'q': d.dim_q, 'f': d.n_f, 'g': d.n_g, 'null': d.n_null, 'c': d.n_f + d.n_g}   # atacom.py:25-40
        self.obs_dim, self.state_dim, self.init_state_dim = d.obs_dim, d.state_dim, d.init_state_dim
        self.record_dim = d.record_dim
        # get_dims reports the ACTION dimension in n_null: dim_q - n_f for ATACOM (atacom.py:39,51), dim_q for the
        # 'E' / 'T' baselines (error_correction_wrapper.py:48, circle_base.py:24)
        if Kc is not None:
            kc = np.broadcast_to(np.asarray(Kc, dtype=np.float64), (self.dims['c'],))
            for i in range(self.dims['c']):
                cfg.Kc[i] = float(kc[i])
        self.cfg = cfg
        self.batch = int(batch)
        h = C.c_void_p()
        _lib.check(lib.atacom_create(C.byref(cfg), self._dev_index, C.byref(h)))
        self._h = h
        self._lib = lib
        self._io_ok = set()
        B, k, D = self.batch, self.dims['null'], self.obs_dim
        self._obs = torch.empty((B, D), device=self.device, dtype=dtype)
        self._reward = torch.empty((B,), device=self.device, dtype=dtype)
        self._absorbing = torch.empty((B,), device=self.device, dtype=torch.uint8)
        self._last = torch.empty((B,), device=self.device, dtype=torch.uint8)
        lo, hi = self.observation_bounds(self.env_id, cfg)
        self._mdp_info = MDPInfo(Box(lo, hi), Box(-np.ones(k), np.ones(k)), cfg.gamma, cfg.horizon)   # atacom.py:50-51
        self.reset()

    @staticmethod
    def observation_bounds(env_id, cfg):
        """Bounds of the observation space exactly as the reference's MDPInfo carries them, so that a caller's
        `MinMaxPreprocessor(mdp_info=mdp.info)` (examples/iiwa_air_hockey_exp.py:32) normalises the same entries:
          circle : Box(-inf, inf) on all four entries (circle_base.py:21-22);
          iiwa   : env_single.py:69-80 -- puck x, y, yaw in [-1, 1] x [-0.5, 0.5] x [-pi, pi], puck velocities unbounded
                   (a PyBullet body velocity has no limit), joint positions within the URDF position limits, joint
                   velocities within the URDF velocity limits (iiwa_1.urdf:74,112,149,186,223,260);
          planar : the same construction with the 3R arm's limits (the reference takes these from MushroomRL's
                   AirHockeyHit, which is not in the tree: an analogue, not a pinned copy)."""
        if env_id in (_lib.ENV_CIRCLE, _lib.ENV_CIRCLE_EC, _lib.ENV_CIRCLE_T):
            inf = np.full(4, np.inf)
            return -inf, inf
        nq = 3 if env_id == _lib.ENV_PLANAR else 6
        pos = np.array(list(cfg.pos_limit)[:nq])
        vel = np.array(list(cfg.vel_max)[:nq])
        hi = np.concatenate([[1.0, 0.5, np.pi], np.full(3, np.inf), pos, vel])
        return -hi, hi

    # ------------------------------------------------------------------ reference surface
    @property
    def info(self):
        return self._mdp_info

    def seed(self, seed):
        """env.seed (atacom.py:90-91): re-keys the device-side generator behind random_init / obs_noise / env_noise
        (atacom_set_seed) from the next call on -- two experiments that call mdp.seed(s) with different s no longer share
        one noise stream.  The rest of the device path draws no random numbers.  Steps already captured in a HIP graph
        (GraphedRollout) keep the seed they were captured with."""
        self._seed = int(seed) & 0x7fffffff
        _lib.check(self._lib.atacom_set_seed(self._h, self._seed))

    def render(self):
        pass

    def stop(self):
        pass

    def set_logger(self, logger):
        self._logger = logger

    def _stream(self):
        if _raw_stream is not None:
            return _raw_stream(self._dev_index)
        return torch.cuda.current_stream(self.device).cuda_stream

    def _as_dev(self, x, shape, dtype=None):
        t = torch.as_tensor(x, dtype=dtype or self.dtype, device=self.device)
        if tuple(t.shape) != tuple(shape):
            raise ValueError("expected shape %s, got %s" % (tuple(shape), tuple(t.shape)))
        return t.contiguous()

    def reset(self, mask=None, state=None):
        """Reset the masked envs (all if mask is None).  `state` ([B, init_state_dim] = [q, dq(, puck6)]) replaces
        their stored initial state first.  Returns the observation of every env."""
        m = None if mask is None else self._as_dev(mask, (self.batch,), torch.uint8)
        s = None if state is None else self._as_dev(state, (self.batch, self.init_state_dim))
        obs = torch.empty((self.batch, self.obs_dim), device=self.device, dtype=self.dtype)     # fresh: the caller keeps it
        _lib.check(self._lib.atacom_reset(self._h, _ptr(m), _ptr(s), _ptr(obs), self._stream()))
        return obs

    def step(self, actions, mask=None):
        """mask (optional, [B] bool / uint8): environments with a zero entry sit the call out ON THE DEVICE -- state, step
        counter and statistics untouched, obs = their current observation, reward 0, flags False."""
        a = self._as_dev(actions, (self.batch, self.dims['null']))
        # fresh output tensors written by the kernel itself (the reference returns copies, atacom.py:115); the flags are
        # 0 / 1 bytes, so the bool tensors are reinterpreting views -- no copy or conversion kernel follows the step
        B = self.batch
        obs = torch.empty((B, self.obs_dim), device=self.device, dtype=self.dtype)
        reward = torch.empty((B,), device=self.device, dtype=self.dtype)
        absorbing = torch.empty((B,), device=self.device, dtype=torch.uint8)
        last = torch.empty((B,), device=self.device, dtype=torch.uint8)
        if mask is None:
            _lib.check(self._lib.atacom_step(self._h, _ptr(a), _ptr(obs), _ptr(reward), _ptr(absorbing), _ptr(last),
                                              self._stream()))
        else:
            m = mask.view(torch.uint8) if (isinstance(mask, torch.Tensor) and mask.dtype == torch.bool
                                           and self._on_my_device(mask) and mask.is_contiguous()) \
                else self._as_dev(mask, (B,), torch.uint8)
            if tuple(m.shape) != (B,):
                raise ValueError("expected a mask of shape (%d,), got %s" % (B, tuple(m.shape)))
            _lib.check(self._lib.atacom_step_masked(self._h, _ptr(m), _ptr(a), _ptr(obs), _ptr(reward), _ptr(absorbing),
                                                     _ptr(last), self._stream()))
        return obs, reward, absorbing.view(torch.bool), {'last': last.view(torch.bool)}

    def _check_io(self, t, shape, dtype, what):
        """Raw-pointer entry points read whatever they are given: a float64, strided or host tensor would be read as
        garbage.  Validated once per distinct tensor (keyed by storage, shape, dtype), so the steady state of a loop that
        reuses its buffers pays one dict lookup per argument and allocates nothing."""
        if t is None:
            return
        if not isinstance(t, torch.Tensor) or not self._on_my_device(t):
            raise ValueError("%s must be a torch tensor on %s" % (what, self.device))
        key = (what, t.data_ptr(), t.dtype, tuple(t.shape), t.stride())
        if key in self._io_ok:
            return
        if t.dtype != dtype or tuple(t.shape) != tuple(shape) or not t.is_contiguous():
            raise ValueError("%s must be a contiguous %s tensor of shape %s (got %s, %s%s)"
                             % (what, dtype, tuple(shape), t.dtype, tuple(t.shape), '' if t.is_contiguous() else ', strided'))
        if len(self._io_ok) > 4096:
            self._io_ok.clear()
        self._io_ok.add(key)

    def step_into(self, actions, obs, reward, absorbing, last=None, mask=None):
        """Allocation-free variant of step(): caller-owned output tensors (uint8 for the flags).  `mask` (uint8 [B],
        optional): environments with a zero byte sit the call out on the device (atacom_step_masked)."""
        B = self.batch
        self._check_io(actions, (B, self.dims['null']), self.dtype, 'actions')
        self._check_io(obs, (B, self.obs_dim), self.dtype, 'obs')
        self._check_io(reward, (B,), self.dtype, 'reward')
        self._check_io(absorbing, (B,), torch.uint8, 'absorbing')
        self._check_io(last, (B,), torch.uint8, 'last')
        if mask is None:
            _lib.check(self._lib.atacom_step(self._h, _ptr(actions), _ptr(obs), _ptr(reward), _ptr(absorbing),
                                              _ptr(last), self._stream()))
        else:
            self._check_io(mask, (B,), torch.uint8, 'mask')
            _lib.check(self._lib.atacom_step_masked(self._h, _ptr(mask), _ptr(actions), _ptr(obs), _ptr(reward),
                                                     _ptr(absorbing), _ptr(last), self._stream()))

    def bind_step(self, actions, obs, reward, absorbing, last=None, mask=None):
        """step_into() with its arguments validated ONCE: returns a zero-argument callable that launches atacom_step (or
        atacom_step_masked) on the caller's current stream with these tensors -- for loops that reuse their buffers and are
        bound by the host (CircularMotion: the kernel runs 4 us, five argument checks and the pointer look-ups of
        step_into cost 3.5 us per call on top of the 5.3 us of the launch itself; profiles/tools/gpu_hostpath_probe.py).  The
        callable keeps the tensors alive; it must not be used after close()."""
        B = self.batch
        self._check_io(actions, (B, self.dims['null']), self.dtype, 'actions')
        self._check_io(obs, (B, self.obs_dim), self.dtype, 'obs')
        self._check_io(reward, (B,), self.dtype, 'reward')
        self._check_io(absorbing, (B,), torch.uint8, 'absorbing')
        self._check_io(last, (B,), torch.uint8, 'last')
        self._check_io(mask, (B,), torch.uint8, 'mask')
        keep = (actions, obs, reward, absorbing, last, mask)
        pa, po, pr, pb, pl, pm = [_ptr(t) for t in keep]
        check, stream = _lib.check, self._stream
        if mask is None:
            fn = self._lib.atacom_step

            def call(_keep=keep):
                h = self._h
                if not h:
                    raise RuntimeError('bind_step: the engine is closed')
                check(fn(h, pa, po, pr, pb, pl, stream()))
        else:
            fn = self._lib.atacom_step_masked

            def call(_keep=keep):
                h = self._h
                if not h:
                    raise RuntimeError('bind_step: the engine is closed')
                check(fn(h, pm, pa, po, pr, pb, pl, stream()))
        return call

    def rollout(self, actions, want_next_obs=True, out=None):
        """T env steps in one kernel launch.  actions [T, B, k] -> dict(obs, next_obs, reward, absorbing, last)."""
        T = int(actions.shape[0])
        a = self._as_dev(actions, (T, self.batch, self.dims['null']))
        B, D = self.batch, self.obs_dim
        if out is None:
            out = {'obs': torch.empty((T, B, D), device=self.device, dtype=self.dtype),
                   'next_obs': torch.empty((T, B, D), device=self.device, dtype=self.dtype) if want_next_obs else None,
                   'reward': torch.empty((T, B), device=self.device, dtype=self.dtype),
                   'absorbing': torch.empty((T, B), device=self.device, dtype=torch.uint8),
                   'last': torch.empty((T, B), device=self.device, dtype=torch.uint8)}
        _lib.check(self._lib.atacom_rollout(self._h, T, _ptr(a), _ptr(out['obs']), _ptr(out.get('next_obs')),
                                             _ptr(out['reward']), _ptr(out['absorbing']), _ptr(out['last']),
                                             self._stream()))
        out['action'] = a
        return out

    def rollout_policy(self, policy, n_steps, noise=None, want_next_obs=True):
        """T env steps with the actor MLP evaluated inside the kernel (row N2): one launch for the whole collection
        phase.  `policy` is an MlpPolicy (see below); `noise` [T, B, k] standard-normal draws supplied by the caller
        (None = deterministic mean action).  Returns the same dict as rollout(), 'action' being what the policy drew."""
        T, B, D, k = int(n_steps), self.batch, self.obs_dim, self.dims['null']
        net = policy.as_struct(self)
        nz = None if noise is None else self._as_dev(noise, (T, B, k))
        out = {'obs': torch.empty((T, B, D), device=self.device, dtype=self.dtype),
               'next_obs': torch.empty((T, B, D), device=self.device, dtype=self.dtype) if want_next_obs else None,
               'action': torch.empty((T, B, k), device=self.device, dtype=self.dtype),
               'reward': torch.empty((T, B), device=self.device, dtype=self.dtype),
               'absorbing': torch.empty((T, B), device=self.device, dtype=torch.uint8),
               'last': torch.empty((T, B), device=self.device, dtype=torch.uint8)}
        _lib.check(self._lib.atacom_rollout_mlp(self._h, T, C.byref(net), _ptr(nz), _ptr(out['obs']),
                                                 _ptr(out['next_obs']), _ptr(out['action']), _ptr(out['reward']),
                                                 _ptr(out['absorbing']), _ptr(out['last']), self._stream()))
        return out

    def rollout_packed(self, actions=None, policy=None, n_steps=None, noise=None, out=None, batch_stride=None):
        """T env steps in one launch, written as ONE packed float record per (step, env):
        records [T, batch_stride, record_dim] = [obs | action | reward | next_obs | absorbing | last] -- the layout the
        sharded collector all-gathers as it is (rollout.py).  Either `actions` [T, B, k] or `policy` (an MlpPolicy,
        evaluated inside the kernel; `noise` [T, B, k] or None) with `n_steps`.  batch_stride > batch pads the env axis
        (ragged shards); the padding rows are zero (filled at allocation, or here when the caller supplies `out`) and never
        written by the kernel."""
        B, k, F = self.batch, self.dims['null'], self.record_dim
        if (actions is None) == (policy is None):
            raise ValueError("give either actions or policy")
        T = int(actions.shape[0]) if actions is not None else int(n_steps)
        ld = B if batch_stride is None else int(batch_stride)
        if out is None:
            alloc = torch.empty if ld == B else torch.zeros
            out = alloc((T, ld, F), device=self.device, dtype=self.dtype)
        elif tuple(out.shape) != (T, ld, F) or not out.is_contiguous() or out.dtype != self.dtype \
                or not self._on_my_device(out):
            raise ValueError("out must be a contiguous [%d, %d, %d] tensor of the engine's dtype on %s" % (T, ld, F, self.device))
        elif ld > B:
            out[:, B:].zero_()                  # a caller's buffer may hold anything: the padding rows are zero (rollout.py)
        if actions is not None:
            a = self._as_dev(actions, (T, B, k))
            _lib.check(self._lib.atacom_rollout_packed(self._h, T, _ptr(a), None, None, _ptr(out), ld, self._stream()))
        else:
            net = policy.as_struct(self)
            nz = None if noise is None else self._as_dev(noise, (T, B, k))
            _lib.check(self._lib.atacom_rollout_packed(self._h, T, None, C.byref(net), _ptr(nz), _ptr(out), ld,
                                                        self._stream()))
        return out

    def unpack_records(self, rec):
        """Views into packed records [..., record_dim] (no copy)."""
        D, k = self.obs_dim, self.dims['null']
        return {'obs': rec[..., :D], 'action': rec[..., D:D + k], 'reward': rec[..., D + k],
                'next_obs': rec[..., D + k + 1:2 * D + k + 1], 'absorbing': rec[..., 2 * D + k + 1] > 0.5,
                'last': rec[..., 2 * D + k + 2] > 0.5}

    def _lanes(self):
        a, b = C.c_int32(0), C.c_int32(0)
        _lib.check(self._lib.atacom_get_lanes(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    @property
    def lanes_per_env(self):
        """The kernel mapping step() runs: lanes per environment (what the library chose when 0 was requested)."""
        return self._lanes()[0]

    @property
    def rollout_lanes_per_env(self):
        """The kernel mapping of rollout() / rollout_packed(actions=...) (may differ from step()'s; the policy kernel's:
        policy_lanes_per_env)."""
        return self._lanes()[1]

    @property
    def policy_lanes_per_env(self):
        """The kernel mapping of rollout_policy() / rollout_packed(policy=...): the T-step mapping where the policy kernel has
        that form (float64: quad or lane; atacom_get_policy_lanes)."""
        a = C.c_int32(0)
        _lib.check(self._lib.atacom_get_policy_lanes(self._h, C.byref(a)))
        return int(a.value)

    def _on_my_device(self, t):
        return t.device.type == 'cuda' and t.device.index == self._dev_index

    def snapshot(self, out=None):
        """Checkpoint of the WHOLE persistent state (atacom_snapshot_save): what get_state() returns plus the stored initial
        states, the constraint-statistics accumulators, the episode counters of the device-side random reset and the servo
        joints -- an opaque uint8 tensor on the device; `restore(image)` followed by the same calls reproduces the run bit
        for bit.  Two device-to-device copies on the current stream, no synchronisation."""
        n = int(self._lib.atacom_snapshot_bytes(self._h))
        if n < 0:
            _lib.check(n)
        if out is None:
            out = torch.empty((n,), device=self.device, dtype=torch.uint8)
        elif out.dtype != torch.uint8 or out.numel() < n or not self._on_my_device(out) or not out.is_contiguous():
            raise ValueError("snapshot buffer must be a contiguous uint8 tensor of >= %d bytes on %s" % (n, self.device))
        _lib.check(self._lib.atacom_snapshot_save(self._h, _ptr(out), self._stream()))
        return out

    def restore(self, image):
        """Reads the image's 64-byte header back first (synchronises the current stream) and raises ValueError for an
        image taken from another handle shape (environment, task, dtype, batch)."""
        n = int(self._lib.atacom_snapshot_bytes(self._h))
        if image.dtype != torch.uint8 or image.numel() < n or not self._on_my_device(image) or not image.is_contiguous():
            raise ValueError("snapshot image must be a contiguous uint8 tensor of >= %d bytes on %s" % (n, self.device))
        rc = self._lib.atacom_snapshot_restore(self._h, _ptr(image), self._stream())
        if rc == _lib.E_INVALID:                      # a bad image is the caller's error ...
            raise ValueError(self._lib.atacom_last_error().decode())
        _lib.check(rc)                                # ... a HIP runtime failure is not (AtacomError)

    def get_constraints_logs(self, clear=True):
        res = (C.c_double * 3)()
        _lib.check(self._lib.atacom_get_stats(self._h, C.byref(res), int(clear), self._stream()))
        return float(res[0]), float(res[1]), float(res[2])

    # ------------------------------------------------------------------ state injection / checkpoint
    def get_state(self):
        """[B, state_dim] = [q, dq, s, puck(6), has_hit, r_hit, vel_hit_x, t]."""
        st = torch.empty((self.batch, self.state_dim), device=self.device, dtype=self.dtype)
        _lib.check(self._lib.atacom_get_state(self._h, _ptr(st), self._stream()))
        return st

    def set_state(self, state):
        """Parity injection of [B, state_dim] (the layout of get_state) -- NOT a checkpoint (snapshot() / restore() are).  On an
        obs_delay handle this RESTARTS the low-pass behind the observed velocities on the injected state, like a reset does, so
        get_state -> set_state does not preserve the filter: to inject a filter state call set_filter_state AFTER set_state."""
        st = self._as_dev(state, (self.batch, self.state_dim))
        _lib.check(self._lib.atacom_set_state(self._h, _ptr(st), self._stream()))

    def get_aux_state(self):
        """Servo joints of the rigid-body mode (iiwa): [B, 6] = [q7, qu1, qu2, dq7, dqu1, dqu2]."""
        aux = torch.empty((self.batch, 6), device=self.device, dtype=self.dtype)
        _lib.check(self._lib.atacom_get_aux_state(self._h, _ptr(aux), self._stream()))
        return aux

    def set_aux_state(self, aux):
        a = self._as_dev(aux, (self.batch, 6))
        _lib.check(self._lib.atacom_set_aux_state(self._h, _ptr(a), self._stream()))

    def get_filter_state(self):
        """obs_delay: the low-pass state behind the observed velocities, [B, 3 + dim_q] = [puck vx, vy, yaw rate, dq]."""
        fv = torch.empty((self.batch, 3 + self.dims['q']), device=self.device, dtype=self.dtype)
        _lib.check(self._lib.atacom_get_filter_state(self._h, _ptr(fv), self._stream()))
        return fv

    def set_filter_state(self, fv):
        """obs_delay: inject the low-pass state.  Call it AFTER set_state (which restarts the filter); the other order is
        overwritten."""
        a = self._as_dev(fv, (self.batch, 3 + self.dims['q']))
        _lib.check(self._lib.atacom_set_filter_state(self._h, _ptr(a), self._stream()))

    def close(self):
        if getattr(self, '_h', None) is not None and self._h:
            self._lib.atacom_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class GraphedRollout:
    """A T-step collection loop -- observe, ARBITRARY torch policy, env step -- captured ONCE in a HIP graph and replayed
    with one host call per collection phase.

    The fused `rollout_policy` kernel covers the reference's actor MLPs; this covers everything else a caller may put
    between observation and action (recurrent nets, ensembles, torch-side preprocessing) without a tracing compiler and
    without per-step Python: the C-ABI entry points only enqueue kernels on the caller's stream, so they are capturable
    as they are.  What it buys depends on how launch-bound the step is (measured, 20-step graphs, MI355X): bare step
    launches circle 7.6 -> 2.7 us per step, planar 11.2 -> 10.7, iiwa 27.8 -> 27.5; this loop (observe + policy output
    copy + step = three kernels per step) 6.2 us per step for the circle against ~20 us for three eager launches.

      loop = GraphedRollout(env, policy, n_steps=120)      # policy: obs [B, D] -> actions [B, k], torch ops on env.device
      data = loop.replay()                                  # dict of static tensors: obs, action, reward, next_obs, absorbing, last

    The returned tensors are the graph's static buffers (overwritten by the next replay).  `policy` must be capturable:
    no host synchronisation, no data-dependent Python control flow."""

    def __init__(self, env, policy, n_steps, warmup=2):
        self.env, self.T = env, int(n_steps)
        T, B, D, k = self.T, env.batch, env.obs_dim, env.dims['null']
        dev, dt = env.device, env.dtype
        self.out = {'obs': torch.empty((T, B, D), device=dev, dtype=dt),
                    'action': torch.empty((T, B, k), device=dev, dtype=dt),
                    'reward': torch.empty((T, B), device=dev, dtype=dt),
                    'next_obs': torch.empty((T, B, D), device=dev, dtype=dt),
                    'absorbing': torch.empty((T, B), device=dev, dtype=torch.uint8),
                    'last': torch.empty((T, B), device=dev, dtype=torch.uint8)}
        self._none = torch.zeros((B,), device=dev, dtype=torch.uint8)       # reset mask selecting nobody = "observe"
        saved = env.snapshot()             # everything: state, statistics, episode counters, servo joints
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                                       # warm-up outside the capture (lazy inits)
            for _ in range(warmup):
                self._body(policy, min(T, 2))
        torch.cuda.current_stream(dev).wait_stream(side)
        env.restore(saved)                 # the warm-up ran real steps: undo them completely (statistics and the random
        self.graph = torch.cuda.CUDAGraph()  # reset's episode counters included -- a replay starts where an eager run would)
        with torch.cuda.graph(self.graph):
            self._body(policy, T)
        env.restore(saved)                 # (the capture itself does not run the kernels)

    def _body(self, policy, n):
        env, o = self.env, self.out
        for t in range(n):
            # observation of the CURRENT state (after an auto-reset it differs from the previous step's terminal obs)
            _lib.check(env._lib.atacom_reset(env._h, _ptr(self._none), None, _ptr(o['obs'][t]), env._stream()))
            o['action'][t].copy_(policy(o['obs'][t]))
            _lib.check(env._lib.atacom_step(env._h, _ptr(o['action'][t]), _ptr(o['next_obs'][t]), _ptr(o['reward'][t]),
                                            _ptr(o['absorbing'][t]), _ptr(o['last'][t]), env._stream()))

    def replay(self):
        self.graph.replay()
        return self.out


class MlpPolicy:
    """Weights of a reference-style actor network for BatchedAtacomEnv.rollout_policy.

    `MlpPolicy.from_module(net)` accepts any module with `_h1/_h2/_h3` nn.Linear layers -- exactly the attribute
    names of the reference's PPONetwork / TRPONetwork / SACActorNetwork (examples/network.py:19-21,50-52,277-279).
    obs_low / obs_high give the MinMaxPreprocessor normalisation x = (obs - mean) / delta of the finite bounds."""

    def __init__(self, W1, b1, W2, b2, W3, b3, std=None, obs_shift=None, obs_scale=None, activation='relu',
                 sigma_weights=None, squash=False, log_std_min=-20.0, log_std_max=2.0):
        self.tensors = dict(W1=W1, b1=b1, W2=W2, b2=b2, W3=W3, b3=b3, std=std, obs_shift=obs_shift, obs_scale=obs_scale)
        if sigma_weights is not None:                     # SAC: second network -> log sigma
            self.tensors.update(dict(zip(('sW1', 'sb1', 'sW2', 'sb2', 'sW3', 'sb3'), sigma_weights)))
        self.activation = {'relu': 0, 'tanh': 1}[activation]
        self.squash, self.log_std_min, self.log_std_max = bool(squash), float(log_std_min), float(log_std_max)
        self._keep = None

    @classmethod
    def from_sac(cls, mu_net, sigma_net, obs_low=None, obs_high=None, log_std_min=-20.0, log_std_max=2.0):
        """The squashed-Gaussian policy SAC samples from: mean and log-sigma networks of the reference's
        SACActorNetwork (examples/network.py:266-293, examples/iiwa_air_hockey_exp.py:301-314), action =
        tanh(mu(obs) + exp(clamp(log_sigma(obs))) * eps)."""
        pol = cls.from_module(mu_net, obs_low=obs_low, obs_high=obs_high)
        g = lambda lin: (lin.weight.detach(), lin.bias.detach())      # noqa: E731
        sw = [t for lin in (sigma_net._h1, sigma_net._h2, sigma_net._h3) for t in g(lin)]
        pol.tensors.update(dict(zip(('sW1', 'sb1', 'sW2', 'sb2', 'sW3', 'sb3'), sw)))
        pol.squash, pol.log_std_min, pol.log_std_max = True, float(log_std_min), float(log_std_max)
        return pol

    @classmethod
    def from_module(cls, net, std=None, obs_low=None, obs_high=None, activation='relu'):
        shift = scale = None
        if obs_low is not None and obs_high is not None:
            lo = torch.as_tensor(obs_low, dtype=torch.float64)
            hi = torch.as_tensor(obs_high, dtype=torch.float64)
            finite = torch.isfinite(lo) & torch.isfinite(hi)
            shift = torch.where(finite, (hi + lo) / 2, torch.zeros_like(lo))
            scale = torch.where(finite, 2.0 / (hi - lo).clamp_min(1e-12), torch.ones_like(lo))
        g = lambda lin: (lin.weight.detach(), lin.bias.detach())      # noqa: E731
        (W1, b1), (W2, b2), (W3, b3) = g(net._h1), g(net._h2), g(net._h3)
        return cls(W1, b1, W2, b2, W3, b3, std=std, obs_shift=shift, obs_scale=scale, activation=activation)

    def as_struct(self, env):
        dev = {k: (None if v is None else torch.as_tensor(v).to(device=env.device, dtype=env.dtype).contiguous())
               for k, v in self.tensors.items()}
        self._keep = dev                                   # keep the device copies alive during the launch
        m = _lib.AtacomMlp()
        m.struct_size = C.sizeof(_lib.AtacomMlp)
        m.n_in, m.hidden, m.n_out = dev['W1'].shape[1], dev['W1'].shape[0], dev['W3'].shape[0]
        if tuple(dev['W2'].shape) != (m.hidden, m.hidden) or dev['W3'].shape[1] != m.hidden:
            raise ValueError("expected Linear(n_in,h) - Linear(h,h) - Linear(h,n_out)")
        m.activation = self.activation
        for k in ('W1', 'b1', 'W2', 'b2', 'W3', 'b3', 'obs_shift', 'obs_scale', 'std', 'sW1', 'sb1', 'sW2', 'sb2',
                  'sW3', 'sb3'):
            v = dev.get(k)
            setattr(m, k, None if v is None else v.data_ptr())
        m.squash, m.log_std_min, m.log_std_max = int(self.squash), self.log_std_min, self.log_std_max
        return m


# ---------------------------------------------------------------------- stand-alone primitives
def nullspace(env, Jc, rhs=None, tol=0.05, lanes_per_env=1):
    """Batched pinv_null + rref on the GPU (null_space_coordinate.py:8-26,40-79).
    Jc [n, c, c+k] (torch, on a ROCm device).  Returns (x = Jc^+ rhs, null basis, rref(null, tol))."""
    lib = _lib.load()
    env_id = _ENV_IDS[env] if isinstance(env, str) else int(env)
    d = _lib.get_dims(env_id)
    c, k = d.n_f + d.n_g, d.n_null
    n = Jc.shape[0]
    assert Jc.is_cuda and tuple(Jc.shape) == (n, c, c + k)
    Jc = Jc.contiguous()
    dt = {torch.float32: _lib.F32, torch.float64: _lib.F64}[Jc.dtype]
    rhs = torch.zeros((n, c), device=Jc.device, dtype=Jc.dtype) if rhs is None else rhs.contiguous()
    x = torch.empty((n, c + k), device=Jc.device, dtype=Jc.dtype)
    nb = torch.empty((n, c + k, k), device=Jc.device, dtype=Jc.dtype)
    rr = torch.empty((n, c + k, k), device=Jc.device, dtype=Jc.dtype)
    stream = C.c_void_p(torch.cuda.current_stream(Jc.device).cuda_stream)
    with torch.cuda.device(Jc.device):
        _lib.check(lib.atacom_nullspace(env_id, dt, int(lanes_per_env), n, _ptr(Jc), _ptr(rhs), float(tol), _ptr(x), _ptr(nb), _ptr(rr),
                                        stream))
    return x, nb, rr


def constraint_terms(env, q, dq, bias_mode='reference'):
    """fun / J / b callables of an environment evaluated on the GPU for q, dq [n, dim_q]."""
    lib = _lib.load()
    env_id = _ENV_IDS[env] if isinstance(env, str) else int(env)
    d = _lib.get_dims(env_id)
    cfg = _lib.default_config(env_id)
    cfg.dtype = {torch.float32: _lib.F32, torch.float64: _lib.F64}[q.dtype]
    cfg.bias_mode = {'reference': 0, 'exact': 1}[bias_mode]
    n, c = q.shape[0], d.n_f + d.n_g
    q, dq = q.contiguous(), dq.contiguous()
    fun = torch.empty((n, c), device=q.device, dtype=q.dtype)
    J = torch.empty((n, c, d.dim_q), device=q.device, dtype=q.dtype)
    b = torch.empty((n, c), device=q.device, dtype=q.dtype)
    stream = C.c_void_p(torch.cuda.current_stream(q.device).cuda_stream)
    with torch.cuda.device(q.device):
        _lib.check(lib.atacom_constraint_terms(C.byref(cfg), n, _ptr(q), _ptr(dq), _ptr(fun), _ptr(J), _ptr(b), stream))
    return fun, J, b


def _check_same(ref, shape, **tensors):
    """Every tensor: a ROCm tensor of `shape` with ref's dtype and device (the primitives take raw pointers)."""
    if ref.dtype not in (torch.float32, torch.float64) or not ref.is_cuda:
        raise ValueError("expected float32 / float64 ROCm tensors")
    for name, t in tensors.items():
        if not isinstance(t, torch.Tensor) or tuple(t.shape) != tuple(shape) or t.dtype != ref.dtype or t.device != ref.device:
            raise ValueError("%s must be a %s tensor of shape %s on %s (got %s %s on %s)"
                             % (name, ref.dtype, tuple(shape), ref.device, getattr(t, 'dtype', type(t)),
                                tuple(getattr(t, 'shape', ())), getattr(t, 'device', None)))


def canonical_mu(env, A, s, y, alpha, tol=0.05):
    """The canonical chart (chart_mode='canonical') as a primitive on the GPU: A [n, c, dim_q] = K J (equality row first),
    s [n, n_g], y [n, c] = psi + Kc c, alpha [n, k]  ->  mu [n, dim_q + n_g] = -Jc^+ y + N alpha.
    Contract: the entries of A that the environment's constraint Jacobian leaves STRUCTURALLY zero (planar / iiwa joint-limit
    rows off their diagonal; iiwa row 4, joints 3..6) are not read -- a general matrix is treated as having zeros there
    (include/atacom_hip.h)."""
    lib = _lib.load()
    env_id = _ENV_IDS[env] if isinstance(env, str) else int(env)
    d = _lib.get_dims(env_id)
    n, c, nq, ng, k = A.shape[0], d.n_f + d.n_g, d.dim_q, d.n_g, d.dim_q - d.n_f
    _check_same(A, (n, c, nq), A=A)
    _check_same(A, (n, ng), s=s)
    _check_same(A, (n, c), y=y)
    _check_same(A, (n, k), alpha=alpha)
    A, s, y, alpha = A.contiguous(), s.contiguous(), y.contiguous(), alpha.contiguous()
    dt = {torch.float32: _lib.F32, torch.float64: _lib.F64}[A.dtype]
    mu = torch.empty((n, nq + ng), device=A.device, dtype=A.dtype)
    stream = C.c_void_p(torch.cuda.current_stream(A.device).cuda_stream)
    with torch.cuda.device(A.device):
        _lib.check(lib.atacom_canonical_mu(env_id, dt, n, _ptr(A), _ptr(s), _ptr(y), _ptr(alpha), float(tol), _ptr(mu), stream))
    return mu


def inverse_dynamics(q, dq, ddq, want_mass_matrix=False):
    """Row N4 primitive: tau = M(q) ddq + C(q, dq) dq + g(q) of the nine-joint iiwa + striker chain on the GPU
    (what the reference asks PyBullet for, iiwa_hit_atacom.py:58-63).  q, dq, ddq [n, 9]; returns tau [n, 9]
    (and M [n, 9, 9])."""
    lib = _lib.load()
    n = q.shape[0]
    _check_same(q, (n, 9), q=q, dq=dq, ddq=ddq)
    q, dq, ddq = q.contiguous(), dq.contiguous(), ddq.contiguous()
    dt = {torch.float32: _lib.F32, torch.float64: _lib.F64}[q.dtype]
    tau = torch.empty((n, 9), device=q.device, dtype=q.dtype)
    M = torch.empty((n, 9, 9), device=q.device, dtype=q.dtype) if want_mass_matrix else None
    stream = C.c_void_p(torch.cuda.current_stream(q.device).cuda_stream)
    with torch.cuda.device(q.device):
        _lib.check(lib.atacom_inverse_dynamics(dt, n, _ptr(q), _ptr(dq), _ptr(ddq), _ptr(tau), _ptr(M), stream))
    return (tau, M) if want_mass_matrix else tau


def forward_dynamics(q, dq, tau6, ddq_aux=None, damping=True):
    """Row N4 primitive: accelerations [n, 6] of the controlled joints under torques tau6 [n, 6]; the servo joints follow
    ddq_aux [n, 3] (None = at rest)."""
    lib = _lib.load()
    n = q.shape[0]
    _check_same(q, (n, 9), q=q, dq=dq)
    _check_same(q, (n, 6), tau6=tau6)
    if ddq_aux is not None:
        _check_same(q, (n, 3), ddq_aux=ddq_aux)
    q, dq, tau6 = q.contiguous(), dq.contiguous(), tau6.contiguous()
    aux = None if ddq_aux is None else ddq_aux.contiguous()
    dt = {torch.float32: _lib.F32, torch.float64: _lib.F64}[q.dtype]
    out = torch.empty((n, 6), device=q.device, dtype=q.dtype)
    stream = C.c_void_p(torch.cuda.current_stream(q.device).cuda_stream)
    with torch.cuda.device(q.device):
        _lib.check(lib.atacom_forward_dynamics(dt, n, _ptr(q), _ptr(dq), _ptr(tau6), _ptr(aux), int(bool(damping)), _ptr(out),
                                               stream))
    return out
