"""Sharded on-policy rollout collection: one process per GPU, the env batch split into contiguous blocks,
NO collective while stepping (environments are independent), and ONE all-gather of the finished rollout
buffer per collection phase (RCCL over xGMI on the GPU box: `torch.distributed` backend "nccl"; the same code
runs on "gloo" for the CPU tests).

The reference has no distributed code at all; this is the data-parallel axis the
workload offers: BASELINE.json config 5 = 65536 IiwaAirHockey envs = 8 GPUs x 8192.

Data path of one collection (every pass over the data is listed; there are two):
  1. the rollout kernel writes the (state, action, reward, next_state, absorbing, last) tuples mushroom_rl.Core.learn
     hands to an on-policy agent as packed float records [T, B_shard, F] (atacom_rollout_packed, one launch);
  2. one `all_gather_into_tensor` of that buffer into [W, T, B_shard, F] -- which IS the final layout
     ("shard-major"): `unpack` returns views into it, `reshape(-1, F)` is the flat sample set a PPO / TRPO fit
     consumes, and the time axis of every environment stays strided-contiguous for GAE.
No size exchange (every rank's shard size is a pure function of (global_batch, world)), no packing copy, no
concatenation.  Ragged shards (global_batch not a multiple of the world size) are padded to the largest shard: block r
of the gathered buffer holds sizes[r] valid env rows followed by ZERO rows, so `reshape(-1, F)` of a ragged gather also
contains those padding records -- `RecordLayout.valid_mask()` selects the real ones, `time_major()` drops them.  For config 5: 120 x 8192 x 44 floats = 173 MB sent per rank, 1.38 GB received; xGMI is
point-to-point, so one large collective amortises the per-link setup far better than six small ones.
Constraint statistics are reduced with one MAX and one SUM all-reduce of two numbers each.
"""
import torch
import torch.distributed as dist


def shard_bounds(global_batch, world_size, rank):
    """Contiguous block [lo, hi) of the global env index range owned by `rank` (remainder to the low ranks)."""
    base, rem = divmod(int(global_batch), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class RecordLayout:
    """The shard-major layout of a gathered collection, [W, T, Bm, F] with F = 2 D + k + 3 packed floats per (step, env):
    [obs | action | reward | next_obs | absorbing | last].  Pure index arithmetic -- usable without a process group
    (e.g. for a buffer that several engines of ONE process filled block by block)."""

    def __init__(self, sizes, obs_dim, n_null):
        self.sizes = [int(s) for s in sizes]
        self.world = len(self.sizes)
        self.Bm = max(self.sizes)
        self.D, self.k = int(obs_dim), int(n_null)
        self.F = 2 * self.D + self.k + 3

    def unpack(self, g):
        """Views (no copy) into records [..., F]."""
        D, k = self.D, self.k
        return {'obs': g[..., :D], 'action': g[..., D:D + k], 'reward': g[..., D + k],
                'next_obs': g[..., D + k + 1:2 * D + k + 1], 'absorbing': g[..., 2 * D + k + 1] > 0.5,
                'last': g[..., 2 * D + k + 2] > 0.5}

    def time_major(self, data):
        """[W, T, Bm, ...] -> [T, B_global, ...] with rank r's envs in block shard_bounds(global_batch, W, r); the
        padding rows of ragged shards are dropped.  This one COPIES (a permute + concatenation)."""
        return {key: torch.cat([v[r, :, :self.sizes[r]] for r in range(self.world)], 1) for key, v in data.items()}

    def valid_mask(self, device=None):
        """bool [W, Bm]: True where block r, env row b is a real environment (False on the padding of ragged shards)."""
        idx = torch.arange(self.Bm, device=device)
        return idx[None, :] < torch.tensor(self.sizes, device=device)[:, None]


class RolloutCollector:
    """Drive one local engine (a BatchedAtacomEnv, or anything with its surface) and assemble global rollouts.

    env          : local engine holding this rank's shard (env.batch envs)
    group        : torch.distributed process group (None = default group; no-op if dist is not initialised)
    global_batch : total number of envs over all ranks (default env.batch * world, i.e. equal shards); ragged
                   shards follow shard_bounds(global_batch, world, rank) and are padded to the largest shard
    """

    def __init__(self, env, group=None, global_batch=None, force_collective=False):
        self.env = env
        self.group = group
        # force_collective: run the collectives even in a world of one rank (they are a copy through the backend then):
        # the way to exercise the RCCL transport -- buffer registration, the backend's stream, async work objects -- on a
        # single GPU.  Needs an initialised process group.
        self.force_collective = bool(force_collective)
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.rank = dist.get_rank(group) if self.distributed else 0
        self.global_batch = int(global_batch) if global_batch is not None else env.batch * self.world
        self.sizes = []
        for r in range(self.world):
            lo, hi = shard_bounds(self.global_batch, self.world, r)
            self.sizes.append(hi - lo)
        if self.sizes[self.rank] != env.batch:
            raise ValueError("rank %d holds %d envs, shard_bounds(%d, %d) assigns it %d"
                             % (self.rank, env.batch, self.global_batch, self.world, self.sizes[self.rank]))
        self.Bm = max(self.sizes)                 # env-axis length of every rank's send buffer
        self.k = env.dims['null']
        self.D = env.obs_dim
        self.F = 2 * self.D + self.k + 3          # obs, action, reward, next_obs, absorbing, last
        self.layout = RecordLayout(self.sizes, self.D, self.k)
        self._recv = None
        self.mappings = self._agree_on_mappings()

    def _agree_on_mappings(self):
        """The kernel mappings (lanes per environment of step() and of the T-step kernels) sum in different orders, so
        ranks that hold equally sized shards must run the SAME mappings or a sharded collection is not reproducible
        against a single-process run of the same batch size per rank.  The library's policy is static (a pure function of
        the configuration, include/atacom_hip.h), so this can only fail when a rank was configured differently
        (lanes_per_env, ATACOM_CALIBRATE=1): checked ONCE here, at construction -- one all-gather of three integers,
        never in the data path -- and refused loudly.  Returns [(batch, step_lanes, rollout_lanes)] per rank."""
        mine = self._my_mappings()
        self._agreed = tuple(mine)
        if self.world == 1:
            return [tuple(mine)]
        dev = getattr(self.env, 'device', torch.device('cpu'))
        if dist.get_backend(self.group) == 'gloo':
            dev = torch.device('cpu')
        t = torch.tensor(mine, dtype=torch.int64, device=dev)
        allm = torch.empty((self.world * 3,), dtype=torch.int64, device=dev)     # the concatenated form every backend takes
        dist.all_gather_into_tensor(allm, t, group=self.group)
        rows = [tuple(int(v) for v in r) for r in allm.cpu().view(self.world, 3)]
        by_batch = {}
        for r, (b, sl, rl) in enumerate(rows):
            first = by_batch.setdefault(b, (r, sl, rl))
            if (sl, rl) != first[1:]:
                raise ValueError("ranks %d and %d hold shards of %d environments but run different kernel mappings "
                                 "(step %d / rollout %d lanes per environment against %d / %d): name lanes_per_env on "
                                 "every rank (and leave ATACOM_CALIBRATE unset) -- the bits depend on the mapping"
                                 % (first[0], r, b, first[1], first[2], sl, rl))
        return rows

    def _my_mappings(self):
        return [int(self.env.batch), int(getattr(self.env, 'lanes_per_env', 0)),
                int(getattr(self.env, 'rollout_lanes_per_env', 0))]

    def _check_mappings_unchanged(self):
        """A snapshot restore can make a handle ADOPT the image writer's kernel mappings (include/atacom_hip.h:
        atacom_snapshot_restore).  The agreement above was reached at construction: a rank whose mappings have changed since
        refuses to collect -- locally, no collective -- until a new collector is built (on every rank)."""
        now = tuple(self._my_mappings())
        if now != self._agreed:
            raise ValueError("this rank's kernel mappings changed after the collector was built (batch, step, rollout lanes "
                             "%s -> %s; a snapshot restore adopts the image's mappings): build a new RolloutCollector on every "
                             "rank" % (self._agreed, now))

    # ------------------------------------------------------------------ local collection
    def collect_local(self, n_steps, actions=None, policy=None, noise=None, out=None):
        """T = n_steps env steps of the local shard -> packed records [T, Bm, F].
        actions [T, B_local, k]  : pre-generated actions, ONE kernel launch;
        policy = MlpPolicy       : the actor network evaluated inside the rollout kernel, ONE launch (`noise` optional);
        policy = callable        : policy(obs) -> actions, one launch per step (host-driven loop)."""
        env = self.env
        self._check_mappings_unchanged()
        fused = hasattr(env, 'rollout_packed')
        if fused and actions is not None:
            return env.rollout_packed(actions=actions, out=out, batch_stride=self.Bm)
        if fused and policy is not None and hasattr(policy, 'as_struct'):
            return env.rollout_packed(policy=policy, n_steps=n_steps, noise=noise, out=out, batch_stride=self.Bm)
        # engines without the packed kernel (the CPU test double) and host-side policies: pack here
        B = env.batch
        if actions is not None:
            o = env.rollout(actions)
            obs, nobs, rew = o['obs'], o['next_obs'], o['reward']
            ab, last, act = o['absorbing'], o['last'], o['action']
        else:
            assert policy is not None
            obs_l, act_l, rew_l, nobs_l, ab_l, last_l = [], [], [], [], [], []
            o = env.reset()
            for _ in range(n_steps):
                a = policy(o)
                no, r, absorbing, info = env.step(a)
                obs_l.append(o); act_l.append(torch.as_tensor(a, dtype=no.dtype, device=no.device))
                rew_l.append(r); nobs_l.append(no); ab_l.append(absorbing); last_l.append(info['last'])
                o = no
                if bool(info['last'].any()):
                    # mushroom_rl.Core resets finished episodes between steps; engines created with
                    # auto_reset=True have already done it on the device, others get a masked reset
                    if not getattr(env, 'cfg', None) or not env.cfg.auto_reset:
                        o = env.reset(mask=info['last'])
                    else:
                        o = env.reset(mask=torch.zeros_like(info['last']))
            obs, act, rew = torch.stack(obs_l), torch.stack(act_l), torch.stack(rew_l)
            nobs, ab, last = torch.stack(nobs_l), torch.stack(ab_l), torch.stack(last_l)
        T = obs.shape[0]
        buf = torch.zeros((T, self.Bm, self.F), device=obs.device, dtype=obs.dtype) if out is None else out
        if out is not None and self.Bm > B:
            out[:, B:] = 0                         # a caller's buffer may hold anything: the padding rows are zero
        D, k = self.D, self.k
        buf[:, :B, :D] = obs
        buf[:, :B, D:D + k] = act
        buf[:, :B, D + k] = rew
        buf[:, :B, D + k + 1:2 * D + k + 1] = nobs
        buf[:, :B, 2 * D + k + 1] = ab.to(obs.dtype)
        buf[:, :B, 2 * D + k + 2] = last.to(obs.dtype)
        return buf

    # ------------------------------------------------------------------ the one collective
    def gather(self, buf, out=None, async_op=False):
        """All-gather the packed rollout: [T, Bm, F] on every rank -> [W, T, Bm, F] on every rank, rank r's shard in
        block r (its first sizes[r] env rows are valid).  One collective, written straight into the final buffer.
        Without `out` the result lives in a buffer the collector keeps and REUSES: the next gather overwrites it (an
        on-policy learner consumes a dataset before collecting the next one); pass `out` or clone to keep it.
        async_op=True returns (result, work): the collective runs on the backend's own stream (RCCL) while the caller
        goes on -- e.g. launches the next rollout -- and `work.wait()` orders the result before its first use."""
        if self.world == 1 and not (self.force_collective and self.distributed):
            if out is not None:
                out.view(buf.shape).copy_(buf)
                return (out, _Done()) if async_op else out
            return (buf.unsqueeze(0), _Done()) if async_op else buf.unsqueeze(0)
        T, Bm, F = buf.shape
        shape = (self.world, T, Bm, F)
        if out is not None and (tuple(out.shape) != shape or out.dtype != buf.dtype or not out.is_contiguous()):
            raise ValueError("out must be a contiguous %s tensor of dtype %s" % (shape, buf.dtype))
        if buf.is_cuda and dist.get_backend(self.group) == 'gloo':
            # gloo moves host memory: the control-flow check mode (several ranks sharing one GPU in the tests;
            # BENCH_DIST_BACKEND=gloo).  The production transport is RCCL, device to device, below.
            host = torch.empty(shape, dtype=buf.dtype)
            dist.all_gather_into_tensor(host.view(self.world * T, Bm, F), buf.cpu(), group=self.group)
            res = host.to(buf.device) if out is None else out.copy_(host)
            return (res, _Done()) if async_op else res
        if out is None:
            if self._recv is None or tuple(self._recv.shape) != shape or self._recv.dtype != buf.dtype \
                    or self._recv.device != buf.device:
                self._recv = torch.empty(shape, device=buf.device, dtype=buf.dtype)
            out = self._recv
        # output handed over as the concatenation along dim 0 (the form every backend accepts)
        work = dist.all_gather_into_tensor(out.view(self.world * T, Bm, F), buf, group=self.group, async_op=async_op)
        return (out, work) if async_op else out

    def unpack(self, g):
        """Views (no copy) into gathered records [W, T, Bm, F]: every entry is [W, T, Bm, ...]."""
        return self.layout.unpack(g)

    def time_major(self, data):
        """[W, T, Bm, ...] -> [T, B_global, ...] with rank r's envs in block shard_bounds(global_batch, W, r).
        This one COPIES (a permute + concatenation); it is for consumers that insist on a single env axis and for
        comparing against a single-process run -- the collection path itself never needs it."""
        return self.layout.time_major(data)

    def collect(self, n_steps, actions=None, policy=None, noise=None):
        """Local rollout + global all-gather.  Returns the unpacked global dataset, shard-major [W, T, Bm, ...]."""
        return self.unpack(self.gather(self.collect_local(n_steps, actions=actions, policy=policy, noise=noise)))

    def collect_async(self, n_steps, actions=None, policy=None, noise=None, out=None):
        """Like collect(), but the all-gather is left in flight: returns a PendingRollout whose .wait() gives the
        dataset.  Lets a learner overlap the collective (1.4 GB received per rank for config 5) with whatever it does
        next on the compute stream -- typically the first kernels of its update, or the next rollout into another
        buffer (pass a distinct `out` per buffer in flight)."""
        local = self.collect_local(n_steps, actions=actions, policy=policy, noise=noise)
        g, work = self.gather(local, out=out, async_op=True)
        return PendingRollout(self, g, work, local)

    # ------------------------------------------------------------------ constraint statistics
    def get_constraints_logs(self, n_logged):
        """Global (c_avg, c_max, c_dq_max): the reference's get_constraints_logs (atacom.py:207-216) over every env
        of every rank.  n_logged = number of (env, step) entries this rank logged since the last call."""
        c_avg, c_max, c_dq = self.env.get_constraints_logs()
        if self.world == 1 and not (self.force_collective and self.distributed):
            return c_avg, c_max, c_dq
        dev = getattr(self.env, 'device', torch.device('cpu'))
        if self.distributed and dist.get_backend(self.group) == 'gloo':
            dev = torch.device('cpu')
        mx = torch.tensor([c_max, c_dq], dtype=torch.float64, device=dev)
        sm = torch.tensor([c_avg * n_logged, float(n_logged)], dtype=torch.float64, device=dev)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=self.group)
        dist.all_reduce(sm, op=dist.ReduceOp.SUM, group=self.group)
        return float(sm[0] / sm[1]), float(mx[0]), float(mx[1])


class _Done:
    def wait(self):
        return True


class PendingRollout:
    """A collection whose all-gather may still be running (RolloutCollector.collect_async)."""

    def __init__(self, collector, gathered, work, local):
        self._c, self._g, self._work, self._local = collector, gathered, work, local     # `local` is the send buffer:
                                                                                           # kept alive until the wait
    def wait(self):
        self._work.wait()
        self._local = None
        return self._c.unpack(self._g)


def to_mushroom_dataset(data):
    """Flatten a time-major rollout [T, B, ...] into MushroomRL's list-of-tuples dataset (s, a, r, s', absorbing,
    last), env by env (what Core.learn would have produced running the envs one after another)."""
    obs, act, rew = (data[k].cpu().numpy() for k in ('obs', 'action', 'reward'))
    nobs, ab, last = (data[k].cpu().numpy() for k in ('next_obs', 'absorbing', 'last'))
    T, B = rew.shape
    out = []
    for b in range(B):
        for t in range(T):
            out.append((obs[t, b], act[t, b], float(rew[t, b]), nobs[t, b], bool(ab[t, b]), bool(last[t, b]) or t == T - 1))
    return out
