"""Sharded on-policy rollout collection: one process per GPU, the env batch split into contiguous blocks,
NO collective while stepping (environments are independent), and ONE all-gather of the finished rollout
buffer per collection phase (RCCL over xGMI on the GPU box: `torch.distributed` backend "nccl"; the same code
runs on "gloo" for the CPU tests).

The reference has no distributed code at all; this is the data-parallel axis the
workload offers: BASELINE.json config 5 = 65536 IiwaAirHockey envs = 8 GPUs x 8192.

What is gathered: the (state, action, reward, next_state, absorbing, last) tuples mushroom_rl.Core.learn hands to
an on-policy agent (PPO / TRPO fit on the whole dataset), packed into ONE float buffer [T, B_local, F] per
rank so a single large collective moves it (for config 5: 120 x 8192 x 44 floats = 173 MB per rank).  xGMI is
point-to-point, so one big all-gather amortises the per-link setup far better than six small ones.
Constraint statistics are reduced with one MAX and one SUM all-reduce of two numbers each.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(global_batch, world_size, rank):
    """Contiguous block [lo, hi) of the global env index range owned by `rank` (remainder to the low ranks)."""
    base, rem = divmod(int(global_batch), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class RolloutCollector:
    """Drive one local engine (a BatchedAtacomEnv, or anything with its surface) and assemble global rollouts.

    env        : local engine holding this rank's shard (env.batch envs)
    group      : torch.distributed process group (None = default group; no-op if dist is not initialised)
    """

    def __init__(self, env, group=None):
        self.env = env
        self.group = group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.rank = dist.get_rank(group) if self.distributed else 0
        self.k = env.dims['null']
        self.D = env.obs_dim
        self.F = 2 * self.D + self.k + 3          # obs, action, reward, next_obs, absorbing, last

    # ------------------------------------------------------------------ local collection
    def collect_local(self, n_steps, actions=None, policy=None):
        """T = n_steps env steps of the local shard.  Either `actions` [T, B_local, k] (pre-generated, ONE kernel
        launch via env.rollout) or `policy(obs) -> actions` (one launch per step).  Returns the packed buffer
        [T, B_local, F]."""
        env = self.env
        B = env.batch
        if actions is not None:
            out = env.rollout(actions)
            obs, nobs, rew = out['obs'], out['next_obs'], out['reward']
            ab, last, act = out['absorbing'], out['last'], out['action']
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
        buf = torch.empty((T, B, self.F), device=obs.device, dtype=obs.dtype)
        D, k = self.D, self.k
        buf[..., :D] = obs
        buf[..., D:D + k] = act
        buf[..., D + k] = rew
        buf[..., D + k + 1:2 * D + k + 1] = nobs
        buf[..., 2 * D + k + 1] = ab.to(obs.dtype)
        buf[..., 2 * D + k + 2] = last.to(obs.dtype)
        return buf

    # ------------------------------------------------------------------ the one collective
    def gather(self, buf):
        """All-gather the packed rollout: [T, B_local, F] on every rank -> [T, B_global, F] on every rank
        (rank r's envs occupy the contiguous block shard_bounds(...) gives it)."""
        if self.world == 1:
            return buf
        T, B, F = buf.shape
        # equal shards are the common case (one all_gather_into_tensor); ragged shards fall back to padding
        bmax = torch.tensor([B], device=buf.device, dtype=torch.int64)
        blist = [torch.zeros_like(bmax) for _ in range(self.world)]
        dist.all_gather(blist, bmax, group=self.group)
        bs = [int(x.item()) for x in blist]
        Bm = max(bs)
        send = buf if B == Bm else torch.cat([buf, buf.new_zeros((T, Bm - B, F))], 1)
        send = send.contiguous()
        recv = torch.empty((self.world, T, Bm, F), device=buf.device, dtype=buf.dtype)
        try:
            dist.all_gather_into_tensor(recv, send, group=self.group)
        except (RuntimeError, NotImplementedError):
            parts = [torch.empty_like(send) for _ in range(self.world)]
            dist.all_gather(parts, send, group=self.group)
            recv = torch.stack(parts)
        return torch.cat([recv[r, :, :bs[r]] for r in range(self.world)], 1)

    def unpack(self, buf):
        D, k = self.D, self.k
        return {'obs': buf[..., :D], 'action': buf[..., D:D + k], 'reward': buf[..., D + k],
                'next_obs': buf[..., D + k + 1:2 * D + k + 1], 'absorbing': buf[..., 2 * D + k + 1] > 0.5,
                'last': buf[..., 2 * D + k + 2] > 0.5}

    def collect(self, n_steps, actions=None, policy=None):
        """Local rollout + global all-gather.  Returns the unpacked global dataset (time-major)."""
        return self.unpack(self.gather(self.collect_local(n_steps, actions=actions, policy=policy)))

    # ------------------------------------------------------------------ constraint statistics
    def get_constraints_logs(self, n_logged):
        """Global (c_avg, c_max, c_dq_max): the reference's get_constraints_logs (atacom.py:207-216) over every env
        of every rank.  n_logged = number of (env, step) entries this rank logged since the last call."""
        c_avg, c_max, c_dq = self.env.get_constraints_logs()
        if self.world == 1:
            return c_avg, c_max, c_dq
        dev = getattr(self.env, 'device', torch.device('cpu'))
        mx = torch.tensor([c_max, c_dq], dtype=torch.float64, device=dev)
        sm = torch.tensor([c_avg * n_logged, float(n_logged)], dtype=torch.float64, device=dev)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=self.group)
        dist.all_reduce(sm, op=dist.ReduceOp.SUM, group=self.group)
        return float(sm[0] / sm[1]), float(mx[0]), float(mx[1])


def to_mushroom_dataset(data):
    """Flatten a time-major rollout into MushroomRL's list-of-tuples dataset (s, a, r, s', absorbing, last),
    env by env (what Core.learn would have produced running the envs one after another)."""
    obs, act, rew = (data[k].cpu().numpy() for k in ('obs', 'action', 'reward'))
    nobs, ab, last = (data[k].cpu().numpy() for k in ('next_obs', 'absorbing', 'last'))
    T, B = rew.shape
    out = []
    for b in range(B):
        for t in range(T):
            out.append((obs[t, b], act[t, b], float(rew[t, b]), nobs[t, b], bool(ab[t, b]), bool(last[t, b]) or t == T - 1))
    return out
