#!/usr/bin/env python3
"""Minimal on-GPU PPO on the batched ATACOM air-hockey hitting task -- an end-to-end use of the engine.

Not part of the hot path: the reference trains with MushroomRL's PPO (examples/planar_air_hockey_exp.py,
examples/iiwa_air_hockey_exp.py:137-170), which is not installed here.  This script shows the same loop shape on the
engine: collection = ONE kernel launch per iteration (policy MLP + exploration noise + ATACOM env step fused,
`rollout_policy`), policy / value update = plain torch autograd on the GPU.  Network = the reference's PPONetwork
(examples/network.py:8-36: Linear-ReLU-Linear-ReLU-Linear, 64 units), Gaussian policy with state-independent std.

    python examples/ppo_air_hockey.py --env planar --iters 60
"""
import argparse
import os
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from anywhere in the checkout
from rl_on_manifold_amd import BatchedAtacomEnv, MlpPolicy


class Net(nn.Module):                      # same layer names as the reference's PPONetwork
    def __init__(self, n_in, n_out, h=64):
        super().__init__()
        self._h1, self._h2, self._h3 = nn.Linear(n_in, h), nn.Linear(h, h), nn.Linear(h, n_out)
        for lin, g in ((self._h1, 'relu'), (self._h2, 'relu'), (self._h3, 'linear')):
            nn.init.xavier_uniform_(lin.weight, gain=nn.init.calculate_gain(g))

    def forward(self, x):
        return self._h3(torch.relu(self._h2(torch.relu(self._h1(x)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--env', default='planar')
    ap.add_argument('--task', default='H', choices=['H', 'D'], help="planar only: 'D' = defending (horizon 180 in the reference)")
    ap.add_argument('--batch', type=int, default=4096)
    ap.add_argument('--iters', type=int, default=60)
    ap.add_argument('--horizon', type=int, default=120)
    ap.add_argument('--lr', type=float, default=3e-4)
    ap.add_argument('--seed', type=int, default=0)
    args = ap.parse_args()
    torch.manual_seed(args.seed)
    dev = torch.device('cuda:0')
    B, T = args.batch, args.horizon
    env = BatchedAtacomEnv(args.env, B, device=dev, auto_reset=True, horizon=T, random_init=True, seed=args.seed, task=args.task)
    D, k = env.obs_dim, env.dims['null']
    # observation normalisation (what MinMaxPreprocessor does with finite bounds): fixed shift / scale
    shift = torch.zeros(D, device=dev)
    scale = torch.ones(D, device=dev)
    shift[0] = 1.1
    scale[3:6] = 0.2
    actor, critic = Net(D, k).to(dev), Net(D, 1).to(dev)
    log_std = torch.full((k,), -0.7, device=dev, requires_grad=True)          # std_0 = 0.5
    opt = torch.optim.Adam(list(actor.parameters()) + list(critic.parameters()) + [log_std], lr=args.lr)
    gamma, lam, clip, epochs, mb = 0.99, 0.95, 0.2, 4, 16384
    norm = lambda o: (o - shift) * scale                                     # noqa: E731
    t_collect = t_fit = 0.0
    for it in range(args.iters):
        t0 = time.perf_counter()
        pol = MlpPolicy.from_module(actor, std=log_std.detach().exp())
        pol.tensors['obs_shift'], pol.tensors['obs_scale'] = shift, scale
        noise = torch.randn((T, B, k), device=dev)
        env.reset()
        d = env.rollout_policy(pol, T, noise=noise)                           # ONE launch: T steps of B envs
        c_avg, c_max, c_dq = env.get_constraints_logs()
        torch.cuda.synchronize()
        t_collect += time.perf_counter() - t0
        t0 = time.perf_counter()
        obs, nobs, act, rew = norm(d['obs']), norm(d['next_obs']), d['action'], d['reward']
        ab, last = d['absorbing'].bool(), d['last'].bool()
        with torch.no_grad():
            v, nv = critic(obs).squeeze(-1), critic(nobs).squeeze(-1)
            nv = torch.where(ab, torch.zeros_like(nv), nv)
            adv = torch.zeros_like(rew)
            g = torch.zeros(B, device=dev)
            for t in reversed(range(T)):
                delta = rew[t] + gamma * nv[t] - v[t]
                g = delta + gamma * lam * torch.where(last[t], torch.zeros_like(g), g)
                adv[t] = g
            ret = adv + v
            adv = (adv - adv.mean()) / (adv.std() + 1e-8)
            mu_old = actor(obs)
            logp_old = (-0.5 * ((act - mu_old) / log_std.exp()) ** 2 - log_std).sum(-1)
        flat = lambda x: x.reshape(T * B, *x.shape[2:])                       # noqa: E731
        fo, fa, fadv, fret, flp = flat(obs), flat(act), flat(adv), flat(ret), flat(logp_old)
        for _ in range(epochs):
            perm = torch.randperm(T * B, device=dev)
            for i in range(0, T * B, mb):
                idx = perm[i:i + mb]
                mu = actor(fo[idx])
                logp = (-0.5 * ((fa[idx] - mu) / log_std.exp()) ** 2 - log_std).sum(-1)
                ratio = (logp - flp[idx]).exp()
                pl = -torch.min(ratio * fadv[idx], ratio.clamp(1 - clip, 1 + clip) * fadv[idx]).mean()
                vl = (critic(fo[idx]).squeeze(-1) - fret[idx]).pow(2).mean()
                opt.zero_grad()
                (pl + 0.5 * vl).backward()
                opt.step()
        torch.cuda.synchronize()
        t_fit += time.perf_counter() - t0
        ep_ret = rew.sum(0).mean().item() if not last[:-1].any() else (rew.sum() / last.sum().clamp_min(1)).item()
        goals = ((rew > 70) if args.task == 'H' else (rew < -40)).sum().item()      # task 'D': goals CONCEDED
        hits = (d['reward'] > 0.99).any(0).float().mean().item()
        print('iter %3d  return/episode %8.3f  goals %5d  frac envs with a hit %.3f  c_max %.4f  c_dq_max %.4f  std %.3f'
              % (it, ep_ret, goals, hits, c_max, c_dq, log_std.exp().mean().item()), flush=True)
    print('collection %.2f s (%.3g env-steps/s incl. policy), fitting %.2f s' % (
        t_collect, args.iters * T * B / t_collect, t_fit))


if __name__ == '__main__':
    main()
