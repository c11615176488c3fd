"""Test double with the BatchedAtacomEnv surface, backed by the float64 oracle on the CPU.  Lives under
tests/ on purpose: the product package never falls back to it; it only lets the multi-process collection
logic (rl_on_manifold_amd/rollout.py) be exercised with gloo where there is no GPU."""
import numpy as np
import torch

from oracle import atacom_scalar as osc
from oracle import atacom_batched as ob


class OracleEngine:
    def __init__(self, name, batch, init_q=None, horizon=None, auto_reset=True):
        spec = {'circle': osc.circle_spec, 'planar': osc.planar_spec, 'iiwa': osc.iiwa_spec}[name]()
        if horizon is not None:
            spec.horizon = horizon
        self.spec = spec
        self.env = ob.BatchedAtacomEnv(spec, batch, init_q=init_q)
        self.batch = batch
        self.dims = {'q': spec.dim_q, 'f': spec.n_f, 'g': spec.n_g, 'null': spec.n_null, 'c': spec.n_c}
        self.obs_dim = spec.obs_dim
        self.device = torch.device('cpu')
        self.auto_reset = auto_reset

    def reset(self, mask=None, state=None):
        m = None if mask is None else np.asarray(mask, dtype=bool)
        return torch.tensor(self.env.reset(m))

    def step(self, actions):
        o, r, ab, _ = self.env.step(np.asarray(actions, dtype=np.float64))
        last = ab | (self.env.t >= self.spec.horizon)
        out = (torch.tensor(o), torch.tensor(r), torch.tensor(ab), {'last': torch.tensor(last)})
        if self.auto_reset and last.any():
            self.env.reset(last)
        return out

    def rollout(self, actions, want_next_obs=True, out=None):
        T = actions.shape[0]
        obs, nobs, rew, ab, last = [], [], [], [], []
        for t in range(T):
            obs.append(torch.tensor(self.env.observation()))
            o, r, a, info = self.step(actions[t])
            nobs.append(o); rew.append(r); ab.append(a); last.append(info['last'])
        return {'obs': torch.stack(obs), 'next_obs': torch.stack(nobs), 'reward': torch.stack(rew),
                'absorbing': torch.stack(ab), 'last': torch.stack(last),
                'action': torch.as_tensor(actions, dtype=torch.float64)}

    def get_constraints_logs(self):
        return self.env.get_constraints_logs()
