"""Float64 numpy restatement of the policy side of row N2 (oracle; test infrastructure only).

Network: examples/network.py:8-36 (PPONetwork) / :39-68 (TRPONetwork) / :266-293 (SACActorNetwork) of the
reference -- Linear(n_in, h) -> ReLU -> Linear(h, h) -> ReLU -> Linear(h, n_out), torch.nn.Linear layout
W[out][in].  Policy: mean + std * eps (GaussianTorchPolicy, examples/iiwa_air_hockey_exp.py:138-146) on
min-max normalised observations (MinMaxPreprocessor, :32-34)."""
import numpy as np


class MlpPolicy:
    def __init__(self, W1, b1, W2, b2, W3, b3, obs_shift=None, obs_scale=None, std=None, activation='relu',
                 sigma_weights=None, squash=False, log_std_min=-20.0, log_std_max=2.0):
        self.W1, self.b1, self.W2, self.b2, self.W3, self.b3 = (np.asarray(a, dtype=np.float64)
                                                                for a in (W1, b1, W2, b2, W3, b3))
        n_in, n_out = self.W1.shape[1], self.W3.shape[0]
        self.shift = np.zeros(n_in) if obs_shift is None else np.asarray(obs_shift, dtype=np.float64)
        self.scale = np.ones(n_in) if obs_scale is None else np.asarray(obs_scale, dtype=np.float64)
        self.std = np.zeros(n_out) if std is None else np.asarray(std, dtype=np.float64)
        self.act = (lambda v: np.maximum(v, 0.0)) if activation == 'relu' else np.tanh
        # SAC (examples/iiwa_air_hockey_exp.py:301-339): a second network gives log sigma, the sample is tanh-squashed
        self.sigma_weights = None if sigma_weights is None else [np.asarray(a, dtype=np.float64) for a in sigma_weights]
        self.squash, self.log_std_min, self.log_std_max = squash, log_std_min, log_std_max

    def mean(self, obs):
        x = (obs - self.shift) * self.scale
        h1 = self.act(x @ self.W1.T + self.b1)
        h2 = self.act(h1 @ self.W2.T + self.b2)
        return h2 @ self.W3.T + self.b3

    def sigma(self, obs):
        if self.sigma_weights is None:
            return self.std
        W1, b1, W2, b2, W3, b3 = self.sigma_weights
        x = (obs - self.shift) * self.scale
        ls = self.act(self.act(x @ W1.T + b1) @ W2.T + b2) @ W3.T + b3
        return np.exp(np.clip(ls, self.log_std_min, self.log_std_max))

    def draw(self, obs, eps=None):
        a = self.mean(obs)
        if eps is not None:
            a = a + self.sigma(obs) * eps
        return np.tanh(a) if self.squash else a


def rollout(env, policy, n_steps, noise=None, auto_reset=True):
    """T steps of a batched oracle env driven by the policy; returns time-major arrays like atacom_rollout_mlp."""
    out = {k: [] for k in ('obs', 'action', 'reward', 'next_obs', 'absorbing', 'last')}
    for t in range(n_steps):
        o = env.observation()
        a = policy.draw(o, None if noise is None else noise[t])
        no, r, ab, _ = env.step(a)
        last = ab | (env.t >= env.spec.horizon)
        for k, v in zip(out, (o, a, r, no, ab, last)):
            out[k].append(np.array(v).copy())
        if auto_reset and last.any():
            env.reset(last)
    return {k: np.stack(v) for k, v in out.items()}
