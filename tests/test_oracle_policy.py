"""Row N2 oracle pinned to the reference: the actor MLP restatement reproduces the forward() of the reference's
own network classes (examples/network.py), golden set G8."""
import numpy as np
import pytest

from oracle.policy import MlpPolicy, rollout
from oracle import atacom_scalar as osc, atacom_batched as ob


@pytest.mark.parametrize('name', ['ppo_iiwa', 'sac_planar', 'trpo_iiwa'])
def test_mlp_matches_reference_networks(golden, name):
    g = golden('policy_net')
    pol = MlpPolicy(g[name + '._h1.weight'], g[name + '._h1.bias'], g[name + '._h2.weight'], g[name + '._h2.bias'],
                    g[name + '._h3.weight'], g[name + '._h3.bias'])
    y = pol.mean(g[name + '.x'].astype(np.float64))
    assert y.shape == g[name + '.y'].shape
    assert np.abs(y - g[name + '.y']).max() < 2e-5          # the reference evaluates in float32


def test_policy_rollout_shapes_and_noise():
    g = np.random.default_rng(0)
    spec = osc.planar_spec()
    pol = MlpPolicy(g.normal(0, 0.3, (64, 12)), g.normal(0, 0.1, 64), g.normal(0, 0.2, (64, 64)), g.normal(0, 0.1, 64),
                    g.normal(0, 0.2, (3, 64)), np.zeros(3), obs_shift=np.zeros(12), obs_scale=np.ones(12), std=np.full(3, 0.5))
    env = ob.BatchedAtacomEnv(spec, 5)
    eps = g.standard_normal((7, 5, 3))
    out = rollout(env, pol, 7, noise=eps)
    assert out['obs'].shape == (7, 5, 12) and out['action'].shape == (7, 5, 3)
    assert np.allclose(out['action'][0], pol.mean(out['obs'][0]) + 0.5 * eps[0])
    assert np.allclose(out['obs'][1], out['next_obs'][0])
