"""Tiny target for rocprofv3 --pmc passes: a few policy-rollout launches (the matrix-core policy kernel)."""
import sys
import torch
sys.path.insert(0, '.')
from rl_on_manifold_amd import BatchedAtacomEnv, MlpPolicy
name = sys.argv[1] if len(sys.argv) > 1 else 'iiwa'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
T = 24
env = BatchedAtacomEnv(name, B, dtype=torch.float32, auto_reset=True)
k, D = env.dims['null'], env.obs_dim
g = torch.Generator(device='cpu'); g.manual_seed(0)
W = [torch.randn(64, D, generator=g) * 0.2, torch.zeros(64), torch.randn(64, 64, generator=g) * 0.1, torch.zeros(64),
     torch.randn(k, 64, generator=g) * 0.1, torch.zeros(k)]
pol = MlpPolicy(*W, std=torch.full((k,), 0.5))
eps = torch.randn((T, B, k), device='cuda:0')
for _ in range(4):
    env.rollout_policy(pol, T, noise=eps)
torch.cuda.synchronize()
