"""Tiny target for rocprofv3 --pmc passes: a few atacom_step launches, iiwa B=8192, lanes from argv."""
import sys
import torch
sys.path.insert(0, '.')
from rl_on_manifold_amd import BatchedAtacomEnv
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
env = BatchedAtacomEnv('iiwa', B, dtype=torch.float32, auto_reset=True, lanes_per_env=lanes)
gen = torch.Generator(device='cuda:0'); gen.manual_seed(0)
st = env.get_state()
init = torch.zeros((B, env.init_state_dim), device='cuda:0')
init[:, :6] = st[:, :6] + 0.05 * torch.randn((B, 6), device='cuda:0', generator=gen)
init[:, 12:] = st[:, 23:29]
env.reset(state=init)
a = torch.rand((B, 5), device='cuda:0', generator=gen) * 2 - 1
for _ in range(20):
    env.step_into(a, env._obs, env._reward, env._absorbing, env._last)
torch.cuda.synchronize()
