"""Tiny target for rocprofv3 --pmc passes: a few atacom_step launches.
    python profiles/tools/gpu_pmc_target.py LANES [BATCH] [ENV] [CHART] [DYNAMICS] [DTYPE]        (defaults: 0 8192 iiwa reference kinematic f32)"""
import sys
import torch
sys.path.insert(0, '.')
from rl_on_manifold_amd import BatchedAtacomEnv
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
name = sys.argv[3] if len(sys.argv) > 3 else 'iiwa'
chart = sys.argv[4] if len(sys.argv) > 4 else 'reference'
dyn = sys.argv[5] if len(sys.argv) > 5 else 'kinematic'
DT = {'f32': torch.float32, 'f64': torch.float64}[sys.argv[6] if len(sys.argv) > 6 else 'f32']
env = BatchedAtacomEnv(name, B, dtype=DT, auto_reset=True, lanes_per_env=lanes, chart_mode=chart, dynamics_mode=dyn)
gen = torch.Generator(device='cuda:0'); gen.manual_seed(0)
st = env.get_state()
nq, ng, k = env.dims['q'], env.dims['g'], env.dims['null']
if name != 'circle':
    init = torch.zeros((B, env.init_state_dim), device='cuda:0', dtype=DT)
    init[:, :nq] = st[:, :nq] + 0.05 * torch.randn((B, nq), device='cuda:0', generator=gen).to(DT)
    init[:, 2 * nq:] = st[:, 2 * nq + ng:2 * nq + ng + 6]
    env.reset(state=init)
a = (torch.rand((B, k), device='cuda:0', generator=gen) * 2 - 1).to(DT)
for _ in range(20):
    env.step_into(a, env._obs, env._reward, env._absorbing, env._last)
torch.cuda.synchronize()
