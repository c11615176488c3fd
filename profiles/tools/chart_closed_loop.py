import sys, torch
sys.path.insert(0, '.')          # run from the repo root on the GPU box
import bench
from rl_on_manifold_amd import BatchedAtacomEnv
DEV = 'cuda:0'
B, T = 8192, 120
for seed in (0, 1, 2):
    gen = torch.Generator(device=DEV); gen.manual_seed(seed)
    init = bench.feasible_init('iiwa', B, torch.device(DEV), gen)[0]
    acts = torch.rand((T, B, 5), device=DEV, generator=gen) * 2 - 1
    for mode in ('reference', 'canonical'):
        for lanes in (1, 2, 4, 8):
            env = BatchedAtacomEnv('iiwa', B, device=DEV, chart_mode=mode, auto_reset=True, lanes_per_env=lanes)
            env.reset(state=init)
            env.rollout(acts)
            print(seed, mode, lanes, 'c_avg %.5f c_max %.5f c_dq_max %.2e' % env.get_constraints_logs(), flush=True)
