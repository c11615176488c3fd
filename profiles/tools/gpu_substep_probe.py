"""Diagnostic (not a test): atacom_step time as a function of the number of physics sub-steps -- separates the per-launch
cost (dispatch, state load / store, COLD instruction fetch) from the per-sub-step cost.
    python profiles/tools/gpu_substep_probe.py [reference|canonical] [lanes] [batch]"""
import sys
import torch
sys.path.insert(0, '.')
from rl_on_manifold_amd import BatchedAtacomEnv
chart = sys.argv[1] if len(sys.argv) > 1 else 'canonical'
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 1
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
dev = 'cuda:0'
for n_sub in (1, 2, 4, 8, 16):
    env = BatchedAtacomEnv('iiwa', B, device=dev, auto_reset=True, lanes_per_env=lanes, chart_mode=chart,
                           n_intermediate_steps=n_sub)
    gen = torch.Generator(device=dev); gen.manual_seed(0)
    st = env.get_state()
    init = torch.zeros((B, env.init_state_dim), device=dev)
    init[:, :6] = st[:, :6] + 0.05 * torch.randn((B, 6), device=dev, generator=gen)
    init[:, 12:] = st[:, 23:29]
    env.reset(state=init)
    a = torch.rand((16, B, 5), device=dev, generator=gen) * 2 - 1
    for i in range(10):
        env.step_into(a[i % 16], env._obs, env._reward, env._absorbing, env._last)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(200):
        env.step_into(a[i % 16], env._obs, env._reward, env._absorbing, env._last)
    e1.record(); torch.cuda.synchronize()
    print('%s lanes %d B %d sub-steps %2d: %.1f us per step' % (chart, lanes, B, n_sub, e0.elapsed_time(e1) / 200 * 1e3), flush=True)
