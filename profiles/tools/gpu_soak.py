"""Bounded soak / scale run (not a pytest file): large batches, long rollouts, finiteness and constraint statistics."""
import sys, time
import torch
sys.path.insert(0, '.')
from rl_on_manifold_amd import BatchedAtacomEnv
dev = 'cuda:0'
for name, B, T in (('iiwa', 262144, 120), ('iiwa', 1048576, 24), ('planar', 1048576, 60), ('circle', 4194304, 100)):
    env = BatchedAtacomEnv(name, B, device=dev, dtype=torch.float32, auto_reset=True)
    k, nq = env.dims['null'], env.dims['q']
    gen = torch.Generator(device=dev); gen.manual_seed(0)
    if name != 'circle':
        st = env.get_state()
        init = torch.zeros((B, env.init_state_dim), device=dev)
        init[:, :nq] = st[:, :nq] + 0.05 * torch.randn((B, nq), device=dev, generator=gen)
        init[:, 2 * nq:] = st[:, 2 * nq + env.dims['g']: 2 * nq + env.dims['g'] + 6]
        env.reset(state=init)
    acts = torch.rand((T, B, k), device=dev, generator=gen) * 2 - 1
    out = env.rollout(acts, want_next_obs=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = env.rollout(acts, want_next_obs=False, out=out)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ok = bool(torch.isfinite(out['obs']).all() and torch.isfinite(out['reward']).all())
    print('%s B=%d T=%d: %.3g env-steps/s, finite=%s, logs=%s, mem=%.2f GB' % (
        name, B, T, B * T / dt, ok, env.get_constraints_logs(), torch.cuda.max_memory_allocated() / 1e9), flush=True)
    del env, out, acts
    torch.cuda.empty_cache()
