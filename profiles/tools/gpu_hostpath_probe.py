"""Tuning helper (not a pytest file): where a launch-bound step goes -- CPU time per step_into call against the GPU's
step-to-step period, circle / planar / iiwa."""
import sys, time
import torch
sys.path.insert(0, '.')
from rl_on_manifold_amd import BatchedAtacomEnv
for name, B in (('circle', 4096), ('planar', 8192), ('iiwa', 8192)):
    env = BatchedAtacomEnv(name, B, device='cuda:0', auto_reset=True)
    a = torch.zeros((B, env.dims['null']), device='cuda:0')
    args = (a, env._obs, env._reward, env._absorbing, env._last)
    for _ in range(200): env.step_into(*args)
    torch.cuda.synchronize()
    # (a) CPU cost of a call while the queue is short: bursts of 32 calls, synchronise in between
    cpu = []
    for _ in range(50):
        t0 = time.perf_counter()
        for _ in range(32): env.step_into(*args)
        cpu.append((time.perf_counter() - t0) / 32)
        torch.cuda.synchronize()
    # (b) sustained period
    n = 5000
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): env.step_into(*args)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize(); t_all = time.perf_counter() - t0
    # (c) the same through the raw ctypes call (no Python-side checks)
    from rl_on_manifold_amd import engine as E
    lib, h = env._lib, env._h
    p = [E._ptr(t) for t in args]
    s = env._stream()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): lib.atacom_step(h, p[0], p[1], p[2], p[3], p[4], s)
    t_enq_raw = time.perf_counter() - t0
    torch.cuda.synchronize(); t_raw = time.perf_counter() - t0
    bound = env.bind_step(*args)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): bound()
    t_enq_b = time.perf_counter() - t0
    torch.cuda.synchronize(); t_b = time.perf_counter() - t0
    print('        bind_step: enqueue %.2f us, period %.2f us' % (t_enq_b / n * 1e6, t_b / n * 1e6))
    print('%-7s B=%d  CPU per step_into (short queue) %.2f us median | sustained: enqueue %.2f us, period %.2f us | raw ctypes: enqueue %.2f us, period %.2f us'
          % (name, B, sorted(cpu)[len(cpu) // 2] * 1e6, t_enq / n * 1e6, t_all / n * 1e6, t_enq_raw / n * 1e6, t_raw / n * 1e6), flush=True)
