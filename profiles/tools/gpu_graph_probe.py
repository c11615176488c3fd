"""Probe: are the C-ABI launches HIP-graph capturable, and what does a replayed 20-step graph cost per step?"""
import os, sys, time
import torch
HERE = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests")      # parity_tools etc. live in tests/ (these probes lived there until round 6)
sys.path.insert(0, os.path.dirname(HERE))
from rl_on_manifold_amd import BatchedAtacomEnv
dev = 'cuda:0'
for name, B in (('circle', 4096), ('planar', 8192), ('iiwa', 8192)):
    env = BatchedAtacomEnv(name, B, device=dev, auto_reset=True)
    k, D = env.dims['null'], env.obs_dim
    g = torch.Generator(device=dev).manual_seed(0)
    acts = torch.rand((20, B, k), device=dev, generator=g) * 2 - 1
    obs = torch.empty((B, D), device=dev); rew = torch.empty((B,), device=dev)
    ab = torch.empty((B,), device=dev, dtype=torch.uint8); last = torch.empty((B,), device=dev, dtype=torch.uint8)
    st = env.get_state().clone()
    for i in range(20): env.step_into(acts[i], obs, rew, ab, last)
    ref = obs.clone()
    env.set_state(st)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        for i in range(3): env.step_into(acts[i], obs, rew, ab, last)      # warm-up on the side stream
    torch.cuda.current_stream().wait_stream(s)
    env.set_state(st)
    with torch.cuda.graph(graph):
        for i in range(20): env.step_into(acts[i], obs, rew, ab, last)
    env.set_state(st)
    graph.replay(); torch.cuda.synchronize()
    print(name, 'graph replay == eager:', bool(torch.equal(obs, ref)))
    for label, fn in (('eager', lambda: [env.step_into(acts[i], obs, rew, ab, last) for i in range(20)]), ('graph', graph.replay)):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50): fn()
        torch.cuda.synchronize()
        print('   %s: %.2f us per step' % (label, (time.perf_counter() - t0) / 50 / 20 * 1e6))
