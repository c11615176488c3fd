"""Run-to-run reproducibility: the same launch on the same state must give bitwise identical results."""
import os, sys
import numpy as np, torch
HERE = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests")      # parity_tools etc. live in tests/ (these probes lived there until round 6)
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
from rl_on_manifold_amd import BatchedAtacomEnv
B = 8192
for name in ('iiwa', 'planar'):
    for lanes in (8, 4, 2, 1):
        env = BatchedAtacomEnv(name, B, device='cuda:0', dtype=torch.float32, lanes_per_env=lanes)
        g = torch.Generator(device='cuda:0').manual_seed(0)
        st = env.get_state()
        nq = env.dims['q']
        st[:, :nq] += 0.05 * torch.randn((B, nq), device='cuda:0', generator=g)
        env.set_state(st)
        # advance a few steps so that the states are diverse, then freeze
        for _ in range(20):
            env.step(torch.rand((B, env.dims['null']), device='cuda:0', generator=g) * 2 - 1)
        st = env.get_state().clone()
        a = torch.rand((B, env.dims['null']), device='cuda:0', generator=g) * 2.6 - 1.3
        ref = None
        nbad = 0
        for rep in range(200):
            env.set_state(st)
            obs, r, ab, _ = env.step(a)
            out = torch.cat([obs, r[:, None], env.get_state()], 1)
            if ref is None:
                ref = out.clone()
            else:
                d = (out != ref).any(1)
                nbad += int(d.sum())
                if d.any() and nbad < 20:
                    i = int(torch.nonzero(d)[0])
                    print('   rep', rep, 'env', i, 'maxdiff', float((out[i] - ref[i]).abs().max()))
        print(name, 'lanes', lanes, 'mismatching env-launches over 199 repeats:', nbad, flush=True)
