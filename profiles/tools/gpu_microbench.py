"""Kernel-tuning helper (not a pytest file): time atacom_step / rollout for one library build."""
import os, sys, time
import torch
sys.path.insert(0, '.')
from rl_on_manifold_amd import BatchedAtacomEnv
dev = 'cuda:0'
DT = {'f32': torch.float32, 'f64': torch.float64}[os.environ.get('MB_DTYPE', 'f32')]
tag = os.environ.get('ATACOM_LIB', 'default')
for name in sys.argv[1:] or ['iiwa']:
    for B in [int(x) for x in os.environ.get("MB_BATCHES", "8192").split(",")]:
      for lanes in [int(x) for x in os.environ.get("MB_LANES", "1,4").split(",")]:
        env = BatchedAtacomEnv(name, B, device=dev, dtype=DT, auto_reset=True, lanes_per_env=lanes,
                               dynamics_mode=os.environ.get('MB_DYN', 'kinematic'),
                               chart_mode=os.environ.get('MB_CHART', 'reference'))
        k = env.dims['null']
        gen = torch.Generator(device=dev); gen.manual_seed(0)
        st = env.get_state(); nq = env.dims['q']
        if name != 'circle':
            init = torch.zeros((B, env.init_state_dim), device=dev, dtype=DT)
            init[:, :nq] = st[:, :nq] + 0.05 * torch.randn((B, nq), device=dev, generator=gen).to(DT)
            init[:, 2 * nq:] = st[:, 2 * nq + env.dims['g']: 2 * nq + env.dims['g'] + 6]
            env.reset(state=init)
        a = (torch.rand((16, B, k), device=dev, generator=gen) * 2 - 1).to(DT)
        for i in range(int(os.environ.get('MB_WARM', '10'))): env.step_into(a[i % 16], env._obs, env._reward, env._absorbing, env._last)
        torch.cuda.synchronize()
        # let the clocks settle: 100 launches are 3 ms, and the configuration measured first in a process read 1 - 1.5 us
        # per step slower than the same configuration measured second (profiles/r04_microbench_order.log)
        t_end = time.perf_counter() + float(os.environ.get('MB_SETTLE_MS', '300')) * 1e-3
        while time.perf_counter() < t_end:
            for i in range(64): env.step_into(a[i % 16], env._obs, env._reward, env._absorbing, env._last)
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 100
        e0.record()
        for i in range(n): env.step_into(a[i % 16], env._obs, env._reward, env._absorbing, env._last)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        print('%s %s lanes=%d B=%d step %.1f us -> %.3g env-steps/s  logs %s' % (os.path.basename(tag), name, lanes, B, us, B / us * 1e6, env.get_constraints_logs()), flush=True)
        if os.environ.get('MB_ROLLOUT'):
            # the T-step kernels: plain rollout and (planar / iiwa) the rollout with the actor MLP inside
            from rl_on_manifold_amd import MlpPolicy
            T = 40
            acts = (torch.rand((T, B, k), device=dev, generator=gen) * 2 - 1).to(DT)
            out = env.rollout(acts)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(5): env.rollout(acts, out=out)
            e1.record(); torch.cuda.synchronize()
            ur = e0.elapsed_time(e1) / (5 * T) * 1e3
            msg = 'rollout %.1f us/step' % ur
            if name != 'circle' and DT == torch.float32:
                g2 = torch.Generator(device='cpu'); g2.manual_seed(0)
                D = env.obs_dim
                W = [torch.randn((64, D), generator=g2) * 0.1, torch.zeros(64), torch.randn((64, 64), generator=g2) * 0.1,
                     torch.zeros(64), torch.randn((k, 64), generator=g2) * 0.1, torch.zeros(k)]
                pol = MlpPolicy(*[w.to(dev) for w in W], std=torch.ones(k, device=dev))
                eps = torch.randn((T, B, k), device=dev, generator=gen)
                env.rollout_policy(pol, T, noise=eps)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(5): env.rollout_policy(pol, T, noise=eps)
                e1.record(); torch.cuda.synchronize()
                msg += ', policy rollout %.1f us/step' % (e0.elapsed_time(e1) / (5 * T) * 1e3)
            print('    ' + msg, flush=True)
