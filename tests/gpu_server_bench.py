"""Measurement for the step-server experiment (not a pytest file):  python tests/gpu_server_bench.py
us per env step of the bench workload (8192 iiwa environments, feasible starts) through
  (a) atacom_step launches, (b) the same launches replayed from a HIP graph, (c) the step server --
each with the actions (1) already on the device, (2) produced by an external torch policy per step (obs @ W -> tanh -> scale:
two torch kernels + the copy into the action buffer)."""
import os
import sys
import time

import torch
sys.path.insert(0, '.')
import bench
from rl_on_manifold_amd import GraphedRollout

dev = torch.device('cuda:0')
B = int(os.environ.get('MB_BATCH', 8192))
T = int(os.environ.get('MB_STEPS', 120))
chart = os.environ.get('MB_CHART', 'reference')
name = os.environ.get('MB_ENV', 'iiwa')


def fresh():
    gen = torch.Generator(device=dev); gen.manual_seed(1234)
    env, _, _ = bench.make_env(name, B, dev, gen, 0, chart_mode=chart)
    return env, gen


def timed(fn, reps=5):
    fn()
    torch.cuda.current_stream().synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.current_stream().synchronize()
        ts.append((time.perf_counter() - t0) / T * 1e6)
    return min(ts), sorted(ts)[len(ts) // 2]


env, gen = fresh()
k, D = env.dims['null'], env.obs_dim
acts = torch.rand((T, B, k), device=dev, generator=gen) * 2 - 1
W = torch.randn((D, k), device=dev, generator=gen) * 0.3
policy = lambda o: torch.tanh(o @ W) * 1.1          # noqa: E731
out = (env._obs, env._reward, env._absorbing, env._last)
res = {}


def eager_fixed():
    for t in range(T):
        env.step_into(acts[t], *out)


def eager_policy():
    for t in range(T):
        env.step_into(policy(env._obs), *out)


res['atacom_step launches, actions on the device'] = timed(eager_fixed)
res['atacom_step launches, torch policy per step'] = timed(eager_policy)
print('lanes: step %d' % env.lanes_per_env, flush=True)

# HIP graph of the whole loop (observe, policy, step): engine.GraphedRollout
env2, _ = fresh()
gr = GraphedRollout(env2, policy, T)
res['HIP graph of (observe, torch policy, atacom_step) x T'] = timed(gr.replay)
del gr
env2.close()

# the step server, both transports of a submission
buf = torch.empty((B, k), device=dev)
buf.copy_(acts[0])
n_srv = 7 * T
for transport in ('kernel', 'stream_ops'):
    for what in ('actions on the device', 'torch policy per step'):
        e, _ = fresh()
        srv = e.serve(buf, max_steps=n_srv, timeout_s=5.0, transport=transport)

        def loop():
            for t in range(T):
                if what.startswith('torch'):
                    buf.copy_(policy(srv.obs))
                srv.submit()
        t0 = time.perf_counter()
        r = timed(loop, reps=5)
        res['step server (%s), %s' % (transport, what)] = r
        t1 = time.perf_counter()
        try:
            srv.stop()
        except Exception as ex:  # noqa: BLE001
            print('  !! %s / %s: %s' % (transport, what, ex), flush=True)
        print('  server %s / %s: 6 x %d submissions in %.3f s, stop %.3f s' % (transport, what, T, t1 - t0, time.perf_counter() - t1),
              flush=True)
        logs = e.get_constraints_logs()
        e.close()
print('%s, %d environments, %s chart, %d steps per measurement (us per step: best / median of 5)' % (name, B, chart, T))
for kk, v in res.items():
    print('  %-58s %7.2f / %7.2f' % (kk, v[0], v[1]))
print('  constraint logs (last server handle):', logs)
