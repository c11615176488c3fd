"""Kernel-tuning helper (not a pytest file): the phase probe of profiles/tools/gpu_phase_probe.py on the BENCH workload -- feasible
perturbed initial states (bench.feasible_init), random actions, auto-reset, free-running -- over many launches: what a
launch lasts (HIP events), what its median and its slowest wavefront compute, and (canonical chart) how often the
data-dependent parts run.  Needs the -DATACOM_TIMESTAMPS build: ATACOM_LIB=build/ts/libatacom_ts.so python profiles/tools/gpu_bench_probe.py [lanes]
"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, '.')
import bench
from rl_on_manifold_amd import BatchedAtacomEnv

lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B = int(os.environ.get('MB_BATCH', 8192))
chart = os.environ.get('MB_CHART', 'reference')
dev = torch.device('cuda:0')
gen = torch.Generator(device=dev); gen.manual_seed(1234)
env, init, rej = bench.make_env('iiwa', B, dev, gen, lanes, chart_mode=chart, dynamics_mode=os.environ.get('MB_DYN', 'kinematic'))
L = env.lanes_per_env
acts = torch.rand((64, B, 5), device=dev, generator=gen) * 2 - 1
obs, rew = torch.empty((B, 18), device=dev), torch.empty((B,), device=dev)
ab, last = torch.empty((B,), device=dev, dtype=torch.uint8), torch.empty((B,), device=dev, dtype=torch.uint8)
for i in range(100):
    env.step_into(acts[i % 64], obs, rew, ab, last)
torch.cuda.synchronize()
step = max(1, 64 // L)
rows, cnts = [], []
N = int(os.environ.get('MB_LAUNCHES', 240))
for rep in range(N):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    env.step_into(acts[rep % 64], obs, rew, ab, last)
    e1.record(); torch.cuda.synchronize()
    raw = obs[:, :7].contiguous().view(torch.int32).cpu().numpy()
    ts = raw[:, :4].astype(np.int64) & 0xFFFFFFFF
    d = (ts - ts[:, 0].min()) * 0.01
    w = d[::step]
    comp = w[:, 2] - w[:, 1]
    rows.append([e0.elapsed_time(e1) * 1e3, w[:, 0].max(), np.median(w[:, 1] - w[:, 0]), np.median(comp), comp.max(),
                 np.percentile(comp, 90), w[:, 3].max(), np.median(w[:, 3] - w[:, 2])])
    if chart == 'canonical':
        c = raw[::step, 4:7]
        cnts.append(np.concatenate([c.mean(0), c.max(0), [(c[:, 1] > 0).mean(), (c[:, 2] > 0).mean(), (c[:, 0] > 0).mean()]]))
        if rep == N - 1:
            Xm = np.concatenate([c.astype(float), np.ones((len(c), 1))], 1)
            coef = np.linalg.lstsq(Xm, comp, rcond=None)[0]
            print('   last launch, least squares on wave compute time: %.2f us per stiff-row trip, %.2f per stage A trip, %.2f per '
                  'stage B, %.2f base' % tuple(coef))
r = np.array(rows)
print('%s chart, lanes %d, B %d, %d launches on the bench workload (us): events mean %.1f / min %.1f / max %.1f / std %.2f | '
      'last wave starts %.2f | load %.2f | compute: median wave %.2f, p90 %.2f, slowest %.2f | store %.2f | last store lands %.2f'
      % (chart, L, B, N, r[:, 0].mean(), r[:, 0].min(), r[:, 0].max(), r[:, 0].std(), r[:, 1].mean(), r[:, 2].mean(),
         r[:, 3].mean(), r[:, 5].mean(), r[:, 4].mean(), r[:, 7].mean(), r[:, 6].mean()))
if cnts:
    c = np.array(cnts).mean(0)
    print('   per wave and launch (mean): stiff-row trips %.2f, stage A trips %.2f, stage B %.2f | max over the waves of a launch: '
          '%.1f / %.1f / %.1f | share of waves with any: stage A %.3f, stage B %.3f, stiff %.3f' % tuple(c))
print('   constraint logs', env.get_constraints_logs())
