"""Kernel-tuning helper (not a pytest file): where do the ~26 us of one atacom_step launch go?

Needs a library built with -DATACOM_TIMESTAMPS (ATACOM_HIPCC_FLAGS=-DATACOM_TIMESTAMPS ATACOM_LIB_OUT=... python -m
rl_on_manifold_amd.build), whose k_step overwrites obs[:, 0:4] with four 100 MHz wall-clock stamps per environment:
wave start, state loaded, sub-steps done, stores landed.  Usage: ATACOM_LIB=<that .so> python profiles/tools/gpu_phase_probe.py [lanes]
"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, '.')
from rl_on_manifold_amd import BatchedAtacomEnv

lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = int(os.environ.get('MB_BATCH', 8192))
env = BatchedAtacomEnv('iiwa', B, device='cuda:0', dtype=torch.float32, lanes_per_env=lanes,
                       chart_mode=os.environ.get('MB_CHART', 'reference'))
a = torch.zeros((B, 5), device='cuda:0')
if os.environ.get('MB_RANDOM'):
    # the states of profiles/tools/gpu_microbench.py: perturbed start, random actions, 60 steps in -- constraints become active
    gen = torch.Generator(device='cuda:0'); gen.manual_seed(0)
    st = env.get_state(); nq, ng = env.dims['q'], env.dims['g']
    init = torch.zeros((B, env.init_state_dim), device='cuda:0')
    init[:, :nq] = st[:, :nq] + 0.05 * torch.randn((B, nq), device='cuda:0', generator=gen)
    init[:, 2 * nq:] = st[:, 2 * nq + ng: 2 * nq + ng + 6]
    env.reset(state=init)
    acts = torch.rand((16, B, 5), device='cuda:0', generator=gen) * 2 - 1
    a = acts[0]
for i in range(int(os.environ.get('MB_WARM', 20))):
    env.step_into(acts[i % 16] if os.environ.get('MB_RANDOM') else a, env._obs, env._reward, env._absorbing, env._last)
torch.cuda.synchronize()
rows = []
for rep in range(10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    env.step_into(a, env._obs, env._reward, env._absorbing, env._last)
    e1.record(); torch.cuda.synchronize()
    raw = env._obs[:, :7].contiguous().view(torch.int32).cpu().numpy()
    ts = env._obs[:, :4].contiguous().view(torch.int32).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    t0 = ts[:, 0].min()
    d = (ts - t0) * 0.01                                   # us since the first wave started
    rows.append([e0.elapsed_time(e1) * 1e3, d[:, 0].max(), np.median(d[:, 1] - d[:, 0]), np.median(d[:, 2] - d[:, 1]),
                 np.median(d[:, 3] - d[:, 2]), d[:, 3].max(), np.median(d[:, 3] - d[:, 0])])
r = np.median(np.array(rows), 0)
print('lanes %d, B %d: kernel (events) %.1f us | last wave starts %.2f us after the first | per wave (median): load %.2f, '
      'compute %.2f, store %.2f, total %.2f | last store lands %.2f us after the first wave started'
      % (lanes, B, r[0], r[1], r[2], r[3], r[4], r[6], r[5]))
# distribution over waves (one row per environment; take one per wave)
step = max(1, 64 // lanes)
tot = (d[::step, 3] - d[::step, 0])
ld = (d[::step, 1] - d[::step, 0])
cp = (d[::step, 2] - d[::step, 1])
st0 = d[::step, 0]
q = [0, 10, 50, 90, 99, 100]
print('   waves %d | total us pct%s: %s | load: %s | compute: %s | start: %s' % (
    len(tot), q, np.percentile(tot, q).round(1), np.percentile(ld, q).round(1), np.percentile(cp, q).round(1),
    np.percentile(st0, q).round(1)))
slow = np.argsort(-tot)[:8]
print('   slowest waves (index: total)', [(int(i), float(tot[i].round(1))) for i in slow])
if os.environ.get('MB_CHART') == 'canonical':
    # wave-level path counters of the canonical chart over the sub-steps of the launch (csrc/atacom_chart.h, tuning build)
    c = raw[::step, 4:7]
    print('   per wave and launch: stiff-row trips %.2f, stage A trips %.2f, stage B %.2f' % tuple(c.mean(0)))
    for name, col in (('stiff-row trips', 0), ('stage A trips', 1), ('stage B', 2)):
        vals = np.unique(c[:, col])
        print('   wave time by %s: ' % name + ', '.join('%d: %.1f us (%d waves)' % (v, tot[c[:, col] == v].mean(), (c[:, col] == v).sum())
                                                      for v in vals))
    Xm = np.concatenate([c.astype(float), np.ones((len(c), 1))], 1)
    coef = np.linalg.lstsq(Xm, tot, rcond=None)[0]
    print('   least squares: %.2f us per stiff-row trip, %.2f us per stage A trip, %.2f us per stage B, %.2f us base' % tuple(coef))
