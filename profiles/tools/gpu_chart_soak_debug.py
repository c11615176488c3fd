"""Kernel-debugging helper (not a pytest file): the samples of the canonical-chart float32 soak (profiles/tools/gpu_sens_probe.py) whose
error exceeds the quick sensitivity bound, with what the oracle says about them.
    [ATACOM_LIB=...] python profiles/tools/gpu_chart_soak_debug.py [env] [lanes] [B] [T]"""
import dataclasses
import os
import sys
import numpy as np
import torch
HERE = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests")      # parity_tools etc. live in tests/ (these probes lived there until round 6)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from oracle import atacom_scalar as osc, atacom_batched as ob
from rl_on_manifold_amd import BatchedAtacomEnv
from test_gpu_parity import _full_state, _step_outputs
from parity_tools import SensitivityRecorder, C_SENS, FLOOR, slice_env

name = sys.argv[1] if len(sys.argv) > 1 else 'planar'
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
T = int(sys.argv[4]) if len(sys.argv) > 4 else 40
spec = dataclasses.replace({'planar': osc.planar_spec, 'iiwa': osc.iiwa_spec}[name](), chart_mode=1)
envs = {dt: BatchedAtacomEnv(name, B, device='cuda:0', dtype=dt, lanes_per_env=lanes, chart_mode='canonical')
        for dt in (torch.float32, torch.float64)}
nq, ng = spec.dim_q, spec.n_g
st0 = envs[torch.float32].get_state().cpu().numpy().astype(np.float64)
rng = np.random.default_rng(int(os.environ.get('MB_SEED', '11')))
o = ob.BatchedAtacomEnv(spec, B, init_q=st0[:, :nq] + rng.normal(0, 0.05, (B, nq)))
rec = SensitivityRecorder(_step_outputs, seed=5)
tag = os.path.basename(os.environ.get('ATACOM_LIB', 'default'))
n_bad = 0
for t in range(T):
    a = rng.uniform(-1.3, 1.3, (B, spec.n_null))
    a[: B // 8] = np.sign(a[: B // 8])
    dev = {}
    for dt, env in envs.items():
        env.set_state(_full_state(env, o))
        obs, r, ab, info = env.step(a)
        s_dev = env.get_state().cpu().numpy()[:, 2 * nq:2 * nq + ng]
        dev[dt] = np.concatenate([obs.cpu().numpy(), s_dev, r.cpu().numpy()[:, None], ab.cpu().numpy()[:, None] * 1.0], 1).astype(np.float64)
    snap = slice_env(o, np.arange(B))
    base = rec.prepare(o, (a,))
    e32 = (np.abs(dev[torch.float32] - base) / np.maximum(1.0, np.abs(base)))
    e64 = (np.abs(dev[torch.float64] - base) / np.maximum(1.0, np.abs(base))).max(1)
    S = rec.sens[-1]
    bad = np.nonzero(e32.max(1) > C_SENS * S + FLOOR)[0]
    if len(bad):
        sub = slice_env(snap, bad)
        sub.chart_info = {}
        sub.track_margins()
        _step_outputs(sub, (a[bad],))
        arow = np.abs(ob.constraint_terms(spec, snap.q[bad], snap.dq[bad])[1][:, spec.n_f:, :] * spec.K[None, spec.n_f:, None]).max(2)
        for j, b in enumerate(bad):
            n_bad += 1
            rel = np.abs(snap.s[b]) / arow[j]
            print('%s t %d env %d: f32 err %.2e (output %d), f64 err %.1e, quick sens %.1e | slack / row: %s | stiff rows %d | oracle '
                  'margin %.1e, skipped-default %s' % (tag, t, b, e32[b].max(), e32[b].argmax(), e64[b], S[b], np.array2string(rel, precision=4),
                                                       (rel < 3e-2).sum(), sub.decision_margin[j], not sub.chart_default[j]), flush=True)
    o.step(a)
print('%s %s lanes %d: %d of %d samples beyond the quick bound' % (tag, name, lanes, n_bad, B * T))
