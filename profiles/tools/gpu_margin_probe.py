"""One-off measurement (not a test): float32 HIP vs float64 oracle error of one teacher-forced env step as a function of
the oracle's decision margins (oracle/atacom_batched.py: track_margins).  Output feeds the bounds asserted in
tests/test_gpu_parity.py and profiles/r02_parity_margins.md.   python profiles/tools/gpu_margin_probe.py [lanes] [B] [T]"""
import os
import sys
import numpy as np
import torch
HERE = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests")      # parity_tools etc. live in tests/ (these probes lived there until round 6)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from oracle import atacom_scalar as osc, atacom_batched as ob
from rl_on_manifold_amd import BatchedAtacomEnv
from test_gpu_parity import _full_state

lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
T = int(sys.argv[3]) if len(sys.argv) > 3 else 40
for name, spec in (('planar', osc.planar_spec()), ('iiwa', osc.iiwa_spec())):
    env = BatchedAtacomEnv(name, B, device='cuda:0', dtype=torch.float32, lanes_per_env=lanes)
    nq, ng = spec.dim_q, spec.n_g
    st0 = env.get_state().cpu().numpy().astype(np.float64)
    rng = np.random.default_rng(11)
    o = ob.BatchedAtacomEnv(spec, B, init_q=st0[:, :nq] + rng.normal(0, 0.05, (B, nq)))
    o.track_margins()
    E, M, C, K, EC, SM, TR = [], [], [], [], [], [], []
    for t in range(T):
        a = rng.uniform(-1.3, 1.3, (B, spec.n_null))
        a[: B // 8] = np.sign(a[: B // 8])
        env.set_state(_full_state(env, o))
        obs, r, ab, info = env.step(a)
        oo, orr, oab, _ = o.step(a)
        s_dev = env.get_state().cpu().numpy()[:, 2 * nq:2 * nq + ng]
        e = np.maximum(np.abs(obs.cpu().numpy() - oo).max(1), np.abs(s_dev - o.s).max(1))
        e = np.maximum(e, np.abs(r.cpu().numpy() - orr))
        e = np.maximum(e, (ab.cpu().numpy() != oab) * 1.0)
        od = obs.cpu().numpy()
        EC.append(np.stack([np.abs(od[:, :6] - oo[:, :6]).max(1), np.abs(od[:, 6:6 + nq] - oo[:, 6:6 + nq]).max(1),
                            np.abs(od[:, 6 + nq:] - oo[:, 6 + nq:]).max(1), np.abs(s_dev - o.s).max(1),
                            np.abs(r.cpu().numpy() - orr)], 1))
        SM.append(o.s.min(1)); K.append(o.cond_number.copy())
        E.append(e); M.append(o.decision_margin.copy()); C.append(o.contact_margin.copy())
        last = oab | (o.t >= spec.horizon)
        if last.any():
            o.reset(last)
    E, M, C = np.array(E).ravel(), np.array(M).ravel(), np.array(C).ravel()
    os.makedirs(os.path.join(os.path.dirname(HERE), 'gpurun_out'), exist_ok=True)
    np.savez_compressed(os.path.join(os.path.dirname(HERE), 'gpurun_out', 'r02_margin_%s_l%d.npz' % (name, lanes)), E=E, M=M, C=C,
                        K=np.array(K).ravel(), EC=np.array(EC).reshape(-1, 5), SM=np.array(SM).ravel())
    print('== %s lanes %d: %d env-steps, median err %.2e, p99 %.2e, p99.9 %.2e, max %.2e'
          % (name, lanes, E.size, np.median(E), np.quantile(E, .99), np.quantile(E, .999), E.max()))
    edges = [0, 1e-7, 1e-6, 3e-6, 1e-5, 3e-5, 1e-4, 3e-4, 1e-3, 1e-2, np.inf]
    print('  pivot margin bin        count   frac     max err    p99 err   (contact margin > 1e-4 only)')
    okc = C > 1e-4
    for lo, hi in zip(edges[:-1], edges[1:]):
        m = (M >= lo) & (M < hi) & okc
        if m.any():
            print('  [%.0e, %.0e)  %9d  %.5f  %.3e  %.3e' % (lo, hi, m.sum(), m.mean(), E[m].max(), np.quantile(E[m], .99)))
    print('  contact margin bin      count   frac     max err   (pivot margin > 1e-3 only)')
    okm = M > 1e-3
    for lo, hi in zip(edges[:-1], edges[1:]):
        m = (C >= lo) & (C < hi) & okm
        if m.any():
            print('  [%.0e, %.0e)  %9d  %.5f  %.3e' % (lo, hi, m.sum(), m.mean(), E[m].max()))
    for d in (1e-5, 3e-5, 1e-4):
        m = (M > d) & (C > d)
        print('  outside band delta=%.0e: frac %.5f, max err %.3e, count > 1e-5: %d, > 1e-4: %d'
              % (d, m.mean(), E[m].max(), (E[m] > 1e-5).sum(), (E[m] > 1e-4).sum()))
    env.close()
