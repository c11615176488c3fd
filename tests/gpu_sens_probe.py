"""One-off measurement (not a test): is every float32 HIP-vs-oracle error of a teacher-forced env step explained by the
reference algorithm's own sensitivity to float32-sized input perturbations?   python tests/gpu_sens_probe.py [lanes] [B] [T]"""
import copy
import os
import sys
import numpy as np
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from oracle import atacom_scalar as osc, atacom_batched as ob
from rl_on_manifold_amd import BatchedAtacomEnv
from test_gpu_parity import _full_state

lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
T = int(sys.argv[3]) if len(sys.argv) > 3 else 40
SCALES = (2e-7, 1e-6, 4e-6)


def outputs(o, res):
    oo, orr, oab, _ = res
    return np.concatenate([oo, o.s, orr[:, None]], 1)


for name, spec in (('planar', osc.planar_spec()), ('iiwa', osc.iiwa_spec())):
    env = BatchedAtacomEnv(name, B, device='cuda:0', dtype=torch.float32, lanes_per_env=lanes)
    nq, ng = spec.dim_q, spec.n_g
    st0 = env.get_state().cpu().numpy().astype(np.float64)
    rng = np.random.default_rng(11)
    o = ob.BatchedAtacomEnv(spec, B, init_q=st0[:, :nq] + rng.normal(0, 0.05, (B, nq)))
    o.track_margins()
    E, S, M, C = [], [], [], []
    for t in range(T):
        a = rng.uniform(-1.3, 1.3, (B, spec.n_null))
        a[: B // 8] = np.sign(a[: B // 8])
        env.set_state(_full_state(env, o))
        obs, r, ab, info = env.step(a)
        perts = []
        for sc in SCALES:
            for draw in range(2):
                p = copy.deepcopy(o)
                p.track_margins(False)
                for arr in (p.q, p.dq, p.s, p.puck):
                    arr *= 1.0 + sc * rng.choice([-1.0, 1.0], arr.shape)
                ap = a * (1.0 + sc * rng.choice([-1.0, 1.0], a.shape))
                perts.append(outputs(p, p.step(ap)))
        res = o.step(a)
        basev = outputs(o, res)
        sens = np.max([np.abs(pp - basev).max(1) for pp in perts], 0)
        dev = np.concatenate([obs.cpu().numpy(), env.get_state().cpu().numpy()[:, 2 * nq:2 * nq + ng],
                              r.cpu().numpy()[:, None]], 1)
        e = np.abs(dev - basev).max(1)
        e = np.maximum(e, (ab.cpu().numpy() != res[2]) * 1.0)
        E.append(e); S.append(sens); M.append(o.decision_margin.copy()); C.append(o.contact_margin.copy())
        last = res[2] | (o.t >= spec.horizon)
        if last.any():
            o.reset(last)
    E, S, M, C = (np.array(x).ravel() for x in (E, S, M, C))
    np.savez_compressed(os.path.join(os.path.dirname(HERE), 'gpurun_out', 'r02_sens_%s_l%d.npz' % (name, lanes)), E=E, S=S, M=M, C=C)
    ratio = E / np.maximum(S, 1e-7)
    print('== %s lanes %d: %d samples; err median %.2e max %.2e; sens median %.2e max %.2e' % (name, lanes, E.size, np.median(E), E.max(), np.median(S), S.max()))
    print('   err / max(sens, 1e-7): median %.2f p99 %.2f p99.9 %.2f max %.2f' % (np.median(ratio), np.quantile(ratio, .99), np.quantile(ratio, .999), ratio.max()))
    worst = np.argsort(-ratio)[:8]
    for i in worst:
        print('   ratio %.1f err %.2e sens %.2e M %.2e C %.2e' % (ratio[i], E[i], S[i], M[i], C[i]))
    env.close()
