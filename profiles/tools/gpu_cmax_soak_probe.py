"""Audit helper (not a pytest file): the largest constraint value of the iiwa endurance soak (profiles/tools/gpu_long_soak.py, same seeds
and action pool) -- where it happens, and whether the float64 oracle, teacher-forced from the device's own states, produces the
same violation.  Pass 1 finds the window of W steps holding the maximum (statistics read every W steps), pass 2 replays the
deterministic run to that window and keeps every state of it.
    python profiles/tools/gpu_cmax_soak_probe.py [STEPS] [W]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rl_on_manifold_amd import BatchedAtacomEnv, constraint_terms
from oracle import atacom_scalar as osc, atacom_batched as ob
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
dev, B = 'cuda:0', 8192


def make():
    env = BatchedAtacomEnv('iiwa', B, device=dev, auto_reset=True, random_init=True, seed=7)
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    acts = torch.rand((256, B, 5), device=dev, generator=gen) * 2 - 1
    acts[::7] = torch.sign(acts[::7])
    obs = torch.empty((B, env.obs_dim), device=dev); rew = torch.empty((B,), device=dev)
    ab = torch.empty((B,), device=dev, dtype=torch.uint8); la = torch.empty((B,), device=dev, dtype=torch.uint8)
    return env, acts, [env.bind_step(acts[i], obs, rew, ab, la) for i in range(256)]


def cvals(q):
    fun, _, _ = constraint_terms('iiwa', q, torch.zeros_like(q))
    return torch.maximum(fun[:, 0].abs(), fun[:, 1:].max(1).values)

env, acts, steppers = make()
best = (-1.0, -1)
for it in range(STEPS):
    steppers[it % 256]()
    if (it + 1) % W == 0:
        c = env.get_constraints_logs()[1]
        if c > best[0]:
            best = (c, it // W)
print('pass 1: largest c_max %.5f in window %d (steps %d .. %d)' % (best[0], best[1], best[1] * W, best[1] * W + W - 1), flush=True)
env.close()
env, acts, steppers = make()
w0 = best[1] * W
for it in range(w0):
    steppers[it % 256]()
    if (it + 1) % W == 0:
        env.get_constraints_logs()
states = torch.empty((W + 1, B, env.state_dim), device=dev)
for k in range(W):
    states[k] = env.get_state()
    steppers[(w0 + k) % 256]()
states[W] = env.get_state()
cs = torch.stack([cvals(states[k + 1][:, :6]) for k in range(W)])
# (a step that ended an episode is followed by the reset state: its c is the reset pose's, small -- the maximum is a real step)
flat = int(cs.argmax()); k_star, b = flat // B, flat % B
print('pass 2: c %.5f at step %d (window offset %d), env %d; device statistics of the window: %s' % (
    float(cs[k_star, b]), w0 + k_star, k_star, b, env.get_constraints_logs()), flush=True)
print('   that environment over the steps before: ' + ' '.join('%.4f' % float(x) for x in cs[max(0, k_star - 10):k_star + 2, b]))
spec = osc.iiwa_spec()
nq, ng = 6, 11
for k in range(max(0, k_star - 7), k_star + 1):
    st = states[k][b:b + 1].double().cpu().numpy()
    nxt = states[k + 1][b:b + 1]
    if int(nxt[0, -1]) == 0:
        print('step %d: episode ended here (auto-reset follows)' % (w0 + k)); continue
    o = ob.BatchedAtacomEnv(spec, 1, init_q=st[:, :nq])
    o.set_state(st[:, :nq], st[:, nq:2 * nq], st[:, 2 * nq:2 * nq + ng], st[:, 2 * nq + ng:2 * nq + ng + 6])
    o.t[:] = int(st[0, -1])
    o.track_margins(True)
    o.step(acts[(w0 + k) % 256][b:b + 1].double().cpu().numpy())
    qo = torch.tensor(o.q, device=dev, dtype=torch.float32)
    print('step %7d: c after the step  device f32 %.5f | oracle f64 from the same state %.5f   |q32 - oracle| %.2e   '
          'oracle: rref skipped %s, decision margin %.2e, cond %.1e' % (w0 + k, float(cvals(nxt[:, :6])), float(cvals(qo)),
          float((nxt[:, :6].double().cpu() - torch.tensor(o.q)).abs().max()), bool(o.chart_skipped[0]), float(o.decision_margin[0]),
          float(o.cond_number[0])), flush=True)
