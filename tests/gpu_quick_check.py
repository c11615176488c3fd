"""Ad-hoc first GPU check (not a pytest file): HIP vs batched oracle, one step + short rollouts."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
from rl_on_manifold_amd import BatchedAtacomEnv, nullspace, constraint_terms
from oracle import atacom_scalar as osc, atacom_batched as ob

torch.manual_seed(0)
dev = 'cuda:0'
g = np.load('tests/golden/nullspace.npz')
for name in ['circle', 'planar', 'iiwa']:
    for dt in [torch.float64, torch.float32]:
        Jc = torch.tensor(g[name + '_Jc'], device=dev, dtype=dt)
        rhs = torch.ones(Jc.shape[:2], device=dev, dtype=dt)
        x, nb, rr = nullspace(name, Jc, rhs)
        xr = np.einsum('bnc,bc->bn', g[name + '_pinv'], np.ones(Jc.shape[:2]))
        e_x = np.abs(x.cpu().numpy() - xr).max()
        e_n = np.abs(nb.cpu().numpy() - g[name + '_null']).max() if name != 'circle' else np.abs(np.abs(nb.cpu().numpy()) - np.abs(g[name + '_null'])).max()
        d = np.abs(rr.cpu().numpy() - g[name + '_rref']).reshape(len(Jc), -1).max(1)
        print(name, dt, 'pinv err', e_x, 'null err', e_n, 'rref err max', d.max(), 'n bad', (d > 1e-3).sum(), flush=True)

for name, spec in [('circle', osc.circle_spec()), ('planar', osc.planar_spec()), ('iiwa', osc.iiwa_spec())]:
    for dt in [torch.float64, torch.float32]:
        B = 256
        env = BatchedAtacomEnv(name, B, device=dev, dtype=dt)
        st = env.get_state().cpu().numpy().astype(np.float64)
        nq, ng = spec.dim_q, spec.n_g
        rng = np.random.default_rng(1)
        init_q = st[0, :nq]
        o = ob.BatchedAtacomEnv(spec, B, init_q=init_q)
        print(name, dt, 'init s err', np.abs(o.s - st[:, 2*nq:2*nq+ng]).max(), 'obs err', np.abs(o.observation() - env.reset().cpu().numpy()).max())
        T = 30
        acts = rng.uniform(-1.2, 1.2, (T, B, spec.n_null))
        worst = 0
        for t in range(T):
            # teacher forcing: oracle state -> device
            full = np.zeros((B, env.state_dim))
            full[:, :nq] = o.q; full[:, nq:2*nq] = o.dq; full[:, 2*nq:2*nq+ng] = o.s
            full[:, 2*nq+ng:2*nq+ng+6] = o.puck; full[:, -1] = o.t
            env.set_state(full)
            obs, r, ab, info = env.step(acts[t])
            oo, orr, oab, _ = o.step(acts[t])
            e = np.abs(obs.cpu().numpy() - oo).max(1)
            es = np.abs(env.get_state().cpu().numpy()[:, 2*nq:2*nq+ng] - o.s).max(1)
            er = np.abs(r.cpu().numpy() - orr)
            worst = max(worst, e.max(), es.max(), er.max())
            if t % 10 == 0:
                print('  t', t, 'obs', e.max(), 's', es.max(), 'rew', er.max(), 'n>1e-3', (np.maximum(e, es) > 1e-3).sum(), flush=True)
        print(name, dt, 'teacher-forced worst', worst, 'logs', env.get_constraints_logs(), o.get_constraints_logs(), flush=True)

# timing
for name in ['circle', 'planar', 'iiwa']:
    B = 8192
    env = BatchedAtacomEnv(name, B, device=dev, dtype=torch.float32, auto_reset=True)
    k = env.dims['null']
    a = torch.rand((B, k), device=dev) * 2 - 1
    for _ in range(5): env.step(a)
    torch.cuda.synchronize(); t0 = time.time()
    n = 50
    for _ in range(n): env.step_into(a, env._obs, env._reward, env._absorbing, env._last)
    torch.cuda.synchronize(); dt_ = (time.time() - t0) / n
    print(name, 'step B=8192: %.1f us -> %.3g env-steps/s' % (dt_ * 1e6, B / dt_), flush=True)
    T = 120
    acts = torch.rand((T, B, k), device=dev) * 2 - 1
    out = env.rollout(acts)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(3): env.rollout(acts, out=out)
    torch.cuda.synchronize(); dt_ = (time.time() - t0) / 3
    print(name, 'rollout T=120 B=8192: %.1f us/step -> %.3g env-steps/s' % (dt_ / T * 1e6, B * T / dt_), env.get_constraints_logs(), flush=True)
