import sys, numpy as np, torch
sys.path.insert(0, '.')
from rl_on_manifold_amd import BatchedAtacomEnv, constraint_terms
from oracle import atacom_scalar as osc, atacom_batched as ob
spec = osc.iiwa_spec()
rng = np.random.default_rng(3)
n = 777
q = rng.uniform(-1.5, 1.5, (n, 6)); dq = rng.uniform(-1.5, 1.5, (n, 6))
f, J, b = constraint_terms('iiwa', torch.tensor(q, device='cuda:0'), torch.tensor(dq, device='cuda:0'))
fo, Jo, bo = ob.constraint_terms(spec, q, dq)
Jd = J.cpu().numpy()
mism = ((Jd == 0) != (Jo == 0))
idx = np.argwhere(mism)
print('n mismatch', len(idx), 'first', idx[:10])
for (s_, r, c) in idx[:5]:
    print(s_, r, c, 'dev', Jd[s_, r, c], 'oracle', Jo[s_, r, c])
print(np.unique(idx[:, 1:], axis=0))
# full-size stats f32 vs f64
for dt in (torch.float32, torch.float64):
    B, T = 8192, 120
    env = BatchedAtacomEnv('iiwa', B, dtype=dt)
    gen = torch.Generator(device='cuda:0'); gen.manual_seed(0)
    st = env.get_state()
    init = torch.zeros((B, env.init_state_dim), device='cuda:0', dtype=dt)
    init[:, :6] = st[:, :6] + (0.05 * torch.randn((B, 6), device='cuda:0', generator=gen)).to(dt)
    init[:, 12:] = st[:, 23:29]
    env.reset(state=init)
    acts = (torch.rand((T, B, 5), device='cuda:0', generator=gen) * 2 - 1).to(dt)
    out = env.rollout(acts)
    print(dt, env.get_constraints_logs(), 'absorbing frac', out['absorbing'].float().mean().item())
