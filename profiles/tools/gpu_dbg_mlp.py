import sys, torch
sys.path.insert(0, '.')
from rl_on_manifold_amd import BatchedAtacomEnv, MlpPolicy
DEV='cuda:0'
B, T = 160, 4
gw = torch.Generator().manual_seed(1)
W = [torch.randn(64, 18, generator=gw) * 0.2, torch.randn(64, generator=gw) * 0.1, torch.randn(64, 64, generator=gw) * 0.1,
     torch.randn(64, generator=gw) * 0.1, torch.randn(5, 64, generator=gw) * 0.1, torch.zeros(5)]
pol = MlpPolicy(*W, std=torch.full((5,), 0.3))
Wd = [w.to(DEV) for w in W]
for mode in ('kinematic', 'rigid_body'):
  for lanes in (4, 1):
    env = BatchedAtacomEnv('iiwa', B, device=DEV, dynamics_mode=mode, lanes_per_env=lanes)
    g = torch.Generator(device=DEV).manual_seed(4)
    eps = torch.randn((T, B, 5), device=DEV, generator=g)
    out = env.rollout_policy(pol, T, noise=eps)
    for t in range(T):
        h = torch.relu(out['obs'][t] @ Wd[0].T + Wd[1]); h = torch.relu(h @ Wd[2].T + Wd[3])
        a = h @ Wd[4].T + Wd[5] + 0.3 * eps[t]
        d = (a - out['action'][t]).abs().amax(1)
        bad = torch.nonzero(d > 2e-4).flatten().tolist()
        print(mode, 'lanes', lanes, 't', t, 'bad envs', len(bad), bad[:24], 'max', float(d.max()))
    # is next_obs[t] == obs[t+1]?
    print('   obs chain consistent:', [bool(torch.equal(out['next_obs'][t], out['obs'][t+1])) for t in range(T-1)])

# hypotheses for the quad rigid-body kernel: which observation did the in-kernel network see at t = 1?
env = BatchedAtacomEnv('iiwa', B, device=DEV, dynamics_mode='rigid_body', lanes_per_env=4)
g = torch.Generator(device=DEV).manual_seed(4)
eps = torch.randn((T, B, 5), device=DEV, generator=g)
out = env.rollout_policy(pol, T, noise=eps)
def net(o):
    h = torch.relu(o @ Wd[0].T + Wd[1]); h = torch.relu(h @ Wd[2].T + Wd[3]); return h @ Wd[4].T + Wd[5]
t = 1
want = out['action'][t] - 0.3 * eps[t]
idx = torch.arange(B, device=DEV)
for name, o in (('obs[t]', out['obs'][t]), ('obs[t-1]', out['obs'][t - 1]), ('obs[t] of env & ~3', out['obs'][t][idx & ~3]),
                ('obs[t] of env | 3', out['obs'][t][(idx | 3).clamp(max=B - 1)])):
    d = (net(o) - want).abs().amax(1)
    print('%-22s envs matching: %d of %d' % (name, int((d < 2e-4).sum()), B))
# solve for the observation the network must have seen? compare per-component: which obs entries, if replaced, explain it
o1 = out['obs'][t].clone()
for comp in range(18):
    o2 = o1.clone(); o2[:, comp] = out['obs'][t - 1][:, comp]
    d = (net(o2) - want).abs().amax(1)
    print('  component %2d from t-1: matching %d' % (comp, int((d < 2e-4).sum())), end=';')
print()

# chunk-level hypothesis: some 4-float chunks of the staged observation row are stale (from t - 1)
import itertools
best = []
for mask in range(32):
    o2 = out['obs'][t].clone()
    for ch in range(5):
        if mask >> ch & 1:
            lo, hi = 4 * ch, min(4 * ch + 4, 18)
            o2[:, lo:hi] = out['obs'][t - 1][:, lo:hi]
    d = (net(o2) - want).abs().amax(1)
    best.append((int((d < 2e-4).sum()), mask))
best.sort(reverse=True)
print('stale-chunk hypotheses (matching envs, chunk mask):', best[:6])
