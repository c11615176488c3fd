"""Differential soak (not a pytest file): f32 kernels of all four mappings (lane, pair, quad, 8 lanes) vs the f64 lane kernel,
teacher-forced from the f64 state, 16384 envs x 200 steps per env type.  Results: profiles/r01_parity_soak.md,
profiles/r02_parity_soak.md."""
import sys, torch
sys.path.insert(0, '.')
from rl_on_manifold_amd import BatchedAtacomEnv
dev = 'cuda:0'
for name in ('iiwa', 'planar'):
    B = 16384
    e1 = BatchedAtacomEnv(name, B, device=dev, dtype=torch.float32, lanes_per_env=1, auto_reset=True, random_init=True, seed=3)
    e4 = BatchedAtacomEnv(name, B, device=dev, dtype=torch.float32, lanes_per_env=4, auto_reset=True, random_init=True, seed=3)
    e2 = BatchedAtacomEnv(name, B, device=dev, dtype=torch.float32, lanes_per_env=2, auto_reset=True, random_init=True, seed=3)
    eo = BatchedAtacomEnv(name, B, device=dev, dtype=torch.float32, lanes_per_env=8, auto_reset=True, random_init=True, seed=3)
    e8 = BatchedAtacomEnv(name, B, device=dev, dtype=torch.float64, lanes_per_env=1, auto_reset=True, random_init=True, seed=3)
    k = e1.dims['null']
    gen = torch.Generator(device=dev); gen.manual_seed(0)
    st = e1.get_state(); nq = e1.dims['q']
    init = torch.zeros((B, e1.init_state_dim), device=dev)
    init[:, :nq] = st[:, :nq] + 0.05 * torch.randn((B, nq), device=dev, generator=gen)
    init[:, 2 * nq:] = st[:, 2 * nq + e1.dims['g']: 2 * nq + e1.dims['g'] + 6]
    for e in (e1, e2, e4, eo): e.reset(state=init)
    e8.reset(state=init.double())
    tot = 0; bad14 = 0; bad18 = 0; bad48 = 0; bad28 = 0; bado8 = 0; mx = 0.0
    for t in range(200):
        a = torch.rand((B, k), device=dev, generator=gen) * 2.4 - 1.2
        s8 = e8.get_state()
        e1.set_state(s8.float()); e4.set_state(s8.float()); e2.set_state(s8.float()); eo.set_state(s8.float())
        o1 = e1.step(a)[0]; o4 = e4.step(a)[0]; o2 = e2.step(a)[0]; o8 = e8.step(a.double())[0].float()
        d14 = (o1 - o4).abs().amax(1); d18 = (o1 - o8).abs().amax(1); d48 = (o4 - o8).abs().amax(1)
        bad28 += int(((o2 - o8).abs().amax(1) > 2e-3).sum())
        oo = eo.step(a)[0]
        bado8 += int(((oo - o8).abs().amax(1) > 2e-3).sum())
        assert torch.isfinite(oo).all() and torch.isfinite(o2).all()
        tot += B; bad14 += int((d14 > 2e-3).sum()); bad18 += int((d18 > 2e-3).sum()); bad48 += int((d48 > 2e-3).sum())
        assert torch.isfinite(o1).all() and torch.isfinite(o4).all()
    print('%s: %d env-steps teacher-forced from the f64 lane kernel; > 2e-3 obs error: f32 lane vs f32 quad %.4f %%, f32 lane vs f64 %.4f %%, f32 quad vs f64 %.4f %%, f32 pair vs f64 %.4f %%, f32 8 lanes vs f64 %.4f %%; median |quad - f64| %.2e'
          % (name, tot, 100.0 * bad14 / tot, 100.0 * bad18 / tot, 100.0 * bad48 / tot, 100.0 * bad28 / tot, 100.0 * bado8 / tot, float(d48.median())), flush=True)
