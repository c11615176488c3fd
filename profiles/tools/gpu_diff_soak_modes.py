"""Differential soak over modes (not a pytest file): f64 quad vs f64 lane, and f32 quad / f32 lane vs f64 lane, teacher-forced,
for the refresh / exact-bias variants and the circle env.  Result of the round: profiles/r01_parity_soak.md."""
import sys, torch
sys.path.insert(0, '.')
from rl_on_manifold_amd import BatchedAtacomEnv
dev = 'cuda:0'
def run(name, B, T, kw, dt64_quad=False):
    e1 = BatchedAtacomEnv(name, B, device=dev, dtype=torch.float64, lanes_per_env=1, auto_reset=True, random_init=True, seed=3, **kw)
    e4 = BatchedAtacomEnv(name, B, device=dev, dtype=torch.float64, lanes_per_env=4, auto_reset=True, random_init=True, seed=3, **kw)
    f4 = BatchedAtacomEnv(name, B, device=dev, dtype=torch.float32, lanes_per_env=4, auto_reset=True, random_init=True, seed=3, **kw)
    f1 = BatchedAtacomEnv(name, B, device=dev, dtype=torch.float32, lanes_per_env=1, auto_reset=True, random_init=True, seed=3, **kw)
    k = e1.dims['null'] if not isinstance(e1.dims['null'], tuple) else e1.dims['null'][0]
    k = e1.action_dim if hasattr(e1, 'action_dim') else k
    gen = torch.Generator(device=dev); gen.manual_seed(0)
    mx64 = 0.0; bad4 = bad1 = tot = 0
    for t in range(T):
        a = torch.rand((B, k), device=dev, generator=gen, dtype=torch.float64) * 2.4 - 1.2
        s = e1.get_state()
        e4.set_state(s); f4.set_state(s.float()); f1.set_state(s.float())
        o1 = e1.step(a)[0]; o4 = e4.step(a)[0]; p4 = f4.step(a.float())[0].double(); p1 = f1.step(a.float())[0].double()
        assert torch.isfinite(o1).all() and torch.isfinite(p4).all()
        mx64 = max(mx64, float((o1 - o4).abs().max()))
        tot += B; bad4 += int(((p4 - o1).abs().amax(1) > 2e-3).sum()); bad1 += int(((p1 - o1).abs().amax(1) > 2e-3).sum())
    print('%-10s %-34s %8d env-steps: max |f64 quad - f64 lane| %.2e; f32 quad / f32 lane beyond 2e-3: %.4f %% / %.4f %%'
          % (name, str(kw), tot, mx64, 100.0 * bad4 / tot, 100.0 * bad1 / tot), flush=True)
run('iiwa', 4096, 60, {})
run('iiwa', 4096, 60, {'hold_q': False})
run('iiwa', 4096, 60, {'bias_mode': 'exact'})
run('planar', 4096, 60, {'hold_q': False, 'bias_mode': 'exact'})
run('circle', 4096, 200, {})
