"""Endurance run (not a pytest file): free-running environments with random actions, auto-reset and randomised resets for
hundreds of thousands of steps; every CHECK steps the whole persistent state must be finite and the constraint statistics
are printed.  python profiles/tools/gpu_long_soak.py [STEPS] [CHECK]      (SOAK_DTYPE=f64: the float64 build)"""
import os, sys, time
import torch
sys.path.insert(0, '.')
from rl_on_manifold_amd import BatchedAtacomEnv
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
CHECK = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
dev = 'cuda:0'
DT = {'f32': torch.float32, 'f64': torch.float64}[os.environ.get('SOAK_DTYPE', 'f32')]
CASES = [('circle', 4096, {}), ('planar', 8192, {}), ('iiwa', 8192, {}), ('iiwa', 8192, {'chart_mode': 'canonical'}),
         ('planar', 8192, {'chart_mode': 'canonical'}), ('iiwa', 8192, {'obs_noise': True, 'obs_delay': True, 'env_noise': True}),
         ('iiwa', 8192, {'dynamics_mode': 'rigid_body_ff'})]
for name, B, kw in CASES:
    env = BatchedAtacomEnv(name, B, device=dev, dtype=DT, auto_reset=True, random_init=True, seed=7, **kw)
    k = env.dims['null']
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    acts = (torch.rand((256, B, k), device=dev, generator=gen) * 2 - 1).to(DT)
    acts[::7] = torch.sign(acts[::7])                       # bang-bang actions now and then
    obs = torch.empty((B, env.obs_dim), device=dev, dtype=DT); rew = torch.empty((B,), device=dev, dtype=DT)
    ab = torch.empty((B,), device=dev, dtype=torch.uint8); la = torch.empty((B,), device=dev, dtype=torch.uint8)
    steppers = [env.bind_step(acts[i], obs, rew, ab, la) for i in range(256)]
    n_steps = STEPS if 'dynamics_mode' not in kw else STEPS // 2
    t0 = time.time(); worst = [-1e9, -1e9]; episodes = 0; bad = 0
    for it in range(n_steps):
        steppers[it % 256]()
        if (it + 1) % CHECK == 0:
            st = env.get_state()
            fin = bool(torch.isfinite(st).all()) and bool(torch.isfinite(obs).all()) and bool(torch.isfinite(rew).all())
            c_avg, c_max, c_dq = env.get_constraints_logs()
            worst = [max(worst[0], c_max), max(worst[1], c_dq)]
            bad += (not fin)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print('%-7s %-62s %8d steps x %d envs = %.2e env-steps in %5.1f s: state finite at every check: %s; worst c_max %.4f, worst c_dq_max %.2e'
          % (name, str(kw), n_steps, B, n_steps * B, dt, bad == 0, worst[0], worst[1]), flush=True)
    env.close()
