"""CircularMotion: what the T-step kernel gives where atacom_step is launch-bound (run from the repo root on the GPU box)."""
import sys
import torch
sys.path.insert(0, '.')
from rl_on_manifold_amd import BatchedAtacomEnv
dev = 'cuda:0'
for B, T in ((4096, 500), (4096, 120), (65536, 500), (1048576, 120), (4194304, 60)):
    env = BatchedAtacomEnv('circle', B, device=dev, dtype=torch.float32, auto_reset=True)
    g = torch.Generator(device=dev); g.manual_seed(0)
    acts = torch.rand((T, B, 1), device=dev, generator=g) * 2 - 1
    out = env.rollout(acts)
    nbytes = acts.numel() * 4 + sum(v.numel() * v.element_size() for v in out.values() if torch.is_tensor(v))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        env.rollout(acts, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print('circle B=%d T=%d: %.3f ms per rollout, %.2f us per step, %.3g env-steps/s, %d B per env-step -> %.1f GB/s = %.3f of 8 TB/s'
          % (B, T, ms, ms / T * 1e3, B * T / ms * 1e3, nbytes // (B * T), nbytes / ms / 1e6, nbytes / ms / 1e6 / 8000), flush=True)
