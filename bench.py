#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the batched ATACOM step (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--env iiwa|planar|circle] [--batch B]

A "step" is ONE call of the hot path over one batch: `atacom_step` (C ABI, one HIP kernel launch) for
B = 8192 IiwaAirHockey-7H environments per GPU -- action clip/scale, 4 x [constraint Jacobians + FK,
null-space projection, slack integration, truncation, dynamics], reward / termination / observation,
constraint statistics, masked auto-reset at the horizon.  Inputs (actions) are resident in HBM before the
timed region.  N > 1: one process per GPU (torchrun), each rank owns its own 8192-env shard (weak
scaling, no collective on the data path); time = max over ranks between two barriers.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      HBM view: algorithmic bytes per launch / mean kernel time (HIP events on the launch stream)
  roofline_valu fp32 vector-ALU view of the same kernel (this workload is ALU/latency bound)
  cpu_baseline  the float64 oracle in the reference's algorithmic shape (one SVD + RREF per env per
                sub-step), timed on this box's host cores on a bounded sample (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
VALU_F32_PEAK_TF = 157.3       # MI355X_MICROARCH.md: peak FP32 vector

# algorithmic HBM bytes per env-step, fp32, sub-steps fused (SURVEY.md section 8d / BASELINE.md section 4;
# derivation in DESIGN.md "Measurement"): state read once + written once, action in, obs/reward/flags out.
ALGO_BYTES = {'circle': 60, 'planar': 220, 'iiwa': 400}
SHAPES = {'circle': (2, 3, 1, 2, 1), 'planar': (6, 9, 3, 3, 4), 'iiwa': (12, 17, 5, 6, 4)}   # c, n, k, nq, substeps


def algorithmic_flops(env):
    """FLOPs per env-step of the algorithm the kernel runs (FMA = 2), counted loop by loop
    (rl_on_manifold_amd/csrc/atacom_linalg.h); selects / compares / moves are not counted."""
    M, N, K, nq, sub = SHAPES[env]
    f = 0
    for i in range(M):
        f += 2 * (N - 1 - i) + 12 + (N - 1 - i)                       # norm, larfg, scale v
        if i < M - 1:
            f += (M - 1 - i) * (4 * (N - 1 - i) + 3)                  # G(i) on the rows below
            f += 2 * (M - 2 - i) + 12 + (M - 2 - i)                   # norm, larfg, scale u
            f += (N - 1 - i) * (4 * (M - 2 - i) + 3)                  # H(i) on the columns right
            f += 4 * (M - 2 - i) + 3                                  # H(i) on the rhs
    f += 3 * M                                                        # bidiagonal solve
    f += sum((1 + K) * (4 * (N - 1 - i) + 3) for i in range(M))       # P applied to [x | null]
    f += sum((N - 1 - j) * (1 + 2 * K) + 2 for j in range(K))         # rref, K pivots
    f += 2 * N * K + N                                                # Nc @ alpha - x
    f += 6 * M + 14 * nq                                              # rhs assembly, slack, truncation, integration
    per_sub = f
    fk = {'circle': 20, 'planar': 150, 'iiwa': 1500}[env]             # constraint terms + post-step kinematics
    return sub * per_sub + 2 * fk + 4 * M * nq


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=1000)
    ap.add_argument('--warmup', type=int, default=100)
    ap.add_argument('--env', default='iiwa')
    ap.add_argument('--batch', type=int, default=8192)
    ap.add_argument('--lanes', type=int, default=0, help='kernel mapping: 0 = library policy, 1 = env per lane, 4 = env per quad')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from rl_on_manifold_amd import BatchedAtacomEnv

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # BENCH_DIST_BACKEND=gloo lets the multi-rank control flow be exercised on a box with fewer GPUs than ranks
    # (ranks then share devices); the driver's runs use the default, nccl (= RCCL), one rank per GPU.
    backend = os.environ.get('BENCH_DIST_BACKEND', 'nccl')
    dev = torch.device('cuda', local_rank % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group(backend, rank=rank, world_size=world)
    red_dev = dev if backend == 'nccl' else torch.device('cpu')

    B, K, W = args.batch, args.steps, args.warmup
    env = BatchedAtacomEnv(args.env, B, device=dev, dtype=torch.float32, auto_reset=True, lanes_per_env=args.lanes)
    k, D = env.dims['null'], env.obs_dim
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    # synthetic data: per-env random initial joint states around the reset pose + a pool of pre-generated actions
    st = env.get_state()
    nq = env.dims['q']
    if args.env != 'circle':
        init = torch.zeros((B, env.init_state_dim), device=dev)
        init[:, :nq] = st[:, :nq] + 0.05 * torch.randn((B, nq), device=dev, generator=gen)
        init[:, 2 * nq:] = st[:, 2 * nq + env.dims['g']: 2 * nq + env.dims['g'] + 6]
        env.reset(state=init)
    n_pool = 64
    actions = torch.rand((n_pool, B, k), device=dev, generator=gen) * 2 - 1
    obs = torch.empty((B, D), device=dev)
    rew = torch.empty((B,), device=dev)
    ab = torch.empty((B,), device=dev, dtype=torch.uint8)
    last = torch.empty((B,), device=dev, dtype=torch.uint8)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(W):
        env.step_into(actions[i % n_pool], obs, rew, ab, last)
    sync_all()
    # ---- timed region: exactly K steps, bracketed by barrier + synchronize on both sides; two HIP events on the
    # launch stream bracket the same K launches (the queue never drains, so events/K = mean launch duration
    # including the ~1-2 us inter-kernel gap)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for i in range(K):
        env.step_into(actions[i % n_pool], obs, rew, ab, last)
    e1.record()
    sync_all()
    elapsed = time.perf_counter() - t0
    kern_ms = e0.elapsed_time(e1) / K
    # isolated per-launch durations (event pair around each launch, outside the timed region)
    n_iso = min(K, 200)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_iso)]
    for i in range(n_iso):
        ev[i][0].record()
        env.step_into(actions[i % n_pool], obs, rew, ab, last)
        ev[i][1].record()
    torch.cuda.synchronize(dev)
    kern_ms_iso = float(np.median([a.elapsed_time(b) for a, b in ev]))
    c_avg, c_max, c_dq_max = env.get_constraints_logs()
    if world > 1:
        t = torch.tensor([elapsed], device=red_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        cm = torch.tensor([c_max, c_dq_max], device=red_dev, dtype=torch.float64)
        dist.all_reduce(cm, op=dist.ReduceOp.MAX)
        c_max, c_dq_max = float(cm[0].item()), float(cm[1].item())

    # ---- secondary: the same K steps as launches of the multi-step rollout kernel (120 steps per launch)
    T = 120
    racts = torch.rand((T, B, k), device=dev, generator=gen) * 2 - 1
    out = env.rollout(racts)
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    n_roll = max(1, K // T)
    for _ in range(n_roll):
        env.rollout(racts, out=out)
    torch.cuda.synchronize(dev)
    roll_rate = n_roll * T * B / (time.perf_counter() - t1)

    # ---- secondary (row N2): the same rollout with the actor MLP (18-64-64-5, random weights) + Gaussian noise
    # evaluated inside the kernel -- one launch per 120-step collection phase, no per-step host round trip
    pol_rate = None
    if args.env != 'circle':
        from rl_on_manifold_amd import MlpPolicy
        gw = torch.Generator(device='cpu'); gw.manual_seed(0)
        Wts = [torch.randn(64, D, generator=gw) * 0.2, torch.zeros(64), torch.randn(64, 64, generator=gw) * 0.1,
             torch.zeros(64), torch.randn(k, 64, generator=gw) * 0.1, torch.zeros(k)]
        pol = MlpPolicy(*Wts, std=torch.full((k,), 0.5))
        eps = torch.randn((T, B, k), device=dev, generator=gen)
        env.rollout_policy(pol, T, noise=eps)
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        for _ in range(n_roll):
            env.rollout_policy(pol, T, noise=eps)
        torch.cuda.synchronize(dev)
        pol_rate = n_roll * T * B / (time.perf_counter() - t2)

    result = None
    if rank == 0:
        value = world * B * K / elapsed
        algo_bytes = ALGO_BYTES[args.env] * B
        flops = algorithmic_flops(args.env) * B
        achieved_gbs = algo_bytes / (kern_ms * 1e-3) / 1e9
        achieved_tf = flops / (kern_ms * 1e-3) / 1e12
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'traffic_%s.json' % args.env)
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get('hbm_bytes_per_launch')
            except Exception:  # noqa: BLE001
                traffic = None
        result = {
            'metric': 'env-steps/sec', 'value': value, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': K,
            'warmup': W, 'ms_per_step': elapsed / K * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': {'iiwa': 'IiwaAirHockey env 7H', 'planar': 'PlanarAirHockey env H',
                                    'circle': 'CircularMotion env A'}[args.env] + ', batch %d per GPU' % B,
                       'batch_per_gpu': B, 'global_batch': B * world, 'lanes_per_env_requested': int(env.cfg.lanes_per_env), 'substeps': int(env.cfg.substeps),
                       'horizon': int(env.cfg.horizon), 'path': 'atacom_step (1 launch / step) via C ABI',
                       'parallelism': 'env-shard x%d, no data-path collective' % world},
            'max_abs_c': c_max, 'c_avg': c_avg, 'c_dq_max': c_dq_max,
            'rollout_kernel_env_steps_per_s_per_gpu': roll_rate,
            'policy_rollout_kernel_env_steps_per_s_per_gpu': pol_rate,
            'roofline': {'bound': 'hbm', 'achieved': achieved_gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved_gbs / HBM_PEAK_GBS, 'traffic': traffic,
                         'algorithmic_bytes_per_launch': algo_bytes, 'kernel_ms': kern_ms,
                         'kernel_ms_isolated_median': kern_ms_iso,
                         'note': 'workload is fp32-VALU / latency bound, see roofline_valu and DESIGN.md'},
            'roofline_valu': {'bound': 'valu_f32', 'achieved': achieved_tf, 'peak': VALU_F32_PEAK_TF,
                              'unit': 'TFLOP/s', 'frac': achieved_tf / VALU_F32_PEAK_TF,
                              'algorithmic_flops_per_launch': flops},
        }
        if world == 1 and not args.no_cpu_baseline:
            result['cpu_baseline'] = cpu_baseline(args.env, args.cpu_seconds)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return result


def cpu_baseline(env_name, budget_s):
    """The oracle as the CPU baseline ("port"): float64, reference algorithmic shape -- one LAPACK SVD and one
    RREF per environment per physics sub-step, one environment at a time, one core."""
    import numpy as np
    from oracle import atacom_scalar as osc
    spec = {'circle': osc.circle_spec, 'planar': osc.planar_spec, 'iiwa': osc.iiwa_spec}[env_name]()
    init_q = None
    if env_name == 'iiwa':
        init_q = np.array([0.0, 0.7135214629060707, 0.0, -0.5024756033561426, 0.0, 1.9256631778550268])
    rng = np.random.default_rng(0)
    env = osc.ScalarAtacomEnv(spec, init_q=init_q)
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        for _ in range(8):
            env.step(rng.uniform(-1, 1, spec.n_null))
            n += 1
            if env.t >= spec.horizon:
                env.reset()
    dt = time.perf_counter() - t0
    return {'value': n / dt, 'unit': 'env-steps/s', 'cores': 1, 'kind': 'port',
            'sample': '%d sequential env-steps of one %s env (float64 numpy/scipy oracle, %.1f s)' % (n, env_name, dt),
            'host_cpus_visible': os.cpu_count()}


if __name__ == '__main__':
    main()
