"""GPU tests of the collection path: the packed-record rollout kernels, the RolloutCollector
over the real HIP engine (world 1, and two processes sharing one GPU), and bench.py's own rank spawning."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = 'cuda:0'
KEYS = ('obs', 'action', 'reward', 'next_obs', 'absorbing', 'last')


def _env(name, B, **kw):
    from rl_on_manifold_amd import BatchedAtacomEnv
    return BatchedAtacomEnv(name, B, device=DEV, dtype=torch.float32, auto_reset=True, **kw)


def _policy(D, k, seed=0):
    from rl_on_manifold_amd import MlpPolicy
    g = torch.Generator().manual_seed(seed)
    return MlpPolicy(torch.randn(64, D, generator=g) * 0.2, torch.randn(64, generator=g) * 0.1,
                     torch.randn(64, 64, generator=g) * 0.1, torch.randn(64, generator=g) * 0.1,
                     torch.randn(k, 64, generator=g) * 0.1, torch.zeros(k), std=torch.full((k,), 0.3))


@pytest.mark.parametrize('lanes', [1, 2, 4, 8])
@pytest.mark.parametrize('name', ['circle', 'planar', 'iiwa'])
def test_packed_rollout_is_bitwise_the_array_rollout(name, lanes):
    """atacom_rollout_packed writes exactly what atacom_rollout writes, as one record per (step, env); a padded env
    axis (ragged shards) leaves the padding rows untouched."""
    B, T = 333, 17
    env = _env(name, B, lanes_per_env=lanes, horizon=7)
    k = env.dims['null']
    g = torch.Generator(device=DEV).manual_seed(1)
    acts = torch.rand((T, B, k), device=DEV, generator=g) * 2.4 - 1.2
    st = env.get_state()
    ref = env.rollout(acts)
    env.set_state(st)
    rec = env.rollout_packed(actions=acts, batch_stride=B + 5)
    assert rec.shape == (T, B + 5, env.record_dim)
    assert (rec[:, B:] == 0).all()
    got = env.unpack_records(rec[:, :B])
    for key in KEYS:
        a, b = got[key], ref[key]
        assert torch.equal(a.float(), b.float()), key
    assert got['last'][6].all()                                        # horizon 7, auto-reset


@pytest.mark.parametrize('name', ['planar', 'iiwa'])
def test_packed_policy_rollout_is_bitwise_the_array_policy_rollout(name):
    B, T = 200, 9
    env = _env(name, B)
    pol = _policy(env.obs_dim, env.dims['null'])
    g = torch.Generator(device=DEV).manual_seed(2)
    eps = torch.randn((T, B, env.dims['null']), device=DEV, generator=g)
    st = env.get_state()
    ref = env.rollout_policy(pol, T, noise=eps)
    env.set_state(st)
    got = env.unpack_records(env.rollout_packed(policy=pol, n_steps=T, noise=eps))
    for key in KEYS:
        assert torch.equal(got[key].float(), ref[key].float()), key


def test_collector_over_the_hip_engine_world_1():
    """RolloutCollector driving a real BatchedAtacomEnv: fused actions path, fused policy path, host-policy path."""
    from rl_on_manifold_amd.rollout import RolloutCollector, to_mushroom_dataset
    B, T = 96, 11
    env = _env('iiwa', B, horizon=5)
    col = RolloutCollector(env)
    k = env.dims['null']
    g = torch.Generator(device=DEV).manual_seed(3)
    acts = torch.rand((T, B, k), device=DEV, generator=g) * 2 - 1
    st = env.get_state()
    ref = env.rollout(acts)
    env.set_state(st)
    data = col.collect(T, actions=acts)
    assert data['obs'].shape == (1, T, B, env.obs_dim) and data['obs'].is_cuda
    assert data['absorbing'].dtype == torch.bool and data['last'].dtype == torch.bool
    tm = col.time_major(data)
    for key in KEYS:
        assert torch.equal(tm[key].float(), ref[key].float()), key
    ds = to_mushroom_dataset(tm)
    assert len(ds) == T * B and ds[4][5] is True
    # fused policy
    pol = _policy(env.obs_dim, k)
    data = col.collect(T, policy=pol, noise=torch.randn((T, B, k), device=DEV, generator=g))
    assert data['action'].shape == (1, T, B, k) and torch.isfinite(data['reward']).all()
    # host-side policy (one launch per step)
    data = col.collect(T, policy=lambda o: torch.zeros((B, k), device=DEV))
    assert data['obs'].shape == (1, T, B, env.obs_dim)
    c_avg, c_max, c_dq = col.get_constraints_logs(n_logged=3 * T * B)
    assert np.isfinite([c_avg, c_max, c_dq]).all()


def _worker(rank, world, port, gb, T, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from rl_on_manifold_amd import BatchedAtacomEnv
        from rl_on_manifold_amd.rollout import RolloutCollector, shard_bounds
        lo, hi = shard_bounds(gb, world, rank)
        env = BatchedAtacomEnv('planar', hi - lo, device=DEV, dtype=torch.float32, auto_reset=True, horizon=6)
        g = torch.Generator().manual_seed(9)
        acts = (torch.rand((T, gb, 3), generator=g) * 2 - 1)[:, lo:hi].to(DEV)
        init = torch.zeros((gb, env.init_state_dim))
        init[:, :3] = torch.tensor([-0.9273, 0.9273, np.pi / 2]) + 0.03 * torch.randn((gb, 3), generator=g)
        init[:, 6] = -0.5
        env.reset(state=init[lo:hi].to(DEV))
        col = RolloutCollector(env, global_batch=gb)
        data = col.time_major(col.collect(T, actions=acts))
        stats = col.get_constraints_logs(n_logged=T * (hi - lo))
        q.put((rank, {k_: v.cpu().numpy() for k_, v in data.items()}, stats))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_collector_two_processes_sharing_the_gpu_equal_one_process():
    """Two ranks (ragged shards 6 + 5), each with its own HIP engine on the same GPU, gloo as the transport (RCCL
    refuses two ranks on one device): the gathered global dataset equals a single-process run."""
    import torch.multiprocessing as mp
    gb, T, world = 11, 13, 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29800 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, gb, T, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from rl_on_manifold_amd.rollout import RolloutCollector
    env = _env('planar', gb, horizon=6)
    g = torch.Generator().manual_seed(9)
    acts = (torch.rand((T, gb, 3), generator=g) * 2 - 1).to(DEV)
    init = torch.zeros((gb, env.init_state_dim))
    init[:, :3] = torch.tensor([-0.9273, 0.9273, np.pi / 2]) + 0.03 * torch.randn((gb, 3), generator=g)
    init[:, 6] = -0.5
    env.reset(state=init.to(DEV))
    col = RolloutCollector(env)
    ref = col.time_major(col.collect(T, actions=acts))
    ref_stats = env.get_constraints_logs()
    for rank, data, stats in results:
        for key in KEYS:
            assert data[key].shape == tuple(ref[key].shape), key
            assert np.array_equal(data[key], ref[key].cpu().numpy()), (rank, key)
        assert np.allclose(stats, ref_stats, rtol=1e-6, atol=1e-7)


# ---------------------------------------------------------------------------------------------- BASELINE config 5
def _config5_shard(r, B, T):
    """Shard r of config 5 exactly as bench.py builds a rank's workload: feasible initial states,
    auto-reset at the horizon, uniform actions."""
    sys.path.insert(0, ROOT)
    import bench
    gen = torch.Generator(device=DEV)
    gen.manual_seed(1234 + r)
    env, init, _ = bench.make_env('iiwa', B, torch.device(DEV), gen)
    acts = torch.rand((T, B, 5), device=DEV, generator=gen) * 2 - 1
    return env, init, acts


def _explain_cmax(env, init, acts, c_reported):
    """Replay a shard step by step, find the (step, environment) of its largest constraint value, and teacher-force the float64
    oracle through the six steps up to it from the float32 device's OWN states: same violation (1e-4), same joint state."""
    from rl_on_manifold_amd import constraint_terms
    from oracle import atacom_scalar as osc, atacom_batched as ob
    T, B = acts.shape[0], env.batch

    def cvals(q):
        fun, _, _ = constraint_terms('iiwa', q, torch.zeros_like(q))
        return torch.maximum(fun[:, 0].abs(), fun[:, 1:].max(1).values)
    env.reset(state=init)
    env.get_constraints_logs()
    states, cs = [], []
    for t in range(T):
        states.append(env.get_state().clone())
        env.step(acts[t])
        cs.append(cvals(env.get_state()[:, :6]))
    states.append(env.get_state().clone())
    cs = torch.stack(cs)
    assert abs(float(cs.max()) - c_reported) < 1e-6          # single steps == the rollout kernel, and c recomputed from the states
    flat = int(cs.argmax())
    t_star, b = flat // B, flat % B
    spec = osc.iiwa_spec()
    nq, ng = 6, 11
    for t in range(max(0, t_star - 5), t_star + 1):
        st = states[t][b:b + 1].double().cpu().numpy()
        if int(st[0, -1]) == 0 and t > 0:
            continue                                          # an auto-reset in between: the next state is not this step's
        o = ob.BatchedAtacomEnv(spec, 1, init_q=st[:, :nq])
        o.set_state(st[:, :nq], st[:, nq:2 * nq], st[:, 2 * nq:2 * nq + ng], st[:, 2 * nq + ng:2 * nq + ng + 6])
        o.t[:] = int(st[0, -1])
        o.step(acts[t][b:b + 1].double().cpu().numpy())
        q_dev = states[t + 1][b:b + 1, :nq]
        c_dev, c_orc = float(cvals(q_dev)), float(cvals(torch.tensor(o.q, device=DEV, dtype=torch.float32)))
        assert abs(c_dev - c_orc) < 1e-4 and float((q_dev.double().cpu() - torch.tensor(o.q)).abs().max()) < 1e-4, (t, b, c_dev, c_orc)


P999_BOUND, NABOVE_BOUND = 0.018, 8           # measured (profiles/r06_config5_c_statistics.log): 0.0091 and 1 of 65536


def test_config5_dress_rehearsal_eight_shards_on_one_gpu():
    """BASELINE config 5 at full size, everything but the xGMI hop: 8 engines x 8192 IiwaAirHockey environments on ONE
    device, each rolled out for the full 120-step horizon by one launch straight into ITS block of the final
    [8, 120, 8192, 44] buffer (1.38 GB) -- the layout the all-gather produces on every rank."""
    from rl_on_manifold_amd import RecordLayout
    W, B, T = 8, 8192, 120
    shards = [_config5_shard(r, B, T) for r in range(W)]
    env0 = shards[0][0]
    F = env0.record_dim
    assert F == 44
    buf = torch.empty((W, T, B, F), device=DEV)
    for r, (env, _, acts) in enumerate(shards):
        env.get_constraints_logs()
        env.rollout_packed(actions=acts, out=buf[r])
    torch.cuda.synchronize()
    stats = np.array([env.get_constraints_logs() for env, _, _ in shards])
    assert bool(torch.isfinite(buf).all())
    lay = RecordLayout([B] * W, env0.obs_dim, 5)
    data = lay.unpack(buf)
    # episodes end at the horizon -- step 120 of an environment that never hit an absorbing state -- or earlier by
    # absorbing (then the auto-reset restarts its step counter); absorbing implies last
    quiet = ~data['absorbing'].any(1)                          # [W, B]
    assert float(quiet.float().mean()) > 0.5
    assert data['last'][:, -1][quiet].all() and not data['last'][:, :-1].permute(0, 2, 1)[quiet].any()
    assert not (data['absorbing'] & ~data['last']).any()
    # the constraint metric of the whole 7.9 M env-steps, from feasible initial states.  c_max is a heavy-tailed statistic of
    # the REFERENCE ALGORITHM (its rref tolerance branch zeroes basis entries and leaks constraint error): typical
    # 0.010, single environments up to 0.03 - 0.06 depending on how the float32 rounding falls (profiles/r03_chart_closed_loop.log,
    # r05_cmax_probe.log).  So: below 0.02 in most shards, and every shard above 0.05 must be EXPLAINED -- the float64 oracle,
    # teacher-forced from the device's own states, produces the same violation step for step.
    assert np.median(stats[:, 1]) < 0.02 and stats[:, 1].max() < 0.15, stats
    assert stats[:, 2].max() <= 1e-4, stats
    # ... and a tight bound on ROBUST statistics of the same quantity (a shard maximum cannot see a regression that
    # lifts a minority of environments into the 0.02 - 0.05 band): the largest constraint value of every ENVIRONMENT over its 120
    # steps, recomputed from the recorded joint positions -- its 99.9th percentile over the 65536 environments and the number of
    # environments above 0.03 (measured round 6: median 0.0038, p99.9 0.0091, max 0.0552, 1 environment above 0.03; bounds at 2 x / 8)
    from rl_on_manifold_amd import constraint_terms
    per_env = []
    for r in range(W):
        q = data['next_obs'][r][..., 6:12].reshape(-1, 6).contiguous()
        fun, _, _ = constraint_terms('iiwa', q, torch.zeros_like(q))
        c = torch.maximum(fun[:, 0].abs(), fun[:, 1:].max(1).values).reshape(T, B)
        assert abs(float(c.max()) - stats[r, 1]) < 1e-6, (r, float(c.max()), stats[r, 1])      # the same quantity the kernel logged
        per_env.append(c.max(0).values)
    per_env = torch.cat(per_env)
    p999, n_above = float(torch.quantile(per_env, 0.999)), int((per_env > 0.03).sum())
    print('config 5 per-environment max c: median %.4f p99.9 %.4f max %.4f, %d of %d environments above 0.03'
          % (float(per_env.median()), p999, float(per_env.max()), n_above, per_env.numel()))
    assert p999 < P999_BOUND and n_above <= NABOVE_BOUND, (p999, n_above)
    # the worst shard is always replayed through the float64 oracle from the device's own states, the others above 0.05 too
    worst = int(np.argmax(stats[:, 1]))
    for r in sorted(set([worst]) | set(np.nonzero(stats[:, 1] >= 0.05)[0].tolist())):
        _explain_cmax(*shards[r], float(stats[r, 1]))
    # (i) the shard-major buffer, time-majored, IS the single-engine array rollout of every shard
    tm = lay.time_major(data)
    assert tm['obs'].shape == (T, W * B, env0.obs_dim)
    for r in (0, 3, 7):
        env, init, acts = shards[r]
        env.reset(state=init)
        ref = env.rollout(acts)
        for key in KEYS:
            assert torch.equal(tm[key][:, r * B:(r + 1) * B].float(), ref[key].float()), (r, key)
    # (ii) ONE 65536-environment engine (one environment per lane instead of per lane group) on the concatenated states
    # and actions: same episodes.  The two mappings round differently and the closed loop amplifies that (DESIGN.md
    # section 2), so: first step to float32 tolerance, the later ones in distribution.
    from rl_on_manifold_amd import BatchedAtacomEnv
    big = BatchedAtacomEnv('iiwa', W * B, device=DEV, dtype=torch.float32, auto_reset=True)
    assert big.rollout_lanes_per_env == 1 and env0.rollout_lanes_per_env == 8
    big.reset(state=torch.cat([s[1] for s in shards]))
    sub = slice(0, None, 61)                                    # a subsample of the 65536 environments
    ref = big.rollout(torch.cat([s[2] for s in shards], 1))
    big_stats = big.get_constraints_logs()
    d0 = (ref['next_obs'][0] - tm['next_obs'][0]).abs().max(1).values
    assert float((d0 < 2e-4).float().mean()) > 0.995 and float(d0.median()) < 2e-6
    dm = (ref['next_obs'][:, sub] - tm['next_obs'][:, sub]).abs().amax(2)
    assert float(dm[:10].median()) < 1e-5
    # (episode ends by absorbing -- puck events -- are decisions the diverging trajectories may take differently)
    assert float((ref['last'][-1].bool() == tm['last'][-1]).float().mean()) > 0.98
    # (c_max is heavy-tailed, see above: the one-lane mapping's maximum lies between half the shards' typical value and twice
    # their largest)
    assert 0.5 * np.median(stats[:, 1]) < big_stats[1] < 2.0 * stats[:, 1].max() and big_stats[2] <= 1e-4
    big.close()
    for env, _, _ in shards:
        env.close()


def _rccl_world1_worker(port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import torch.distributed as dist
    dev = torch.device(DEV)
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    try:
        from rl_on_manifold_amd.rollout import RolloutCollector
        B, T = 8192, 120
        env, _, acts = _config5_shard(0, B, T)
        col = RolloutCollector(env, force_collective=True)
        assert dist.get_backend() == 'nccl'
        rec = col.collect_local(T, actions=acts)
        assert rec.is_cuda and rec.numel() * 4 == 173015040          # the real 173 MB record buffer of one rank
        g = col.gather(rec)                                           # blocking all_gather_into_tensor on device tensors
        torch.cuda.synchronize()
        ok = [g.is_cuda and tuple(g.shape) == (1, T, B, 44) and g.data_ptr() != rec.data_ptr() and torch.equal(g[0], rec)]
        out = torch.full((1, T, B, 44), float('nan'), device=dev)
        g2, work = col.gather(rec, out=out, async_op=True)            # async_op on RCCL's stream + work.wait()
        work.wait()
        ok.append(g2.data_ptr() == out.data_ptr() and torch.equal(out[0], rec))
        # collect_async: rollout kernel -> all-gather in flight -> a second rollout on the compute stream meanwhile
        env2, _, acts2 = _config5_shard(0, B, T)
        col2 = RolloutCollector(env2, force_collective=True)
        out2 = torch.empty((1, T, B, 44), device=dev)
        pend = col2.collect_async(T, actions=acts2, out=out2)
        rec_b = env.rollout_packed(actions=acts)                      # overlaps the collective
        data = pend.wait()
        torch.cuda.synchronize()
        ok.append(torch.equal(out2[0], rec))                          # same seed, same states: the same records
        ok.append(bool(torch.isfinite(rec_b).all()) and data['obs'].shape == (1, T, B, 18))
        stats = col.get_constraints_logs(n_logged=2 * T * B)          # MAX / SUM all-reduces over RCCL
        ok.append(bool(np.isfinite(stats).all()))
        # timing of the self-gather: a lower bound on what the collective costs per collection
        import time
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            col.gather(rec)
        torch.cuda.synchronize()
        q.put((ok, (time.perf_counter() - t0) / 5 * 1e3))
    finally:
        dist.destroy_process_group()


def test_rccl_all_gather_path_executes_world_1():
    """The production transport at last: `nccl` (= RCCL) process group of ONE rank, and the collector told not to
    short-circuit -- the 173 MB packed record buffer of a config-5 rank goes through all_gather_into_tensor on device
    memory (blocking, async_op + wait, collect_async) and the statistics through RCCL all-reduces."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_world1_worker, args=(30800 + (os.getpid() % 1000), q))
    p.start()
    ok, ms = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert all(ok), ok
    print('RCCL world-1 self-gather of 173 MB: %.3f ms' % ms)


def test_calls_leave_the_current_device_alone():
    """no entry point may change the calling thread's current HIP device."""
    if torch.cuda.device_count() < 2:
        env = _env('circle', 8)
        env.step(torch.zeros((8, 1), device=DEV))
        assert torch.cuda.current_device() == 0
        return
    from rl_on_manifold_amd import BatchedAtacomEnv
    torch.cuda.set_device(0)
    env = BatchedAtacomEnv('circle', 8, device='cuda:1')
    assert torch.cuda.current_device() == 0
    env.step(torch.zeros((8, 1), device='cuda:1'))
    env.get_constraints_logs()
    env.get_state()
    env.close()
    assert torch.cuda.current_device() == 0


def _run_bench(extra, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '20', '--warmup', '5',
                        '--min-time', '0.1', '--no-cpu-baseline', '--no-secondary'] + extra,
                       capture_output=True, text=True, timeout=600, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    # the contract: stdout is ONE line, the JSON (RCCL's version banner, printed through C stdio at process exit, used to
    # land behind it)
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_single_gpu():
    res = _run_bench(['--gpus', '1'])
    assert res['n_gpus'] == 1 and res['steps'] == 20 and res['timing']['blocks'] >= 3
    assert res['config']['batch_per_gpu'] == 8192 and 'IiwaAirHockey' in res['config']['workload']
    assert 0 < res['max_abs_c'] < 0.05                       # feasible initial states: the engine's residual, not the init's
    assert res['collection']['records'] == [1, 120, 8192, 44] and res['collection']['allgather_ms'] is None
    assert res['roofline']['bound'] == 'valu_f32' and abs(res['roofline']['frac'] - res['roofline']['achieved'] / 157.3) < 1e-12
    assert abs(res['roofline_hbm']['frac'] - res['roofline_hbm']['achieved'] / 8000.0) < 1e-12
    assert res['collection']['rccl_world1_selfgather']['identical'] is True


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` launches two ranks itself (gloo here: two ranks share the one GPU of this box)."""
    res = _run_bench(['--gpus', '2', '--batch', '2048'], env={'BENCH_DIST_BACKEND': 'gloo'})
    assert res['n_gpus'] == 2 and res['config']['global_batch'] == 4096
    col = res['collection']
    assert col['records'] == [2, 120, 2048, 44] and col['allgather_ms'] > 0
    # what an N > 1 line must carry for the driver's scaling run: the measured all-gather, its rate per
    # rank and the two-collection pipeline with the first gather left in flight -- none of them null
    assert col['allgather_GBps_per_rank'] > 0 and col['two_collections_overlapped_ms'] > 0
    assert col['bytes_received_per_rank'] == 2 * col['bytes_sent_per_rank'] == 2 * 120 * 2048 * 44 * 4
    assert col['backend'] == 'gloo' and res['scaling'] == 'weak'
    assert col['two_collections_overlapped_ms'] <= 2.2 * (col['rollout_ms'] + col['allgather_ms'])


def test_bench_eight_ranks_control_flow():
    """the 8-rank line the driver asks for at round end (`bench.py --gpus 8`), executed once before a node
    appears -- eight ranks over gloo sharing this box's one GPU: the port, the ranks, the barrier-bracketed blocks, the
    [8, 120, B, 44] layout of config 5's collection and the non-null fields of the all-gather leg."""
    res = _run_bench(['--gpus', '8', '--batch', '512', '--steps', '10', '--warmup', '2', '--min-time', '0.2'],
                     env={'BENCH_DIST_BACKEND': 'gloo'})
    assert res['n_gpus'] == 8 and res['config']['global_batch'] == 8 * 512 and res['scaling'] == 'weak'
    assert res['value'] > 0 and res['steps'] == 10 and res['warmup'] == 2
    col = res['collection']
    assert col['records'] == [8, 120, 512, 44] and col['backend'] == 'gloo'
    assert col['allgather_ms'] > 0 and col['allgather_GBps_per_rank'] > 0 and col['two_collections_overlapped_ms'] > 0
    assert col['bytes_received_per_rank'] == 8 * col['bytes_sent_per_rank'] == 8 * 120 * 512 * 44 * 4
    assert res['roofline']['frac'] > 0 and res['block_ms_p99'] >= res['timing']['block_ms_median']
    assert 'secondary' not in res and 'cpu_baseline' not in res          # N = 1 extras


def test_vectorized_env_core_shaped_loop():
    """The loop a vectorised Core runs (examples/circle_exp.py:26,42,72-73 in batch form): reset_all, step_all until every
    environment delivered an episode (finished ones masked out and frozen), then get_constraints_logs."""
    from rl_on_manifold_amd import VectorizedAtacomEnv
    n = 300
    env = VectorizedAtacomEnv('iiwa', n, horizon=9)
    assert env.number == n and env.info.horizon == 9
    lo, hi = env.info.observation_space.low, env.info.observation_space.high
    assert np.isfinite(hi[6:]).all() and np.isinf(hi[3:6]).all()
    active = torch.ones(n, dtype=torch.bool, device=DEV)
    obs, _ = env.reset_all(active)
    assert obs.shape == (n, 18)
    g = torch.Generator(device=DEV).manual_seed(5)
    steps = torch.zeros(n, dtype=torch.long, device=DEV)
    frozen = None
    for t in range(12):
        a = torch.rand((n, 5), device=DEV, generator=g) * 2 - 1
        prev = obs.clone()
        obs, r, ab, info = env.step_all(active.clone(), a)
        steps += active
        assert torch.equal(obs[~active], prev[~active])            # masked-out environments do not move
        assert (r[~active] == 0).all() and not ab[~active].any()
        active = active & ~info['last']
        if not bool(active.any()):
            break
    assert (steps <= 9).all() and int((steps == 9).sum()) > 0 and not bool(active.any())
    q = obs[:, 6:12].cpu().numpy()
    assert (np.abs(q) <= hi[6:12] + 1e-6).all()                     # ATACOM keeps the joints inside the bounds it advertises
    c_avg, c_max, c_dq = env.get_constraints_logs()
    assert np.isfinite([c_avg, c_max, c_dq]).all() and c_max < 0.05


@pytest.mark.parametrize('name', ['circle', 'iiwa'])
def test_graphed_rollout_equals_the_rollout_kernel(name):
    """GraphedRollout: observe -> torch policy -> atacom_step, T times, captured in one HIP graph.  With a deterministic
    policy that depends on the observation it must reproduce `rollout()` fed with the actions it chose (auto-reset
    included), replay after replay."""
    from rl_on_manifold_amd import GraphedRollout
    B, T = 300, 14
    env = _env(name, B, horizon=6)
    k, D = env.dims['null'], env.obs_dim
    g = torch.Generator(device=DEV).manual_seed(7)
    W = torch.randn((D, k), device=DEV, generator=g) * 0.5

    def policy(obs):                       # any capturable torch code
        return torch.tanh(obs @ W) * 1.2

    st = env.get_state().clone()
    loop = GraphedRollout(env, policy, T)
    assert torch.equal(env.get_state(), st)                        # building the graph leaves the engine where it was
    for rep in range(2):
        env.set_state(st)
        data = loop.replay()
        torch.cuda.synchronize()
        acts = data['action'].clone()
        env.set_state(st)
        ref = env.rollout(acts)
        for key in ('obs', 'next_obs', 'reward'):
            assert torch.equal(data[key], ref[key]), (rep, key)
        assert torch.equal(data['last'], ref['last']) and torch.equal(data['absorbing'], ref['absorbing'])
        assert torch.allclose(acts, torch.tanh(data['obs'] @ W) * 1.2, atol=1e-6)
        assert data['last'][5].all()                               # horizon 6, auto-reset inside the graph


@pytest.mark.parametrize('lanes', [1, 4, 8])
@pytest.mark.parametrize('kw', [{}, {'chart_mode': 'canonical'}, {'dynamics_mode': 'rigid_body'}])
def test_masked_step_on_the_device(kw, lanes):
    """atacom_step_masked (SURVEY 8b "no hidden synchronisation"; VERDICT r2 missing 4): the masked-out environments
    neither advance nor log -- a partial-mask step equals stepping a compacted batch of the active environments, state,
    servo joints and statistics included -- and the call is one capturable kernel launch."""
    from rl_on_manifold_amd import BatchedAtacomEnv
    n = 333
    g = torch.Generator(device=DEV).manual_seed(11)
    mask = torch.rand((n,), device=DEV, generator=g) < 0.6
    idx = torch.nonzero(mask)[:, 0]
    env = BatchedAtacomEnv('iiwa', n, device=DEV, lanes_per_env=lanes, horizon=9, auto_reset=True, **kw)
    init = torch.zeros((n, env.init_state_dim), device=DEV)
    init[:, :6] = torch.tensor([0.0, 0.7135214629060707, 0.0, -0.5024756033561426, 0.0, 1.9256631778550268], device=DEV) \
        + 0.03 * torch.randn((n, 6), device=DEV, generator=g)
    init[:, 12] = -0.5
    env.reset(state=init)
    small = BatchedAtacomEnv('iiwa', int(mask.sum()), device=DEV, lanes_per_env=lanes, horizon=9, auto_reset=True, **kw)
    small.reset(state=init[idx])
    env.get_constraints_logs(); small.get_constraints_logs()
    for t in range(12):
        a = torch.rand((n, 5), device=DEV, generator=g) * 2 - 1
        st0, aux0, obs0 = env.get_state().clone(), env.get_aux_state().clone(), env.reset(mask=torch.zeros(n, dtype=torch.uint8, device=DEV))
        obs, r, ab, info = env.step(a, mask=mask)
        so, sr, sab, sinfo = small.step(a[idx])
        assert torch.equal(obs[idx], so) and torch.equal(r[idx], sr) and torch.equal(ab[idx], sab)
        assert torch.equal(info['last'][idx], sinfo['last'])
        assert torch.equal(env.get_state()[~mask], st0[~mask]) and torch.equal(env.get_aux_state()[~mask], aux0[~mask])
        assert torch.equal(obs[~mask], obs0[~mask]) and (r[~mask] == 0).all() and not ab[~mask].any() and not info['last'][~mask].any()
    assert torch.equal(env.get_state()[idx], small.get_state())
    a_, b_ = env.get_constraints_logs(), small.get_constraints_logs()
    assert np.allclose(a_, b_, rtol=1e-6, atol=1e-7), (a_, b_)                  # the log counts the active steps only
    # capturable: no host synchronisation anywhere in a masked step (the vectorised surface included)
    from rl_on_manifold_amd import VectorizedAtacomEnv
    venv = VectorizedAtacomEnv('planar', 64, horizon=50, **({} if 'dynamics_mode' in kw else kw))
    m = torch.ones(64, dtype=torch.bool, device=DEV); m[::3] = False
    act = torch.zeros((64, 3), device=DEV)
    venv.step_all(m, act)                                                       # warm-up outside the capture
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        o2, r2, ab2, i2 = venv.step_all(m, act)
    before = venv.engine.get_state().clone()
    gr.replay()
    torch.cuda.synchronize()
    after = venv.engine.get_state()
    assert torch.equal(after[::3], before[::3]) and not torch.equal(after[1::3], before[1::3])


def test_step_into_rejects_what_the_raw_pointers_would_misread():
    """float64 / strided / host / mis-shaped tensors raise instead of being read as garbage."""
    env = _env('planar', 16)
    B, k, D = 16, 3, 12
    good = dict(actions=torch.zeros((B, k), device=DEV), obs=torch.empty((B, D), device=DEV),
                reward=torch.empty((B,), device=DEV), absorbing=torch.empty((B,), device=DEV, dtype=torch.uint8))
    env.step_into(**good)
    env.step_into(**good)                                                       # second call: the cached fast path
    for key, bad in (('actions', torch.zeros((B, k), device=DEV, dtype=torch.float64)),
                     ('actions', torch.zeros((B, 2 * k), device=DEV)[:, ::2]),
                     ('actions', torch.zeros((B, k))),
                     ('obs', torch.empty((B, D + 1), device=DEV)),
                     ('absorbing', torch.empty((B,), device=DEV, dtype=torch.bool)),
                     ('reward', torch.empty((B, 1), device=DEV))):
        with pytest.raises(ValueError):
            env.step_into(**dict(good, **{key: bad}))


def test_bound_step_is_step_into_with_the_checks_done_once():
    """BatchedAtacomEnv.bind_step: same launches as step_into (plain and masked), the same refusals -- at bind time --,
    the tensors kept alive by the callable, an error after close()."""
    B, k, D = 64, 3, 12
    e1, e2 = _env('planar', B), _env('planar', B)
    gen = torch.Generator(device=DEV); gen.manual_seed(3)
    acts = torch.rand((6, B, k), device=DEV, generator=gen) * 2 - 1
    mask = (torch.arange(B, device=DEV) % 3 != 0).to(torch.uint8)

    def bufs():
        return (torch.empty((B, D), device=DEV), torch.empty((B,), device=DEV),
                torch.empty((B,), device=DEV, dtype=torch.uint8), torch.empty((B,), device=DEV, dtype=torch.uint8))
    o1, r1, a1, l1 = bufs()
    o2, r2, a2, l2 = bufs()
    plain = [e2.bind_step(acts[i], o2, r2, a2, l2) for i in range(6)]
    masked = [e2.bind_step(acts[i], o2, r2, a2, l2, mask=mask) for i in range(6)]
    for i in range(6):
        m = mask if i % 2 else None
        e1.step_into(acts[i], o1, r1, a1, l1, mask=m)
        (masked if i % 2 else plain)[i]()
        for x, y in ((o1, o2), (r1, r2), (a1, a2), (l1, l2)):
            assert torch.equal(x, y)
    assert torch.equal(e1.get_state(), e2.get_state())
    with pytest.raises(ValueError):
        e2.bind_step(acts[0].double(), o2, r2, a2, l2)
    with pytest.raises(ValueError):
        e2.bind_step(acts[0], o2, r2, a2.bool(), l2)
    call = e2.bind_step(acts[0].clone(), *bufs())               # temporaries: the callable holds them
    call()
    torch.cuda.synchronize()
    e2.close()
    with pytest.raises(RuntimeError):
        call()


@pytest.mark.parametrize('name,kw', [('circle', {}), ('planar', {'task': 'D'}), ('iiwa', {'dynamics_mode': 'rigid_body'})])
def test_snapshot_restore_reproduces_the_run_bit_for_bit(name, kw):
    """atacom_snapshot_save / _restore: the whole persistent state -- what get_state() returns AND the stored initial states,
    the statistics accumulators, the episode counters of the device-side random reset, the servo joints.  Restore, repeat
    the same calls: identical outputs, identical statistics (the reference's counterpart would be pickling the env)."""
    from rl_on_manifold_amd import BatchedAtacomEnv
    B, T = 700, 40
    env = BatchedAtacomEnv(name, B, device=DEV, random_init=True, seed=4, auto_reset=True, horizon=15, **kw)
    env.reset()
    g = torch.Generator(device='cpu').manual_seed(0)
    acts = (torch.rand(3, T, B, env.dims['null'], generator=g) * 2 - 1).to(DEV)
    env.rollout(acts[0])                                     # some history: statistics, several auto-resets
    image = env.snapshot()
    assert image.dtype == torch.uint8 and image.numel() == env._lib.atacom_snapshot_bytes(env._h)
    first = env.rollout(acts[1])
    logs_first = env.get_constraints_logs()
    env.rollout(acts[2])                                     # wander off ...
    env.restore(image)                                       # ... and come back
    again = env.rollout(acts[1])
    logs_again = env.get_constraints_logs()
    for k in first:
        assert torch.equal(first[k], again[k]), k
    assert logs_first == logs_again
    # an image of another configuration is refused, not silently mis-read: by size ...
    other = BatchedAtacomEnv(name, B + 64, device=DEV, **kw)
    with pytest.raises(ValueError):
        other.restore(image)
    # ... and, where the byte size would fit, by the header every image starts with: a smaller batch, another
    # task / environment of the same footprint, another dtype
    smaller = BatchedAtacomEnv(name, B - 60, device=DEV, **kw)
    with pytest.raises(ValueError, match='another handle shape'):
        smaller.restore(image)
    twin = {'circle': ('circle_ec', {}), 'planar': ('planar', {'task': 'H'}), 'iiwa': ('iiwa', {})}[name]
    shape_twin = BatchedAtacomEnv(twin[0], B, device=DEV, **twin[1])
    if name != 'iiwa':                                       # (iiwa kinematic / rigid-body handles share one state layout)
        with pytest.raises(ValueError, match='another handle shape'):
            shape_twin.restore(image)
    else:
        shape_twin.restore(image)
    junk = torch.zeros_like(image)
    with pytest.raises(ValueError, match='bad magic'):
        env.restore(junk)
    env.restore(image)                                       # and the good image still restores
    assert torch.equal(env.snapshot(), image)


_DETERMINISM_CHILD = r"""
import hashlib, sys, torch
sys.path.insert(0, %(root)r)
from rl_on_manifold_amd import BatchedAtacomEnv
env = BatchedAtacomEnv('iiwa', 8192, device='cuda:0', auto_reset=True)          # lanes_per_env = 0: the library's choice
image = torch.load(%(image)r).to('cuda:0')
acts = torch.load(%(acts)r).to('cuda:0')
env.restore(image)
h = hashlib.sha256()
for t in range(acts.shape[0]):
    o, r, ab, info = env.step(acts[t])
    for x in (o, r, ab, info['last']):
        h.update(x.cpu().numpy().tobytes())
roll = env.rollout(acts[:8])
for k in sorted(roll):
    h.update(roll[k].cpu().numpy().tobytes())
h.update(env.snapshot().cpu().numpy().tobytes())
print('RESULT', env.lanes_per_env, env.rollout_lanes_per_env, h.hexdigest())
"""


def test_default_mapping_is_deterministic_across_processes(tmp_path):
    """VERDICT r4 weak 2 / ADVICE r4: two FRESH processes create the default handle (lanes_per_env = 0) at the headline
    batch, must report the same kernel mappings and -- restored from one snapshot -- produce 50 steps, a T-step rollout and
    a final state that are bit-identical.  (Round 4 chose 8 lanes against 4 by timing both at create: a wall-clock race per
    process.)  The same snapshot restored into a handle created with ANOTHER named mapping still replays the writer's bits
    if that handle left the choice to the library, because the image carries the writer's mappings."""
    from rl_on_manifold_amd import BatchedAtacomEnv
    B, T = 8192, 50
    env = BatchedAtacomEnv('iiwa', B, device=DEV, auto_reset=True, random_init=True, seed=11)
    env.reset()
    g = torch.Generator(device='cpu').manual_seed(5)
    acts = torch.rand(T, B, env.dims['null'], generator=g) * 2.4 - 1.2
    env.rollout(acts[:30].to(DEV))                                  # off the reset pose, some auto-resets
    image = env.snapshot()
    torch.save(image.cpu(), str(tmp_path / 'image.pt'))
    torch.save(acts, str(tmp_path / 'acts.pt'))
    code = _DETERMINISM_CHILD % {'root': ROOT, 'image': str(tmp_path / 'image.pt'), 'acts': str(tmp_path / 'acts.pt')}
    envv = {k: v for k, v in os.environ.items() if k != 'ATACOM_CALIBRATE'}
    outs = []
    for _ in range(2):
        r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600, env=envv)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([l for l in r.stdout.splitlines() if l.startswith('RESULT')][-1].split())
    assert outs[0] == outs[1], outs
    assert outs[0][1:3] == ['8', '8']                                # the static policy at this batch
    # a quad-mapped writer: an auto handle adopts the image's mappings and replays its bits; a named handle keeps its own
    quad = BatchedAtacomEnv('iiwa', B, device=DEV, auto_reset=True, lanes_per_env=4)
    quad.restore(image)
    assert quad.lanes_per_env == 4                                  # named: kept
    a0 = acts[0].to(DEV)
    want = quad.step(a0)
    qimg_env = BatchedAtacomEnv('iiwa', B, device=DEV, auto_reset=True, lanes_per_env=4)
    qimg_env.restore(image)
    qimg = qimg_env.snapshot()                                      # an image written by a quad handle
    auto = BatchedAtacomEnv('iiwa', B, device=DEV, auto_reset=True)
    assert auto.lanes_per_env == 8
    auto.restore(qimg)
    assert (auto.lanes_per_env, auto.rollout_lanes_per_env) == (4, 4)          # adopted
    got = auto.step(a0)
    for x, y in zip(want[:3], got[:3]):
        assert torch.equal(x, y)
    auto.restore(image)                                             # and back to the 8-lane writer's mappings
    assert (auto.lanes_per_env, auto.rollout_lanes_per_env) == (8, 8)


@pytest.mark.parametrize('name,B,T', [('circle', 1 << 20, 12), ('planar', 1 << 18, 6), ('iiwa', 1 << 18, 4)])
def test_saturation_batches_equal_small_batches(name, B, T):
    """The largest batches the bench line runs (its `saturation` records: a million circle environments, 262144 planar; and
    262144 iiwa): everything finite, and -- environments being independent -- the first and the last 192 environments of the big
    launch are bit for bit what a 192-environment engine on the same mapping computes from the same states and actions
    (index arithmetic at the far end of the buffers)."""
    from rl_on_manifold_amd import BatchedAtacomEnv
    g = torch.Generator(device=DEV).manual_seed(4)
    big = BatchedAtacomEnv(name, B, device=DEV, auto_reset=True, random_init=True, seed=5, horizon=3)
    assert big.rollout_lanes_per_env == 1
    big.reset()
    k = big.dims['null']
    acts = torch.rand((T, B, k), device=DEV, generator=g) * 2 - 1
    st0 = big.get_state().clone()
    out = big.rollout(acts)
    for key in ('obs', 'next_obs', 'reward'):
        assert bool(torch.isfinite(out[key]).all()), key
    assert bool(torch.isfinite(big.get_state()).all())
    c_avg, c_max, c_dq = big.get_constraints_logs()
    assert np.isfinite([c_avg, c_max, c_dq]).all() and c_max < 0.5
    n = 192
    for sl in (slice(0, n), slice(B - n, B)):
        small = BatchedAtacomEnv(name, n, device=DEV, auto_reset=False, lanes_per_env=1, horizon=10 ** 6)
        small.set_state(st0[sl])
        ref = small.rollout(acts[:2, sl].contiguous())           # two steps: before the first auto-reset of the big run
        for key in ('obs', 'next_obs', 'reward', 'absorbing'):
            assert torch.equal(ref[key], out[key][:2, sl]), (key, sl)


def test_mapping_census_matches_the_library():
    """Round 6: which lane mappings are instantiated is ONE rule (csrc/atacom_ops_impl.h: has_mapping); a request for a
    mapping that does not exist runs the widest narrower one and the handle SAYS so (float64 8-lane policy
    rollouts silently ran the quad kernel and reported 8).  tests/conftest.py: mapping_exists mirrors the rule for test
    selection -- held against the library here, for every (environment, dtype, request), step / T-step / policy kernels,
    kinematic and rigid-body handles."""
    from conftest import mapping_exists
    from rl_on_manifold_amd import BatchedAtacomEnv

    def runs(name, dt, lanes, kind, dyn):
        return max(l for l in (1, 2, 4, 8) if l <= lanes and mapping_exists(name, dt, l, kind, dyn))

    for name in ('circle', 'circle_ec', 'circle_t', 'planar', 'iiwa'):
        for dt in ('f32', 'f64'):
            for dyn in ((False, True) if name == 'iiwa' else (False,)):
                for lanes in (1, 2, 4, 8):
                    kw = {'dynamics_mode': 'rigid_body_ff'} if dyn else {}
                    env = BatchedAtacomEnv(name, 64, device=DEV, dtype={'f32': torch.float32, 'f64': torch.float64}[dt],
                                           lanes_per_env=lanes, **kw)
                    got = (env.lanes_per_env, env.rollout_lanes_per_env, env.policy_lanes_per_env)
                    want = (runs(name, dt, lanes, 'step', dyn), runs(name, dt, lanes, 'step', dyn), runs(name, dt, lanes, 'mlp', dyn))
                    assert got == want, (name, dt, dyn, lanes, got, want)
                    env.close()
    # the static policy (lanes_per_env = 0) only ever names instantiated mappings
    for name, dt, B, want in (('iiwa', 'f64', 8192, (8, 8, 4)), ('iiwa', 'f64', 16384, (4, 4, 4)), ('iiwa', 'f64', 20000, (4, 4, 4)),
                              ('planar', 'f64', 8192, (4, 4, 4)), ('planar', 'f32', 8192, (4, 8, 8)), ('iiwa', 'f32', 8192, (8, 8, 8)),
                              ('circle', 'f32', 4096, (1, 1, 1)), ('circle', 'f64', 4096, (1, 1, 1))):
        env = BatchedAtacomEnv(name, B, device=DEV, dtype={'f32': torch.float32, 'f64': torch.float64}[dt])
        assert (env.lanes_per_env, env.rollout_lanes_per_env, env.policy_lanes_per_env) == want, (name, dt, B)
        env.close()


def test_snapshot_carries_the_seed_and_a_format_number():
    """(1) an image of another format is refused as such; (2) the generator key travels with the image -- a handle
    re-keyed by seed() writes images that replay bit for bit in a handle created with ANOTHER seed; (3) a RolloutCollector
    built before a restore that changed the mappings refuses to collect."""
    from rl_on_manifold_amd import BatchedAtacomEnv, AtacomError
    from rl_on_manifold_amd.rollout import RolloutCollector
    B, T = 256, 12
    kw = dict(device=DEV, auto_reset=True, random_init=True, horizon=5, obs_noise=True, env_noise=True)
    a = BatchedAtacomEnv('planar', B, seed=3, **kw)
    a.reset()
    a.seed(12345)
    g = torch.Generator(device=DEV).manual_seed(0)
    acts = torch.rand((T, B, 3), device=DEV, generator=g) * 2 - 1
    a.rollout(acts[:4])
    img = a.snapshot()
    want = a.rollout(acts)
    b = BatchedAtacomEnv('planar', B, seed=99, **kw)               # another key: adopts the image's
    b.restore(img)
    got = b.rollout(acts)
    for k in ('obs', 'next_obs', 'reward', 'last'):
        assert torch.equal(want[k], got[k]), k
    # (1) another format number
    bad = img.clone()
    bad[48:52] = torch.tensor([0, 0, 0, 0], dtype=torch.uint8, device=DEV)      # SnapHeader.version (csrc/atacom_capi.cpp)
    with pytest.raises((AtacomError, ValueError), match='image format 0'):
        b.restore(bad)
    # (3) the collector's agreement is on the mappings it saw
    q = BatchedAtacomEnv('iiwa', 512, device=DEV, auto_reset=True, lanes_per_env=4)
    qimg = q.snapshot()
    auto = BatchedAtacomEnv('iiwa', 512, device=DEV, auto_reset=True)
    col = RolloutCollector(auto)
    col.collect_local(2, actions=torch.zeros((2, 512, 5), device=DEV))
    auto.restore(qimg)
    assert auto.lanes_per_env == 4
    with pytest.raises(ValueError, match='mappings changed'):
        col.collect_local(2, actions=torch.zeros((2, 512, 5), device=DEV))
    RolloutCollector(auto).collect_local(2, actions=torch.zeros((2, 512, 5), device=DEV))


@pytest.mark.parametrize('mode,lanes', [('kinematic', 1), ('kinematic', 2), ('kinematic', 4), ('kinematic', 8),
                                        ('rigid_body_ff', 1), ('rigid_body_ff', 4)])
def test_policy_kernel_network_sees_the_observation_it_writes(mode, lanes):
    """A network that COPIES five of its inputs to its outputs (ReLU kept linear by a hidden bias of +10, output bias -10),
    run for every input window: the action of step t must be the observation the kernel itself wrote for step t, in every
    environment of the batch.  Guards the staging of the in-kernel network (observation rows, biases, weights in LDS and
    registers) against lane- or step-dependent corruption -- round 5 met a build that passed step 0 and was off by exactly one
    bias for three of every four environments afterwards (profiles/r05_dyn_mlp_park.md)."""
    from rl_on_manifold_amd import BatchedAtacomEnv, MlpPolicy
    # (the rigid-body kernels -- where the defect of profiles/r06_exec_mask_copies.md was met -- at the headline batch)
    B, T = (1000, 6) if mode == 'kinematic' else (8192, 12)
    for j0 in (0, 5, 10, 13):
        W1 = torch.zeros(64, 18); W1[:18, :18] = torch.eye(18)
        W3 = torch.zeros(5, 64)
        for k in range(5):
            W3[k, j0 + k] = 1.0
        pol = MlpPolicy(W1, torch.full((64,), 10.0), torch.eye(64), torch.zeros(64), W3, torch.full((5,), -10.0), std=torch.zeros(5))
        env = BatchedAtacomEnv('iiwa', B, device=DEV, dynamics_mode=mode, lanes_per_env=lanes, random_init=True, seed=3)
        env.reset()
        out = env.rollout_policy(pol, T)
        d = (out['action'] - out['obs'][:, :, j0:j0 + 5]).abs()
        assert float(d.max()) < 1e-4, (mode, lanes, j0, torch.nonzero(d.amax(2) > 1e-4)[:8].tolist())


def test_graphed_rollout_leaves_no_warmup_residue():
    """GraphedRollout warms up with real steps before the capture: they must not stay in the constraint
    statistics nor shift the device-side random resets -- the first replay equals the rollout kernel of a twin engine that
    never saw a warm-up, fed with the actions the graph's policy chose, and so do the constraint logs."""
    from rl_on_manifold_amd import BatchedAtacomEnv, GraphedRollout
    B, T = 512, 12
    mk = lambda: BatchedAtacomEnv('planar', B, device=DEV, random_init=True, seed=9, auto_reset=True, horizon=5)
    a, b = mk(), mk()
    a.reset(); b.reset()
    a.get_constraints_logs(); b.get_constraints_logs()
    g = torch.Generator(device=DEV).manual_seed(3)
    W = torch.randn((a.obs_dim, 3), device=DEV, generator=g) * 0.5
    policy = lambda obs: torch.tanh(obs @ W) * 1.2
    gr = GraphedRollout(a, policy, T)
    out = gr.replay()
    torch.cuda.synchronize()
    nobody = torch.zeros((B,), device=DEV, dtype=torch.uint8)
    for t in range(T):                                        # the same kernels, launched eagerly on the twin
        obs = b.reset(mask=nobody)
        assert torch.equal(out['obs'][t], obs), t             # incl. the re-drawn puck positions after the auto-resets
        nobs, r, ab, info = b.step(policy(obs))
        assert torch.equal(out['next_obs'][t], nobs) and torch.equal(out['reward'][t], r), t
        assert torch.equal(out['absorbing'][t].bool(), ab.bool()) and torch.equal(out['last'][t].bool(), info['last'].bool()), t
    assert out['last'][4].all() and not torch.equal(out['obs'][5][:, :2], out['obs'][0][:, :2])   # auto-reset, new draw
    assert a.get_constraints_logs() == b.get_constraints_logs()
