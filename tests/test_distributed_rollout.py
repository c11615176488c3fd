"""N > 1 path on CPU: world_size-2 gloo processes shard the env batch, roll out independently and all-gather
the finished rollout; the assembled global dataset must equal a single-process run over the whole batch."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from rl_on_manifold_amd.rollout import RolloutCollector, shard_bounds, to_mushroom_dataset   # noqa: E402

GLOBAL_B, T, NAME = 10, 12, 'planar'        # 10 envs over 2 ranks; a ragged split is tested with 3 ranks' bounds


def _actions():
    rng = np.random.default_rng(42)
    return rng.uniform(-1.2, 1.2, (T, GLOBAL_B, 3))


def _init_q():
    from oracle import robots
    rng = np.random.default_rng(7)
    return robots.PLANAR_INIT_Q + rng.normal(0, 0.05, (GLOBAL_B, 3))


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle_engine import OracleEngine
        lo, hi = shard_bounds(GLOBAL_B, world, rank)
        eng = OracleEngine(NAME, hi - lo, init_q=_init_q()[lo:hi], horizon=5)
        col = RolloutCollector(eng, global_batch=GLOBAL_B)
        sm = col.collect(T, actions=_actions()[:, lo:hi])             # shard-major views [W, T, Bm, ...]
        assert sm['obs'].shape[:3] == (world, T, max(col.sizes))
        data = col.time_major(sm)
        # the same collection with the all-gather left in flight gives the same dataset
        eng2 = OracleEngine(NAME, hi - lo, init_q=_init_q()[lo:hi], horizon=5)
        pend = RolloutCollector(eng2, global_batch=GLOBAL_B).collect_async(T, actions=_actions()[:, lo:hi])
        again = pend.wait()
        assert all(torch.equal(again[k_], sm[k_]) for k_ in sm)
        stats = col.get_constraints_logs(n_logged=T * (hi - lo))
        q.put((rank, {k: v.numpy() for k, v in data.items()}, stats))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_shard_bounds_cover_the_batch():
    for gb, w in ((65536, 8), (10, 3), (7, 8), (8192, 1)):
        spans = [shard_bounds(gb, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == gb
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


@pytest.mark.parametrize('world', [2, 3, 8])       # 8: the shard count of BASELINE config 5 (ragged: 10 envs over 8 ranks)
def test_two_rank_gloo_rollout_equals_single_process(world):
    from oracle_engine import OracleEngine
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference over the whole batch
    eng = OracleEngine(NAME, GLOBAL_B, init_q=_init_q(), horizon=5)
    rc = RolloutCollector(eng)
    ref = rc.time_major(rc.collect(T, actions=_actions()))
    ref_stats = eng.get_constraints_logs()
    for rank, data, stats in results:
        for k in ('obs', 'action', 'reward', 'next_obs', 'absorbing', 'last'):
            assert data[k].shape == tuple(ref[k].shape), (k, data[k].shape)
            assert np.allclose(data[k].astype(np.float64), ref[k].numpy().astype(np.float64), atol=1e-12), (rank, k)
        assert np.allclose(stats, ref_stats, atol=1e-12)
    # every rank holds the same global dataset; episodes end at the horizon (last) and restart from reset
    d0 = results[0][1]
    assert d0['last'][4].all() and not d0['last'][3].any()
    assert np.allclose(d0['obs'][5], d0['obs'][0])          # auto-reset: step 5 starts from the initial state


def _solo_worker(port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=0, world_size=1)
    try:
        from oracle_engine import OracleEngine
        eng = OracleEngine(NAME, 4, init_q=_init_q()[:4], horizon=5)
        col = RolloutCollector(eng, force_collective=True)
        local = col.collect_local(T, actions=_actions()[:, :4])
        out = torch.full((1,) + tuple(local.shape), float('nan'), dtype=local.dtype)
        g = col.gather(local, out=out)                            # the collective runs although world == 1 ...
        ok = g.data_ptr() == out.data_ptr() and torch.equal(out[0], local)       # ... and fills the caller's buffer
        g2, work = col.gather(local, async_op=True)
        work.wait()
        ok = ok and torch.equal(g2[0], local) and g2.data_ptr() != local.data_ptr()
        stats = col.get_constraints_logs(n_logged=T * 4)          # all-reduces of one rank
        q.put((ok, stats))
    finally:
        dist.destroy_process_group()


def _mapping_worker(rank, world, port, q, disagree):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle_engine import OracleEngine
        eng = OracleEngine(NAME, 4, init_q=_init_q()[:4], horizon=5)
        eng.lanes_per_env, eng.rollout_lanes_per_env = (4 if (disagree and rank == 1) else 8), 8
        try:
            col = RolloutCollector(eng)
            q.put((rank, 'ok', col.mappings))
        except ValueError as e:
            q.put((rank, 'refused', str(e)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('disagree', [False, True])
def test_ranks_must_agree_on_the_kernel_mapping(disagree):
    """Equally sized shards that run different lane mappings sum in different orders: the collector gathers the mappings once
    at construction and refuses on EVERY rank (the default used to be decided by a wall-clock race per process)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000) + int(disagree)
    procs = [ctx.Process(target=_mapping_worker, args=(r, 2, port, q, disagree)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if disagree:
        assert all(r[1] == 'refused' and 'different kernel mappings' in r[2] for r in res)
    else:
        assert all(r[1] == 'ok' and r[2] == [(4, 8, 8), (4, 8, 8)] for r in res)


def test_forced_collective_in_a_world_of_one_rank():
    """force_collective=True sends a one-rank world through the real all-gather / all-reduce calls (the CPU twin of the
    RCCL world-1 test in test_gpu_rollout.py)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_solo_worker, args=(31500 + (os.getpid() % 2000), q))
    p.start()
    ok, stats = q.get(timeout=180)
    p.join(timeout=60)
    assert p.exitcode == 0 and ok and np.isfinite(stats).all()


def test_record_layout_ragged_padding():
    from rl_on_manifold_amd.rollout import RecordLayout
    lay = RecordLayout([3, 2, 2], obs_dim=4, n_null=1)
    assert lay.F == 12 and lay.Bm == 3
    g = torch.arange(3 * 5 * 3 * 12, dtype=torch.float32).reshape(3, 5, 3, 12)
    m = lay.valid_mask()
    assert m.tolist() == [[True, True, True], [True, True, False], [True, True, False]]
    tm = lay.time_major(lay.unpack(g))
    assert tm['obs'].shape == (5, 7, 4) and torch.equal(tm['reward'][:, 3], g[1, :, 0, 5])
    # a caller-supplied send buffer gets its padding rows zeroed by the fallback packing path
    from oracle_engine import OracleEngine
    eng = OracleEngine('circle', 2, horizon=6)
    col = RolloutCollector(eng)
    col.Bm = 3                                                   # as on the short rank of a ragged split
    out = torch.full((4, 3, col.F), 7.0, dtype=torch.float64)
    buf = col.collect_local(4, actions=np.zeros((4, 2, 1)), out=out)
    assert (buf[:, 2] == 0).all() and torch.isfinite(buf).all()


def test_policy_driven_collection_and_mushroom_dataset():
    from oracle_engine import OracleEngine
    eng = OracleEngine('circle', 4, horizon=6)
    col = RolloutCollector(eng)
    rng = np.random.default_rng(0)
    data = col.time_major(col.collect(9, policy=lambda obs: rng.uniform(-1, 1, (4, 1))))
    assert data['obs'].shape == (9, 4, 4) and data['action'].shape == (9, 4, 1)
    assert data['last'][5].all() and data['last'].sum() == 4
    ds = to_mushroom_dataset(data)
    assert len(ds) == 36 and len(ds[0]) == 6 and ds[5][5] is True and ds[8][5] is True
    # s' of step t is s of step t+1 inside an episode
    assert np.allclose(ds[0][3], ds[1][0])
