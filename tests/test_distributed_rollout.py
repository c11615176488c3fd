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
