"""Host-side logic that needs neither a GPU nor the library: sharding arithmetic, the packed-record layout and its views,
bench.py's helpers.  Property-based where the property is the specification."""
import os
import sys

import numpy as np
import torch
from hypothesis import given, settings, strategies as st

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from rl_on_manifold_amd.rollout import RolloutCollector, shard_bounds   # noqa: E402


@settings(max_examples=200, deadline=None)
@given(st.integers(1, 200000), st.integers(1, 64))
def test_shard_bounds_partition_the_batch(gb, world):
    spans = [shard_bounds(gb, world, r) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == gb
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))             # contiguous, in rank order
    sizes = [hi - lo for lo, hi in spans]
    assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)   # remainder goes to the low ranks


class _Env:
    """The smallest engine surface the collector touches."""
    def __init__(self, B, D, k):
        self.batch, self.obs_dim, self.dims = B, D, {'null': k}
        self.device = torch.device('cpu')

    def rollout(self, actions):
        T, B, k = actions.shape
        g = torch.Generator().manual_seed(T * 1000 + B)
        return {'obs': torch.randn((T, B, self.obs_dim), generator=g, dtype=torch.float64),
                'next_obs': torch.randn((T, B, self.obs_dim), generator=g, dtype=torch.float64),
                'reward': torch.randn((T, B), generator=g, dtype=torch.float64),
                'absorbing': torch.rand((T, B), generator=g) < 0.1, 'last': torch.rand((T, B), generator=g) < 0.2,
                'action': torch.as_tensor(actions, dtype=torch.float64)}


@settings(max_examples=50, deadline=None)
@given(st.integers(1, 9), st.integers(1, 7), st.integers(1, 6), st.integers(1, 5))
def test_packed_records_round_trip(B, T, D, k):
    """collect_local packs (s, a, r, s', absorbing, last) into F = 2D + k + 3 floats; unpack / time_major give them back."""
    env = _Env(B, D, k)
    col = RolloutCollector(env)
    acts = torch.randn((T, B, k), dtype=torch.float64)
    ref = env.rollout(acts)
    buf = col.collect_local(T, actions=acts)
    assert buf.shape == (T, B, 2 * D + k + 3)
    data = col.time_major(col.unpack(col.gather(buf)))
    for key in ('obs', 'next_obs', 'reward', 'action'):
        assert torch.equal(data[key], ref[key]), key
    assert torch.equal(data['absorbing'], ref['absorbing']) and torch.equal(data['last'], ref['last'])
    flat = col.gather(buf).reshape(-1, col.F)                            # the flat sample set an on-policy fit consumes
    assert flat.shape[0] == T * B and flat.is_contiguous()


def test_collector_rejects_a_shard_of_the_wrong_size():
    import pytest
    with pytest.raises(ValueError):
        RolloutCollector(_Env(5, 4, 1), global_batch=7)                  # world 1: the single shard must hold all 7


def test_bench_helpers():
    import bench
    assert 1 <= bench.usable_cores() <= (os.cpu_count() or 1)
    for name, (M, N, K, nq, sub) in bench.SHAPES.items():
        assert N - M == K and bench.ALGO_BYTES[name] > 0
    rv, r = bench.roofline_objects('iiwa', 8192, 0.028)              # the binding roof (vector ALU) first, then the HBM view
    assert r['bound'] == 'hbm' and rv['bound'] == 'valu_f32'
    assert abs(r['achieved'] - 400 * 8192 / 0.028e-3 / 1e9) < 1e-6 and abs(r['frac'] - r['achieved'] / 8000.0) < 1e-12
    assert rv['unit'] == 'TFLOP/s' and 0.05 < rv['frac'] < 0.2
    rc, _ = bench.roofline_objects('iiwa', 8192, 0.028, chart='canonical')
    assert rc['algorithmic_flops_per_launch'] < rv['algorithmic_flops_per_launch'] / 3
    port = bench._free_port()
    assert 1024 < port < 65536
