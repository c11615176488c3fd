#!/bin/bash
# Round 5, first GPU call: the suite on the build with the deterministic default mapping / snapshot lanes / seed setter, smoke,
# baseline bench lines of this round's box, float64 timing per mapping, planar 8 lanes against the quad.
export TMPDIR=/tmp
O=gpurun_out/prof_r05a
rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -15 > $O/gpu_suite.log
tail -3 $O/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -5 > $O/smoke.log
tail -1 $O/smoke.log
MB_DTYPE=f64 MB_WARM=30 MB_ROLLOUT=1 MB_LANES=1,2,4,8 MB_BATCHES=8192 python profiles/tools/gpu_microbench.py iiwa 2>&1 | grep -v amdgpu.ids > $O/f64_lanes.log
cat $O/f64_lanes.log
MB_WARM=30 MB_ROLLOUT=1 MB_LANES=4,8,4,8 MB_BATCHES=8192 python profiles/tools/gpu_microbench.py planar 2>&1 | grep -v amdgpu.ids > $O/planar_lanes.log
cat $O/planar_lanes.log
python bench.py 2>$O/bench_default.err | tail -1 > $O/bench_default.json
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_cmd.json
cut -c1-300 $O/bench_driver_cmd.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/prof_r05a/bench_default.json'))
print('headline us/step', d['ms_per_step']*1e3, 'kernel us', d['roofline']['kernel_ms']*1e3, 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'], d['roofline_hbm']['traffic_over_algorithmic'])
for r in d.get('secondary', []):
    print(r['workload'], '|', round(r['ms_per_step']*1e3, 2), 'us', r.get('lanes_per_env'), (r.get('roofline') or {}).get('frac'))
PY
