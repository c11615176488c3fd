#!/bin/bash
# Round 4, the coordinate slack of the canonical chart is the STIFFEST row (it was the first in row order): chart tests with
# their printed statistics, the float32 soaks on both seeds and both default mappings, the float64 soak, speed of the
# canonical kernels, then the whole suite
export TMPDIR=/tmp
O=gpurun_out/prof_r04j
rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_chart.py -m gpu -q -s 2>&1 | grep -v amdgpu.ids | grep -i "hand-made\|rollout:\|passed\|failed" | cut -c1-250 > $O/chart_tests.log
cat $O/chart_tests.log
for seed in 23 11; do for l in 8 4; do
  MB_SEED=$seed MB_CHART=canonical python profiles/tools/gpu_sens_probe.py $l 8192 40 2>&1 | grep -v amdgpu.ids > $O/sens_soak_canonical_l${l}_seed$seed.log
  grep -h "verdict\|UNEXPLAINED" $O/sens_soak_canonical_l${l}_seed$seed.log | cut -c1-300
done; done
MB_CHART=canonical MB_DTYPE=f64 python profiles/tools/gpu_sens_probe.py 8 8192 40 2>&1 | grep -v amdgpu.ids > $O/soak_canonical_f64_l8.log
grep -h "err:" $O/soak_canonical_f64_l8.log
for rep in 1 2; do python bench.py --chart-mode canonical --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('canonical bench  %.2f us  collection %.2f us/step' % (r['ms_per_step']*1e3, r['collection']['rollout_ms']/120*1e3))"; done | tee $O/bench_canonical.log
MB_WARM=60 MB_CHART=canonical MB_ROLLOUT=1 MB_LANES=8,4 MB_BATCHES=8192 python profiles/tools/gpu_microbench.py iiwa planar 2>&1 | grep -v amdgpu.ids | tee $O/microbench_canonical.log | cut -c1-110
