#!/bin/bash
# Round 4, the timed choice of the single-step mapping at create (calibrate_step_lanes): what it measures on this box next to
# the sustained figures of both mappings (bench workload, three interleaved runs each), then the suite, smoke and the bench
# lines of the build that ships.
export TMPDIR=/tmp
O=gpurun_out/prof_r04h
rm -rf $O; mkdir -p $O
for rep in 1 2 3; do
  ATACOM_CALIBRATE=verbose python bench.py --no-cpu-baseline --no-secondary 2> $O/err.tmp | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('auto    lanes %d  %.2f us  kernel %.2f us' % (r['config']['lanes_per_env'], r['ms_per_step']*1e3, r['roofline']['kernel_ms']*1e3))"
  grep "^\[atacom\]" $O/err.tmp
  for l in 8 4; do python bench.py --lanes $l --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('lanes=$l         %.2f us  kernel %.2f us' % (r['ms_per_step']*1e3, r['roofline']['kernel_ms']*1e3))"; done
done > $O/calibration_vs_sustained.log 2>&1
rm -f $O/err.tmp
cat $O/calibration_vs_sustained.log
python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -5 > $O/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -5 > $O/smoke.log
tail -2 $O/gpu_suite.log; tail -1 $O/smoke.log
python bench.py 2>/dev/null | tail -1 > $O/bench_default.json
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_cmd.json
cut -c1-200 $O/bench_driver_cmd.json
