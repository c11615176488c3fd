#!/bin/bash
# Round 4, final state: suite, smoke, bench lines (default and the driver's command), kernel statistics of the headline run
export TMPDIR=/tmp
O=gpurun_out/prof_r04i
rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -5 > $O/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -5 > $O/smoke.log
tail -2 $O/gpu_suite.log; tail -1 $O/smoke.log
python bench.py 2>/dev/null | tail -1 > $O/bench_default.json
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_cmd.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- \
    python bench.py --steps 300 --warmup 30 --min-time 0.3 --no-cpu-baseline --no-secondary > $O/bench_under_rocprof.log 2>&1
python profiles/tools/trace_percentiles.py $O/stats "k_step<float, atacom::Iiwa, 8" > $O/launch_percentiles.log
cat $O/launch_percentiles.log
find $O -name '*kernel_trace.csv' -size +2M -delete; find $O -name '*.db' -delete; find $O -name '*agent_info*' -delete
cut -c1-200 $O/bench_driver_cmd.json
