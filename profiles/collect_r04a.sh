#!/bin/bash
# Round 4, first GPU pass (run through gpurun from the repo root):
#   gpurun --timeout 2700 -- 'bash profiles/collect_r04a.sh'
# the new noise-option tests first, then the whole GPU suite, the bench lines, kernel stats of both charts and the
# noise-options A/B (the build without the options compiled in: build/ab/libatacom_nonoise.so).
export TMPDIR=/tmp
O=gpurun_out/r04a
rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_noise.py -x -q -s 2>&1 | grep -v amdgpu.ids | tail -60 > $O/noise_tests.log
tail -3 $O/noise_tests.log
python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_noise.py 2>&1 | grep -v amdgpu.ids | tail -40 > $O/gpu_suite.log
tail -3 $O/gpu_suite.log
python bench.py 2>/dev/null | tail -1 > $O/bench_default.json
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_cmd.json
python bench.py --chart-mode canonical --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_canonical.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- \
    python bench.py --steps 300 --warmup 30 --min-time 0.3 --no-cpu-baseline --no-secondary > $O/bench_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_canonical -o s -- \
    python bench.py --steps 300 --warmup 30 --min-time 0.3 --no-cpu-baseline --no-secondary --chart-mode canonical > $O/bench_canonical_under_rocprof.log 2>&1
# A/B: the domain-randomisation options compiled in (off) vs compiled out -- same box, interleaved, identical results expected
if [ -f build/ab/libatacom_nonoise.so ]; then
  for rep in 1 2 3; do
    for lib in build/ab/libatacom_nonoise.so rl_on_manifold_amd/libatacom_hip.so; do
      ATACOM_LIB=$PWD/$lib MB_WARM=60 MB_ROLLOUT=1 MB_LANES=4,8 MB_BATCHES=8192 python tests/gpu_microbench.py iiwa planar
      ATACOM_LIB=$PWD/$lib MB_WARM=60 MB_ROLLOUT=1 MB_CHART=canonical MB_LANES=8 MB_BATCHES=8192 python tests/gpu_microbench.py iiwa
    done
  done 2>&1 | grep -v amdgpu.ids > $O/ab_noise_options.log
fi
find $O -name '*kernel_trace.csv' -size +2M -delete; find $O -name '*.db' -delete; find $O -name '*agent_info*' -delete
du -sh $O
cut -c1-400 $O/bench_driver_cmd.json
