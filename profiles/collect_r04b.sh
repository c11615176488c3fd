#!/bin/bash
# Round 4, second GPU pass: the whole GPU suite on the build with the noise options as separate kernel instantiations, the
# A/B against the build without them (identical kernels expected), kernel stats + launch-duration percentiles of both charts.
export TMPDIR=/tmp
O=gpurun_out/r04b
rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -q -x -s 2>&1 | grep -v amdgpu.ids | grep -v "^$" | tail -150 > $O/gpu_suite.log
tail -3 $O/gpu_suite.log
if [ -f build/ab/libatacom_nonoise.so ]; then
  for rep in 1 2 3; do
    for lib in build/ab/libatacom_nonoise.so rl_on_manifold_amd/libatacom_hip.so; do
      ATACOM_LIB=$PWD/$lib MB_WARM=60 MB_ROLLOUT=1 MB_LANES=4,8 MB_BATCHES=8192 python tests/gpu_microbench.py iiwa planar
      ATACOM_LIB=$PWD/$lib MB_WARM=60 MB_ROLLOUT=1 MB_CHART=canonical MB_LANES=8 MB_BATCHES=8192 python tests/gpu_microbench.py iiwa
    done
  done 2>&1 | grep -v amdgpu.ids > $O/ab_noise_kernels.log
fi
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- \
    python bench.py --steps 300 --warmup 30 --min-time 0.3 --no-cpu-baseline --no-secondary > $O/bench_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_canonical -o s -- \
    python bench.py --steps 300 --warmup 30 --min-time 0.3 --no-cpu-baseline --no-secondary --chart-mode canonical > $O/bench_canonical_under_rocprof.log 2>&1
python profiles/tools/trace_percentiles.py $O/stats "k_step<float, atacom::Iiwa, 4" > $O/launch_percentiles.log
python profiles/tools/trace_percentiles.py $O/stats_canonical "k_step<float, atacom::Iiwa, 8" >> $O/launch_percentiles.log
cat $O/launch_percentiles.log
python bench.py 2>/dev/null | tail -1 > $O/bench_default.json
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_cmd.json
find $O -name '*kernel_trace.csv' -size +2M -delete; find $O -name '*.db' -delete; find $O -name '*agent_info*' -delete
du -sh $O
cut -c1-300 $O/bench_driver_cmd.json
