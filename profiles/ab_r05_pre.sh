#!/bin/bash
# Round 5: the group prologue (iiwa, 8 lanes) against the build before it, same box, interleaved; then parity of the new build.
export TMPDIR=/tmp
O=gpurun_out/prof_r05b
rm -rf $O; mkdir -p $O
for rep in 1 2 3; do
  for lib in build/ab/libatacom_r05base.so rl_on_manifold_amd/libatacom_hip.so; do
    ATACOM_LIB=$lib MB_WARM=30 MB_ROLLOUT=1 MB_LANES=8 MB_BATCHES=8192 python profiles/tools/gpu_microbench.py iiwa 2>&1 | grep -v amdgpu.ids
  done
done > $O/ab_pre_microbench.log
cat $O/ab_pre_microbench.log
for rep in 1 2; do
  for lib in build/ab/libatacom_r05base.so rl_on_manifold_amd/libatacom_hip.so; do
    ATACOM_LIB=$lib python bench.py --steps 300 --warmup 30 --min-time 0.5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'us/step', round(d['ms_per_step']*1e3, 3), 'kernel us', round(d['roofline']['kernel_ms']*1e3, 3), 'c_max', d['max_abs_c'])"
  done
done > $O/ab_pre_bench.log
cat $O/ab_pre_bench.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_rollout.py tests/test_gpu_noise.py -m gpu -q -x -k "iiwa or mapping or determin" 2>&1 | grep -v amdgpu.ids | tail -8 > $O/parity.log
cat $O/parity.log
