#!/bin/bash
# Round 5: right reflectors with their scalar chain off the critical path (ATACOM_G_EARLY) against the build before, same box
export TMPDIR=/tmp
O=gpurun_out/prof_r05n; rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "nullspace or chart_on_slack or rank_deficient" 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/nullspace_tests.log
for rep in 1 2 3; do for v in r05final hip; do
  lib=build/ab/libatacom_$v.so; [ $v = hip ] && lib=rl_on_manifold_amd/libatacom_hip.so
  ATACOM_LIB=$lib python bench.py --steps 300 --warmup 30 --min-time 0.5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'us/step', round(d['ms_per_step']*1e3, 3), 'kernel us', round(d['roofline']['kernel_ms']*1e3, 3), 'T-step us', round(8192e6/d['collection']['rollout_env_steps_per_s_per_gpu'],3), 'c_max', d['max_abs_c'])"
done; done > $O/ab_gearly.log
sort -s -k1,1 $O/ab_gearly.log
for v in r05final hip; do lib=build/ab/libatacom_$v.so; [ $v = hip ] && lib=rl_on_manifold_amd/libatacom_hip.so
  ATACOM_LIB=$lib MB_WARM=30 MB_ROLLOUT=1 MB_LANES=4,8 MB_BATCHES=8192 python profiles/tools/gpu_microbench.py iiwa planar 2>&1 | grep -v amdgpu.ids; done | tee $O/ab_gearly_microbench.log
