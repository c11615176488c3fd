#!/bin/bash
# Round 5: H(0) generated once per env step (with G(0)) against generating it in every sub-step; bit-identical results expected
export TMPDIR=/tmp
O=gpurun_out/prof_r05o; rm -rf $O; mkdir -p $O
for rep in 1 2 3; do for v in noh0 hip; do
  lib=build/ab/libatacom_$v.so; [ $v = hip ] && lib=rl_on_manifold_amd/libatacom_hip.so
  ATACOM_LIB=$lib python bench.py --steps 300 --warmup 30 --min-time 0.5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'us/step', round(d['ms_per_step']*1e3, 3), 'kernel us', round(d['roofline']['kernel_ms']*1e3, 3), 'T-step us', round(8192e6/d['collection']['rollout_env_steps_per_s_per_gpu'],3), 'c_max', d['max_abs_c'], 'c_avg', d['c_avg'])"
done; done > $O/ab_h0.log
sort -s -k1,1 $O/ab_h0.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_rollout.py -m gpu -q -x -k "iiwa" 2>&1 | grep -v amdgpu.ids | tail -2
