#!/bin/bash
# Round 5: scheduler strategies on the unit of the headline kernels, re-measured on this round's kernels (bench workload + T-step)
export TMPDIR=/tmp
O=gpurun_out/prof_r05k; rm -rf $O; mkdir -p $O
for rep in 1 2 3; do for v in hip itilp maxilp; do
  lib=build/ab/libatacom_$v.so; [ $v = hip ] && lib=rl_on_manifold_amd/libatacom_hip.so
  ATACOM_LIB=$lib python bench.py --steps 300 --warmup 30 --min-time 0.5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'us/step', round(d['ms_per_step']*1e3, 3), 'kernel us', round(d['roofline']['kernel_ms']*1e3, 3), 'T-step us', round(8192e6/d['collection']['rollout_env_steps_per_s_per_gpu'],3), 'policy us', round(8192e6/d['policy_rollout_kernel_env_steps_per_s_per_gpu'],3))"
done; done > $O/ab_sched.log
sort -s -k1,1 $O/ab_sched.log
