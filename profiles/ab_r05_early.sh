#!/bin/bash
# Round 5: early store of the joint state in k_step (lane 0 writes q, dq, s right after the physics sub-steps) against the build
# without it; bench workload, interleaved; then the step tests.
export TMPDIR=/tmp
O=gpurun_out/prof_r05j
rm -rf $O; mkdir -p $O
for rep in 1 2 3; do
  for lib in rl_on_manifold_amd/libatacom_hip.so build/ab/libatacom_early.so; do
    ATACOM_LIB=$lib python bench.py --steps 300 --warmup 30 --min-time 0.5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'iiwa us/step', round(d['ms_per_step']*1e3, 3), 'kernel us', round(d['roofline']['kernel_ms']*1e3, 3))"
    ATACOM_LIB=$lib python bench.py --env planar --steps 300 --warmup 30 --min-time 0.5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'planar us/step', round(d['ms_per_step']*1e3, 3), 'kernel us', round(d['roofline']['kernel_ms']*1e3, 3))"
  done
done > $O/ab_early.log
cat $O/ab_early.log
ATACOM_LIB=build/ab/libatacom_early.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_rollout.py -m gpu -q -x -k "iiwa or planar" 2>&1 | grep -v amdgpu.ids | tail -3
