#!/bin/bash
# Round 5, rigid-body kernels as they ship: link coordinates, no parking, no phase fences.  Timing per mapping + the dynamics tests.
export TMPDIR=/tmp
O=gpurun_out/prof_r05g
rm -rf $O; mkdir -p $O
for mode in rigid_body rigid_body_ff; do
  MB_DYN=$mode MB_WARM=60 MB_ROLLOUT=1 MB_LANES=4,1 MB_BATCHES=8192 python profiles/tools/gpu_microbench.py iiwa 2>&1 | grep -v "amdgpu.ids\|Warning\|BatchedAtacomEnv("
done > $O/rigid_body.log
MB_DYN=rigid_body_ff MB_WARM=60 MB_ROLLOUT=1 MB_LANES=0 MB_BATCHES=16384,65536 python profiles/tools/gpu_microbench.py iiwa 2>&1 | grep -v "amdgpu.ids" >> $O/rigid_body.log
cat $O/rigid_body.log
python -m pytest tests/test_gpu_dynamics.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/dyn_tests.log
