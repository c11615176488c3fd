#!/bin/bash
# Round 5: link-coordinate rigid-body kernels with and without the LDS parking of the solver state (quad), the lane kernels'
# final selection, then the dynamics tests once more.
export TMPDIR=/tmp
O=gpurun_out/prof_r05f
rm -rf $O; mkdir -p $O
for rep in 1 2 3; do
  for lib in rl_on_manifold_amd/libatacom_hip.so build/ab/libatacom_r05nopark.so; do
    ATACOM_LIB=$lib MB_DYN=rigid_body_ff MB_WARM=60 MB_ROLLOUT=1 MB_LANES=4 MB_BATCHES=8192 python profiles/tools/gpu_microbench.py iiwa 2>&1 | grep -v "amdgpu.ids\|Warning\|BatchedAtacomEnv("
  done
done > $O/ab_park.log
cat $O/ab_park.log
MB_DYN=rigid_body_ff MB_WARM=60 MB_ROLLOUT=1 MB_LANES=1 MB_BATCHES=8192,65536 python profiles/tools/gpu_microbench.py iiwa 2>&1 | grep -v "amdgpu.ids" > $O/lane.log
cat $O/lane.log
python -m pytest tests/test_gpu_dynamics.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -3
