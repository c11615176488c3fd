#!/bin/bash
# Round 5: the rigid-body kernels in link coordinates against the world-coordinate build (same box, interleaved), then the
# dynamics parity tests (G11-pinned primitives, env-step against the oracle).
export TMPDIR=/tmp
O=gpurun_out/prof_r05e
rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_dynamics.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -6 > $O/dyn_parity.log
cat $O/dyn_parity.log
for rep in 1 2; do
  for lib in build/ab/libatacom_r05relabel.so rl_on_manifold_amd/libatacom_hip.so; do
    for mode in rigid_body rigid_body_ff; do
      ATACOM_LIB=$lib MB_DYN=$mode MB_WARM=60 MB_ROLLOUT=1 MB_LANES=4,1 MB_BATCHES=8192 python profiles/tools/gpu_microbench.py iiwa 2>&1 | grep -v "amdgpu.ids\|Warning\|BatchedAtacomEnv("
    done
  done
done > $O/ab_dyn.log
cat $O/ab_dyn.log
ATACOM_LIB=rl_on_manifold_amd/libatacom_hip.so MB_DYN=rigid_body_ff MB_WARM=60 MB_ROLLOUT=1 MB_LANES=1 MB_BATCHES=65536 python profiles/tools/gpu_microbench.py iiwa 2>&1 | grep -v "amdgpu.ids" >> $O/ab_dyn.log
ATACOM_LIB=build/ab/libatacom_r05relabel.so MB_DYN=rigid_body_ff MB_WARM=60 MB_ROLLOUT=1 MB_LANES=1 MB_BATCHES=65536 python profiles/tools/gpu_microbench.py iiwa 2>&1 | grep -v "amdgpu.ids" >> $O/ab_dyn.log
tail -4 $O/ab_dyn.log
