#!/bin/bash
# Round 6 A/B: the policy network of the 8-lane rollout kernel on the vector ALU (8 hidden units per lane; default build) against
# the matrix-core form (one GEMM block per wave, half of it padding; build/ab/libatacom_mfma8.so = -DATACOM_MLP8_VALU=0),
# interleaved on one box.     gpurun -- 'bash profiles/ab_r06_mlp8.sh'
for rep in 1 2 3; do
  for lib in "" build/ab/libatacom_mfma8.so; do
    ATACOM_LIB=$lib MB_WARM=30 MB_ROLLOUT=1 MB_LANES=8 MB_BATCHES=4096,8192 python profiles/tools/gpu_microbench.py iiwa planar 2>&1 | grep -v amdgpu.ids | grep -A1 "lanes=8" | sed "s|^|[${lib:-valu8 (default)}] |"
  done
done
