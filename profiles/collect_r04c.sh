#!/bin/bash
# Round 4, third GPU pass: the step-server experiment (tests + measurement), the parity tests added this round, the rigid-body
# kernels after the joint-7 change.  Everything that can spin is under `timeout`.
export TMPDIR=/tmp
O=gpurun_out/r04c
rm -rf $O; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_server.py -x -q 2>&1 | grep -v amdgpu.ids | tail -40 > $O/server_tests.log
tail -5 $O/server_tests.log
timeout 300 python tests/gpu_server_bench.py 2>&1 | grep -v amdgpu.ids > $O/server_bench.log
MB_CHART=canonical timeout 300 python tests/gpu_server_bench.py 2>&1 | grep -v amdgpu.ids >> $O/server_bench.log
MB_ENV=planar timeout 300 python tests/gpu_server_bench.py 2>&1 | grep -v amdgpu.ids >> $O/server_bench.log
cat $O/server_bench.log
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rollout.py tests/test_gpu_dynamics.py tests/test_gpu_noise.py -q -s 2>&1 | grep -v amdgpu.ids | grep -v "^$" | tail -120 > $O/parity_tests.log
tail -4 $O/parity_tests.log
for rep in 1 2; do
  MB_DYN=rigid_body MB_WARM=60 MB_ROLLOUT=1 MB_LANES=4,1 MB_BATCHES=8192 python tests/gpu_microbench.py iiwa
  MB_DYN=rigid_body_ff MB_WARM=60 MB_LANES=4 MB_BATCHES=8192 python tests/gpu_microbench.py iiwa
done 2>&1 | grep -v amdgpu.ids > $O/rigid_body.log
cat $O/rigid_body.log
du -sh $O
