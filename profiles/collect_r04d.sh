#!/bin/bash
# Round 4: the step-server experiment (tests + measurement, both transports); everything that can spin is under `timeout`.
export TMPDIR=/tmp
O=gpurun_out/r04d
rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_server.py -q 2>&1 | grep -v amdgpu.ids | tail -150 > $O/server_tests.log
tail -5 $O/server_tests.log
timeout 400 python tests/gpu_server_bench.py 2>&1 | grep -v amdgpu.ids > $O/server_bench.log
MB_CHART=canonical timeout 400 python tests/gpu_server_bench.py 2>&1 | grep -v amdgpu.ids >> $O/server_bench.log
MB_ENV=planar timeout 400 python tests/gpu_server_bench.py 2>&1 | grep -v amdgpu.ids >> $O/server_bench.log
cat $O/server_bench.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rollout.py tests/test_gpu_defend.py -q -k "full_size or planar or defend" 2>&1 | grep -v amdgpu.ids | tail -8 > $O/planar_tests.log
tail -3 $O/planar_tests.log
