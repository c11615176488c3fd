cd /root/repo
ATACOM_LIB=$PWD/build/ab/libatacom_fz.so python profiles/tools/gpu_chart_soak_debug.py planar 4 8192 40 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/dbg_fz_planar.log &
ATACOM_LIB=$PWD/build/ab/libatacom_fz.so python profiles/tools/gpu_chart_soak_debug.py iiwa 8 8192 40 2>&1 | grep -v amdgpu.ids | tail -12 > gpurun_out/dbg_fz_iiwa.log &
wait
ATACOM_LIB=$PWD/build/ab/libatacom_fz.so MB_WARM=60 MB_ROLLOUT=1 MB_CHART=canonical MB_LANES=8,4 MB_BATCHES=8192 python profiles/tools/gpu_microbench.py iiwa planar 2>&1 | grep -v amdgpu.ids > gpurun_out/dbg_fz_speed.log
echo done
