cd /root/repo
ATACOM_LIB=$PWD/build/ab/libatacom_c3m.so python tests/gpu_chart_soak_debug.py planar 4 8192 36 2>&1 | grep -v amdgpu.ids | tail -12 > gpurun_out/dbg_c3m.log
tail -2 gpurun_out/dbg_c3m.log
