cd /root/repo
O=gpurun_out/ab_block; rm -rf $O; mkdir -p $O
for rep in 1 2 3; do for lib in rl_on_manifold_amd/libatacom_hip.so build/ab/libatacom_bg128.so build/ab/libatacom_bg64.so; do
  ATACOM_LIB=$PWD/$lib MB_WARM=60 MB_ROLLOUT=1 MB_LANES=8,4 MB_BATCHES=8192 python profiles/tools/gpu_microbench.py iiwa
  ATACOM_LIB=$PWD/$lib MB_WARM=60 MB_ROLLOUT=1 MB_CHART=canonical MB_LANES=8 MB_BATCHES=8192 python profiles/tools/gpu_microbench.py iiwa
done; done 2>&1 | grep -v amdgpu.ids > $O/ab_block_group.log
python profiles/tools/gpu_sens_probe.py 8 8192 40 2>&1 | grep -v amdgpu.ids > $O/sens_soak_reference_l8.log
grep -c verdict $O/sens_soak_reference_l8.log
