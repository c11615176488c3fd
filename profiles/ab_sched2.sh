# iterative-ilp on atacom_iiwa.hip alone against the build that ships: single steps, T-step kernels and the policy kernels at
# 8192 environments (8 lanes and the quad), the lane kernels at 262144; three interleaved repetitions
cd /root/repo
O=gpurun_out/ab_sched; mkdir -p $O
for rep in 1 2 3; do for lib in rl_on_manifold_amd/libatacom_hip.so build/ab/libatacom_it1.so; do
  ATACOM_LIB=$PWD/$lib MB_WARM=60 MB_ROLLOUT=1 MB_LANES=8,4 MB_BATCHES=8192 python profiles/tools/gpu_microbench.py iiwa
  ATACOM_LIB=$PWD/$lib MB_WARM=20 MB_ROLLOUT=1 MB_LANES=1 MB_BATCHES=262144 python profiles/tools/gpu_microbench.py iiwa
done; done 2>&1 | grep -v amdgpu.ids > $O/ab_sched_it1.log
cat $O/ab_sched_it1.log | cut -c1-150
