# scheduler options on the rigid-body translation unit (atacom_iiwa_dyn.hip): single step, T-step and policy kernels of the
# quad mapping at 8192 environments, two interleaved repetitions (the microbench settles before it times)
cd /root/repo
O=gpurun_out/ab_sched; mkdir -p $O
for rep in 1 2; do for v in hip maxilp minreg maxmem maxocc; do
  lib=build/ab/libatacom_dyn_$v.so; [ $v = hip ] && lib=rl_on_manifold_amd/libatacom_hip.so
  ATACOM_LIB=$PWD/$lib MB_WARM=60 MB_ROLLOUT=1 MB_DYN=rigid_body MB_LANES=4 MB_BATCHES=8192 python profiles/tools/gpu_microbench.py iiwa
done; done 2>&1 | grep -v amdgpu.ids > $O/ab_sched_dyn.log
cut -c1-120 $O/ab_sched_dyn.log
