# A/B of scheduler options on the translation unit of the headline kernels (atacom_iiwa.hip only; everything else the build
# that ships): bench workload, 8 lanes and the quad, two interleaved repetitions
cd /root/repo
O=gpurun_out/ab_sched; rm -rf $O; mkdir -p $O
for rep in 1 2; do for v in hip maxilp itilp nopost maxmem minreg o2; do
  lib=build/ab/libatacom_$v.so; [ $v = hip ] && lib=rl_on_manifold_amd/libatacom_hip.so
  for l in 8 4; do ATACOM_LIB=$PWD/$lib python bench.py --lanes $l --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$v lanes=$l  %.2f us  kernel %.2f us  c_max %.3e' % (r['ms_per_step']*1e3, r['roofline']['kernel_ms']*1e3, r.get('max_constraint_residual', r.get('c_max', float('nan')))))"; done
done; done > $O/ab_sched.log 2>&1
sort -s -k1,2 $O/ab_sched.log
