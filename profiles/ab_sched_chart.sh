# scheduler options on the canonical-chart iiwa unit (atacom_chart_iiwa.hip): bench workload (single step, 8 lanes) and the
# T-step kernels through the microbench, two interleaved repetitions
cd /root/repo
O=gpurun_out/ab_sched; mkdir -p $O
for rep in 1 2; do for v in hip maxilp itilp maxmem maxocc; do
  lib=build/ab/libatacom_ch_$v.so; [ $v = hip ] && lib=rl_on_manifold_amd/libatacom_hip.so
  ATACOM_LIB=$PWD/$lib python bench.py --chart-mode canonical --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$v canonical bench  %.2f us  collection %.2f us/step' % (r['ms_per_step']*1e3, r['collection']['rollout_ms']/120*1e3))"
done; done > $O/ab_sched_chart.log 2>&1
sort -s -k1,1 $O/ab_sched_chart.log
