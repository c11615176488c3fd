# What the timing at create (calibrate_step_lanes) says on this box next to the sustained figures of both mappings: bench
# workload (1000-step blocks) and the constraint-active states of profiles/tools/gpu_microbench.py.  One line per figure.
cd /root/repo
O=gpurun_out/calib; mkdir -p $O
{
  echo "box $(hostname) $(date -u +%H:%M:%S)"
  ATACOM_CALIBRATE=verbose python bench.py --no-cpu-baseline --no-secondary 2> $O/err.tmp | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('bench auto     lanes %d  %.2f us' % (r['config']['lanes_per_env'], r['ms_per_step']*1e3))"
  grep "^\[atacom\]" $O/err.tmp
  for l in 8 4; do python bench.py --lanes $l --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('bench lanes=$l          %.2f us' % (r['ms_per_step']*1e3))"; done
  MB_WARM=60 MB_LANES=8,4 MB_BATCHES=8192 python profiles/tools/gpu_microbench.py iiwa 2>&1 | grep "step" | cut -c1-60
} >> $O/calibration_probe.log 2>&1
rm -f $O/err.tmp
tail -8 $O/calibration_probe.log
