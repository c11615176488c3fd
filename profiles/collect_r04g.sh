#!/bin/bash
# Round 4, after the fix of the third form's minimum-norm part (z by forward substitution; the planar task on the third form
# again): the whole suite + smoke, the soaks of the canonical chart, its kernel statistics / percentiles / traffic / counters,
# mapping vs batch, the bench lines of the build that ships.
export TMPDIR=/tmp
O=gpurun_out/prof_r04g
rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -5 > $O/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -5 > $O/smoke.log
tail -2 $O/gpu_suite.log; tail -1 $O/smoke.log
for l in 4 8; do MB_CHART=canonical python profiles/tools/gpu_sens_probe.py $l 8192 40 2>&1 | grep -v amdgpu.ids > $O/sens_soak_canonical_l$l.log; done
MB_CHART=canonical MB_DTYPE=f64 python profiles/tools/gpu_sens_probe.py 8 8192 40 2>&1 | grep -v amdgpu.ids > $O/soak_canonical_f64_l8.log
grep -c verdict $O/sens_soak_canonical_l4.log $O/sens_soak_canonical_l8.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_canonical -o s -- \
    python bench.py --steps 300 --warmup 30 --min-time 0.3 --no-cpu-baseline --no-secondary --chart-mode canonical > $O/bench_canonical_under_rocprof.log 2>&1
python profiles/tools/trace_percentiles.py $O/stats_canonical "k_step<float, atacom::Iiwa, 8" > $O/launch_percentiles_canonical.log
cat $O/launch_percentiles_canonical.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_planar_canonical -o s -- \
    python bench.py --env planar --steps 300 --warmup 30 --min-time 0.3 --no-cpu-baseline --no-secondary --chart-mode canonical > $O/bench_planar_canonical_under_rocprof.log 2>&1
for W in "0 8192 iiwa canonical kinematic" "0 8192 planar canonical kinematic"; do
  T=$(echo $W | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$T -o c -- python profiles/tools/gpu_pmc_target.py $W > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$T -o c -- python profiles/tools/gpu_pmc_target.py $W > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU \
      SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_sq_$T -o c -- python profiles/tools/gpu_pmc_target.py $W > /dev/null 2>&1
done
MB_WARM=60 MB_CHART=canonical MB_ROLLOUT=1 MB_LANES=0,1,2,4,8 MB_BATCHES=1024,8192,16384,65536,262144 python profiles/tools/gpu_microbench.py iiwa planar 2>&1 | grep -v amdgpu.ids > $O/lanes_vs_batch_canonical.log
python bench.py 2>/dev/null | tail -1 > $O/bench_default.json
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_cmd.json
find $O -name '*kernel_trace.csv' -size +2M -delete; find $O -name '*.db' -delete; find $O -name '*agent_info*' -delete
du -sh $O
cut -c1-200 $O/bench_driver_cmd.json
