#!/bin/bash
# Round-4 evidence on the final build (run through gpurun from the repo root):
#   gpurun --timeout 3600 -- 'bash profiles/collect_r04.sh'
# then, back in the build container:  python profiles/summarize_r04.py   (writes the tracked files under profiles/).
# Counter passes are separate from each other and use only --kernel-trace next to --pmc.
export TMPDIR=/tmp
O=gpurun_out/prof_r04
rm -rf $O; mkdir -p $O
# 0. the whole GPU suite and the smoke entry point
python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -5 > $O/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -5 > $O/smoke.log
tail -2 $O/gpu_suite.log; tail -1 $O/smoke.log
# 1. kernel trace + stats of the headline bench command, the canonical-chart headline, configs 2 and 3, the rigid-body mode
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- \
    python bench.py --steps 300 --warmup 30 --min-time 0.3 --no-cpu-baseline --no-secondary > $O/bench_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_canonical -o s -- \
    python bench.py --steps 300 --warmup 30 --min-time 0.3 --no-cpu-baseline --no-secondary --chart-mode canonical > $O/bench_canonical_under_rocprof.log 2>&1
python profiles/tools/trace_percentiles.py $O/stats "k_step<float, atacom::Iiwa, 4" > $O/launch_percentiles.log
python profiles/tools/trace_percentiles.py $O/stats_canonical "k_step<float, atacom::Iiwa, 8" >> $O/launch_percentiles.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_planar -o s -- \
    python bench.py --env planar --steps 300 --warmup 30 --min-time 0.3 --no-cpu-baseline --no-secondary > $O/bench_planar_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_circle -o s -- \
    python bench.py --env circle --batch 4096 --steps 300 --warmup 30 --min-time 0.3 --no-cpu-baseline --no-secondary > $O/bench_circle_under_rocprof.log 2>&1
MB_DYN=rigid_body_ff MB_WARM=60 MB_LANES=4 MB_BATCHES=8192 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_dyn -o s -- \
    python profiles/tools/gpu_microbench.py iiwa > $O/dyn_under_rocprof.log 2>&1
# 2. HBM traffic of the step kernel, separate FETCH / WRITE passes, per workload
for W in "0 8192 iiwa reference kinematic" "0 8192 iiwa canonical kinematic" "0 8192 iiwa reference rigid_body_ff"; do
  T=$(echo $W | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$T -o c -- python profiles/tools/gpu_pmc_target.py $W > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$T -o c -- python profiles/tools/gpu_pmc_target.py $W > /dev/null 2>&1
done
# 3. SQ counters: canonical chart (third form) at 8 and 4 lanes, the reference chart, the rigid-body kernel
for W in "0 8192 iiwa canonical kinematic" "4 8192 iiwa canonical kinematic" "0 8192 iiwa reference kinematic" "0 8192 iiwa reference rigid_body_ff" "0 8192 planar canonical kinematic"; do
  T=$(echo $W | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU \
      SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_sq_$T -o c -- python profiles/tools/gpu_pmc_target.py $W > /dev/null 2>&1
done
# 4. mapping vs batch in canonical mode, the reference chart beside it, the rigid-body modes
MB_WARM=60 MB_CHART=canonical MB_ROLLOUT=1 MB_LANES=0,1,2,4,8 MB_BATCHES=1024,8192,16384,65536,262144 python profiles/tools/gpu_microbench.py iiwa planar 2>&1 | grep -v amdgpu.ids > $O/lanes_vs_batch_canonical.log
MB_WARM=60 MB_ROLLOUT=1 MB_LANES=0 MB_BATCHES=8192,65536,262144 python profiles/tools/gpu_microbench.py iiwa planar circle 2>&1 | grep -v amdgpu.ids > $O/lanes_vs_batch_reference.log
MB_DYN=rigid_body MB_WARM=60 MB_ROLLOUT=1 MB_LANES=4,1 MB_BATCHES=8192 python profiles/tools/gpu_microbench.py iiwa 2>&1 | grep -v "amdgpu.ids\|Warning\|BatchedAtacomEnv(" > $O/rigid_body.log
MB_DYN=rigid_body_ff MB_WARM=60 MB_LANES=4 MB_BATCHES=8192 python profiles/tools/gpu_microbench.py iiwa 2>&1 | grep -v amdgpu.ids >> $O/rigid_body.log
# 5. third form of the canonical group kernels against the second (-DATACOM_CHART_FORM=2 build), same box, interleaved
if [ -f build/ab/libatacom_form2.so ]; then
  for rep in 1 2; do for lib in build/ab/libatacom_form2.so rl_on_manifold_amd/libatacom_hip.so; do
    ATACOM_LIB=$PWD/$lib MB_WARM=60 MB_ROLLOUT=1 MB_CHART=canonical MB_LANES=8,4,2 MB_BATCHES=8192 python profiles/tools/gpu_microbench.py iiwa planar
  done; done 2>&1 | grep -v amdgpu.ids > $O/ab_chart_form.log
fi
# 6. where the time of one canonical-chart launch goes (tuning build with wall-clock stamps and path counters)
if [ -f build/ts/libatacom_ts.so ]; then
  for c in reference canonical; do for l in 4 8; do
    echo "== $c chart, $l lanes, bench workload"; ATACOM_LIB=$PWD/build/ts/libatacom_ts.so MB_CHART=$c python profiles/tools/gpu_bench_probe.py $l
    echo "== $c chart, $l lanes, quiet states"; ATACOM_LIB=$PWD/build/ts/libatacom_ts.so MB_CHART=$c python profiles/tools/gpu_phase_probe.py $l
    echo "== $c chart, $l lanes, constraint-active states"; ATACOM_LIB=$PWD/build/ts/libatacom_ts.so MB_RANDOM=1 MB_WARM=60 MB_CHART=$c python profiles/tools/gpu_phase_probe.py $l
  done; done 2>&1 | grep -v amdgpu.ids > $O/phase_probe.log
fi
# 7. soaks of the canonical chart's final kernels: float32 against the float64 specification (sensitivity rule), float64 errors
for l in 4 8; do MB_CHART=canonical python profiles/tools/gpu_sens_probe.py $l 8192 40 2>&1 | grep -v amdgpu.ids > $O/sens_soak_canonical_l$l.log; done
MB_CHART=canonical MB_DTYPE=f64 python profiles/tools/gpu_sens_probe.py 8 8192 40 2>&1 | grep -v amdgpu.ids > $O/soak_canonical_f64_l8.log
python profiles/tools/gpu_sens_probe.py 4 8192 40 2>&1 | grep -v amdgpu.ids > $O/sens_soak_reference_l4.log
# 8. the bench lines: default, the driver's command
python bench.py 2>/dev/null | tail -1 > $O/bench_default.json
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_cmd.json
find $O -name '*kernel_trace.csv' -size +2M -delete; find $O -name '*.db' -delete; find $O -name '*agent_info*' -delete
du -sh $O
cat $O/launch_percentiles.log
cut -c1-300 $O/bench_driver_cmd.json
