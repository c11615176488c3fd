#!/bin/bash
# Round-3 evidence, collected on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 2400 -- 'bash profiles/collect_r03.sh'
# then, back in the build container:  python profiles/summarize_r03.py   (writes the tracked files under profiles/).
# Counter passes are separate from each other and use only --kernel-trace next to --pmc.
export TMPDIR=/tmp
O=gpurun_out/prof_r03
rm -rf $O; mkdir -p $O
# 1. kernel trace + stats of the headline bench command (default = reference chart), and of the canonical-chart headline
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- \
    python bench.py --steps 300 --warmup 30 --min-time 0.3 --no-cpu-baseline --no-secondary > $O/bench_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_canonical -o s -- \
    python bench.py --steps 300 --warmup 30 --min-time 0.3 --no-cpu-baseline --no-secondary --chart-mode canonical > $O/bench_canonical_under_rocprof.log 2>&1
# 2. BASELINE configs 2 and 3: kernel stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_planar -o s -- \
    python bench.py --env planar --steps 300 --warmup 30 --min-time 0.3 --no-cpu-baseline --no-secondary > $O/bench_planar_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_circle -o s -- \
    python bench.py --env circle --batch 4096 --steps 300 --warmup 30 --min-time 0.3 --no-cpu-baseline --no-secondary > $O/bench_circle_under_rocprof.log 2>&1
# 3. HBM traffic of the step kernel, separate FETCH / WRITE passes, per workload
for W in "0 8192 iiwa reference" "0 8192 iiwa canonical" "0 8192 planar reference" "0 4096 circle reference"; do
  T=$(echo $W | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$T -o c -- python profiles/tools/gpu_pmc_target.py $W > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$T -o c -- python profiles/tools/gpu_pmc_target.py $W > /dev/null 2>&1
done
# 4. SQ counters: canonical chart (library mapping at 8192 and 65536 envs, one env per lane at 8192), planar, and the
#    reference chart again
for W in "0 8192 iiwa canonical" "0 65536 iiwa canonical" "1 8192 iiwa canonical" "0 8192 iiwa reference" "0 8192 planar reference" "0 8192 planar canonical"; do
  T=$(echo $W | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU \
      SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_sq_$T -o c -- python profiles/tools/gpu_pmc_target.py $W > /dev/null 2>&1
done
# 5. mapping vs batch in canonical mode (and the reference chart beside it on the same box)
MB_WARM=60 MB_CHART=canonical MB_ROLLOUT=1 MB_LANES=0,1,2,4,8 MB_BATCHES=1024,8192,16384,32768,65536,262144 python profiles/tools/gpu_microbench.py iiwa planar 2>&1 | grep -v amdgpu.ids > $O/lanes_vs_batch_canonical.log
MB_WARM=60 MB_ROLLOUT=1 MB_LANES=0 MB_BATCHES=8192,32768,65536,262144 python profiles/tools/gpu_microbench.py iiwa planar circle 2>&1 | grep -v amdgpu.ids > $O/lanes_vs_batch_reference.log
MB_CHART=canonical MB_ROLLOUT=1 MB_LANES=1 MB_BATCHES=4096,1048576 python profiles/tools/gpu_microbench.py circle 2>&1 | grep -v amdgpu.ids >> $O/lanes_vs_batch_canonical.log
MB_DYN=rigid_body MB_ROLLOUT=1 MB_LANES=4,1 MB_BATCHES=8192 python profiles/tools/gpu_microbench.py iiwa 2>&1 | grep -v amdgpu.ids > $O/rigid_body.log
MB_DYN=rigid_body_ff MB_LANES=4 MB_BATCHES=8192 python profiles/tools/gpu_microbench.py iiwa 2>&1 | grep -v amdgpu.ids >> $O/rigid_body.log
# 5b. where the time of one canonical-chart launch goes (tuning build with wall-clock stamps and path counters, if present):
#     quiet states (zero action from the reset pose) and constraint-active ones (perturbed start, random actions, 60 steps in)
if [ -f build/ts/libatacom_ts.so ]; then
  for c in reference canonical; do for l in 1 4 8; do
    echo "== $c chart, $l lanes, quiet states"; ATACOM_LIB=$PWD/build/ts/libatacom_ts.so MB_CHART=$c python profiles/tools/gpu_phase_probe.py $l
    echo "== $c chart, $l lanes, constraint-active states"; ATACOM_LIB=$PWD/build/ts/libatacom_ts.so MB_RANDOM=1 MB_WARM=60 MB_CHART=$c python profiles/tools/gpu_phase_probe.py $l
  done; done 2>&1 | grep -v amdgpu.ids > $O/phase_probe.log
fi
# 6. the bench lines: default, the driver's command
python bench.py 2>/dev/null | tail -1 > $O/bench_default.json
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_cmd.json
# keep the summaries, drop the per-launch traces (the merge back to the build container is capped at 64 MiB)
find $O -name '*kernel_trace.csv' -size +2M -delete; find $O -name '*.db' -delete; find $O -name '*agent_info*' -delete
du -sh $O
ls -R $O | head -60
cut -c1-600 $O/bench_driver_cmd.json
