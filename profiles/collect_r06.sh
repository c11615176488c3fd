#!/bin/bash
# Round-6 evidence on the shipped build: kernel trace + stats of the bench command (configs 4, 3, 2 and the float64 record),
# launch percentiles, FETCH / WRITE / SQ counter passes (separate runs, --kernel-trace only beside --pmc) for configs 4, 3, 2,
# the float64 headline (its traffic was null) and the saturation batch (65536 iiwa environments, one lane
# per environment), mapping-vs-batch tables in both precisions, bench lines.
#     gpurun --timeout 2400 -- 'bash profiles/collect_r06.sh [tag]'
export TMPDIR=/tmp
TAG=${1:-r06}
O=gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- \
    python bench.py --steps 300 --warmup 30 --min-time 0.3 --no-cpu-baseline --no-secondary > $O/bench_under_rocprof.log 2>&1
python profiles/tools/trace_percentiles.py $O/stats "k_step<float, atacom::Iiwa, 8" > $O/launch_percentiles.log
cat $O/launch_percentiles.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_planar -o s -- \
    python bench.py --env planar --steps 300 --warmup 30 --min-time 0.3 --no-cpu-baseline --no-secondary > $O/bench_planar_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_circle -o s -- \
    python bench.py --env circle --batch 4096 --steps 300 --warmup 30 --min-time 0.3 --no-cpu-baseline --no-secondary > $O/bench_circle_under_rocprof.log 2>&1
MB_DTYPE=f64 MB_WARM=60 MB_LANES=0 MB_BATCHES=8192 MB_ROLLOUT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_f64 -o s -- \
    python profiles/tools/gpu_microbench.py iiwa > $O/f64_under_rocprof.log 2>&1
MB_WARM=60 MB_LANES=0 MB_BATCHES=65536 MB_ROLLOUT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_sat -o s -- \
    python profiles/tools/gpu_microbench.py iiwa > $O/sat_under_rocprof.log 2>&1
for W in "0 8192 iiwa reference kinematic f32" "0 8192 planar reference kinematic f32" "0 4096 circle reference kinematic f32" \
         "0 8192 iiwa reference kinematic f64" "0 65536 iiwa reference kinematic f32"; do
  T=$(echo $W | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$T -o c -- python profiles/tools/gpu_pmc_target.py $W > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$T -o c -- python profiles/tools/gpu_pmc_target.py $W > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU \
      SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_sq_$T -o c -- python profiles/tools/gpu_pmc_target.py $W > /dev/null 2>&1
done
MB_WARM=60 MB_ROLLOUT=1 MB_LANES=0 MB_BATCHES=4096,8192,16384,65536,262144 python profiles/tools/gpu_microbench.py iiwa planar 2>&1 | grep -v amdgpu.ids > $O/lanes_vs_batch_reference.log
MB_DTYPE=f64 MB_WARM=60 MB_ROLLOUT=1 MB_LANES=0 MB_BATCHES=4096,8192,16384,65536 python profiles/tools/gpu_microbench.py iiwa planar 2>&1 | grep -v amdgpu.ids > $O/lanes_vs_batch_f64.log
cat $O/lanes_vs_batch_reference.log $O/lanes_vs_batch_f64.log
( time python bench.py 2>/dev/null | tail -1 > $O/bench_default.json ) 2> $O/bench_default.time
( time python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_cmd.json ) 2> $O/bench_driver_cmd.time
cat $O/bench_driver_cmd.time
find $O -name '*kernel_trace.csv' -size +2M -delete; find $O -name '*.db' -delete; find $O -name '*agent_info*' -delete
du -sh $O
cut -c1-300 $O/bench_driver_cmd.json
