#!/bin/bash
# Round 4, the build that ships (single steps on 8 lanes up to 8192 iiwa environments; rigid-body parking for lane groups only):
# the whole suite + smoke, the headline's kernel statistics / percentiles / traffic / SQ counters, 8 lanes against the quad once
# more on this box, the rigid-body modes, the bench lines.
export TMPDIR=/tmp
O=gpurun_out/prof_r04f
rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -5 > $O/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -5 > $O/smoke.log
tail -2 $O/gpu_suite.log; tail -1 $O/smoke.log
for rep in 1 2 3; do for l in 4 8; do
  python bench.py --lanes $l --steps 300 --warmup 30 --min-time 0.5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lanes', d['config'].get('lanes_per_env'), 'us/step', round(d['ms_per_step']*1e3, 3), 'kernel us', round(d['roofline']['kernel_ms']*1e3, 3))"
done; done > $O/ab_lanes_bench.log
cat $O/ab_lanes_bench.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- \
    python bench.py --steps 300 --warmup 30 --min-time 0.3 --no-cpu-baseline --no-secondary > $O/bench_under_rocprof.log 2>&1
python profiles/tools/trace_percentiles.py $O/stats "k_step<float, atacom::Iiwa, 8" > $O/launch_percentiles.log
cat $O/launch_percentiles.log
W="0 8192 iiwa reference kinematic"; T=$(echo $W | tr ' ' '_')
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$T -o c -- python profiles/tools/gpu_pmc_target.py $W > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$T -o c -- python profiles/tools/gpu_pmc_target.py $W > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU \
    SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_sq_$T -o c -- python profiles/tools/gpu_pmc_target.py $W > /dev/null 2>&1
MB_DYN=rigid_body MB_WARM=60 MB_ROLLOUT=1 MB_LANES=4,1 MB_BATCHES=8192,65536 python profiles/tools/gpu_microbench.py iiwa 2>&1 | grep -v "amdgpu.ids\|Warning\|BatchedAtacomEnv(" > $O/rigid_body.log
MB_DYN=rigid_body_ff MB_WARM=60 MB_ROLLOUT=1 MB_LANES=4 MB_BATCHES=8192 python profiles/tools/gpu_microbench.py iiwa 2>&1 | grep -v amdgpu.ids >> $O/rigid_body.log
python bench.py 2>/dev/null | tail -1 > $O/bench_default.json
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_cmd.json
find $O -name '*kernel_trace.csv' -size +2M -delete; find $O -name '*.db' -delete; find $O -name '*agent_info*' -delete
du -sh $O
cut -c1-200 $O/bench_driver_cmd.json
