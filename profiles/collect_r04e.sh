#!/bin/bash
# Round 4, last GPU pass on the final build: the whole suite + smoke, the soaks of the canonical chart as it ships (iiwa third
# form, planar second), the rigid-body kernels with the parked solver state (A/B against -DATACOM_DYN_PARK=0), their kernel
# statistics / traffic / SQ counters, 8 against 4 lanes on the bench workload, the bench lines.
export TMPDIR=/tmp
O=gpurun_out/prof_r04e
rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -5 > $O/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -5 > $O/smoke.log
tail -2 $O/gpu_suite.log; tail -1 $O/smoke.log
for rep in 1 2 3; do for lib in build/ab/libatacom_nopark.so rl_on_manifold_amd/libatacom_hip.so; do
  ATACOM_LIB=$PWD/$lib MB_DYN=rigid_body_ff MB_WARM=60 MB_ROLLOUT=1 MB_LANES=4 MB_BATCHES=8192 python profiles/tools/gpu_microbench.py iiwa
done; done 2>&1 | grep -v "amdgpu.ids\|Warning\|BatchedAtacomEnv(" > $O/ab_dyn_park.log
cat $O/ab_dyn_park.log
MB_DYN=rigid_body MB_WARM=60 MB_ROLLOUT=1 MB_LANES=4,1 MB_BATCHES=8192,65536 python profiles/tools/gpu_microbench.py iiwa 2>&1 | grep -v "amdgpu.ids\|Warning\|BatchedAtacomEnv(" > $O/rigid_body.log
MB_DYN=rigid_body_ff MB_WARM=60 MB_LANES=4 MB_BATCHES=8192 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_dyn -o s -- \
    python profiles/tools/gpu_microbench.py iiwa > $O/dyn_under_rocprof.log 2>&1
W="0 8192 iiwa reference rigid_body_ff"; T=$(echo $W | tr ' ' '_')
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$T -o c -- python profiles/tools/gpu_pmc_target.py $W > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$T -o c -- python profiles/tools/gpu_pmc_target.py $W > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU \
    SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_sq_$T -o c -- python profiles/tools/gpu_pmc_target.py $W > /dev/null 2>&1
# the reference chart's step kernel on the bench workload: 8 lanes against the quad
for rep in 1 2 3; do for l in 4 8; do
  python bench.py --lanes $l --steps 300 --warmup 30 --min-time 0.5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lanes', d['config'].get('lanes_per_env'), 'us/step', round(d['ms_per_step']*1e3, 3), 'kernel us', round(d['roofline']['kernel_ms']*1e3, 3))"
done; done > $O/ab_lanes_bench.log
cat $O/ab_lanes_bench.log
MB_WARM=60 MB_CHART=canonical MB_ROLLOUT=1 MB_LANES=0,4 MB_BATCHES=8192,65536 python profiles/tools/gpu_microbench.py planar 2>&1 | grep -v amdgpu.ids > $O/planar_canonical.log
for l in 4 8; do MB_CHART=canonical python profiles/tools/gpu_sens_probe.py $l 8192 40 2>&1 | grep -v amdgpu.ids > $O/sens_soak_canonical_l$l.log; done
MB_CHART=canonical MB_DTYPE=f64 python profiles/tools/gpu_sens_probe.py 8 8192 40 2>&1 | grep -v amdgpu.ids > $O/soak_canonical_f64_l8.log
grep -c verdict $O/sens_soak_canonical_l4.log $O/sens_soak_canonical_l8.log
python bench.py 2>/dev/null | tail -1 > $O/bench_default.json
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_cmd.json
find $O -name '*kernel_trace.csv' -size +2M -delete; find $O -name '*.db' -delete; find $O -name '*agent_info*' -delete
du -sh $O
cut -c1-200 $O/bench_driver_cmd.json
