#!/bin/bash
# Round-2 evidence, collected on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 2400 -- 'bash profiles/collect_r02.sh'
# then, back in the build container:  python profiles/summarize_r02.py   (writes the tracked files under profiles/).
# Counter passes are separate from each other and use only --kernel-trace next to --pmc.
export TMPDIR=/tmp
O=gpurun_out/prof_r02
rm -rf $O; mkdir -p $O
# 1. kernel trace + stats of the headline bench command (quad mapping, 8192 envs)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- \
    python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary > $O/bench_under_rocprof.log 2>&1
# 2. SQ counters, lanes 4 and 8 at 8192 envs (and the clock: GRBM_GUI_ACTIVE)
for L in 4 8; do
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU \
      SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_sq_l$L -o c -- python profiles/tools/gpu_pmc_target.py $L > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_grbm_l$L -o c -- python profiles/tools/gpu_pmc_target.py $L > /dev/null 2>&1
done
# 3. HBM traffic of the dominant kernel (separate passes)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o c -- python profiles/tools/gpu_pmc_target.py 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o c -- python profiles/tools/gpu_pmc_target.py 0 > /dev/null 2>&1
# 4. mapping vs batch
MB_LANES=8,4,2,1 MB_BATCHES=1024,4096,8192,16384,32768,65536 python profiles/tools/gpu_microbench.py iiwa planar 2>&1 | grep -v amdgpu.ids > $O/lanes_vs_batch.log
MB_DYN=rigid_body MB_LANES=4,1 MB_BATCHES=8192,65536 python profiles/tools/gpu_microbench.py iiwa 2>&1 | grep -v amdgpu.ids > $O/rigid_body.log
# 5. parity calibration
python profiles/tools/gpu_sens_probe.py 4 2048 40 2>&1 | grep -v amdgpu.ids > $O/sens_l4.log
python profiles/tools/gpu_sens_probe.py 8 2048 40 2>&1 | grep -v amdgpu.ids > $O/sens_l8.log
# 6. the bench lines: default, the driver's command, two ranks sharing the GPU over gloo
python bench.py 2>/dev/null | tail -1 > $O/bench_default.json
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_cmd.json
BENCH_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/bench_gloo2.json
ls -R $O | head -50
cut -c1-400 $O/bench_driver_cmd.json
