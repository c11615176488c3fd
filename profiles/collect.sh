#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 900 -- 'bash profiles/collect.sh'
# then, back in the build container:  python profiles/summarize.py   (writes the tracked files under profiles/).
# Counter passes are separate from each other and use only --kernel-trace next to --pmc.
export TMPDIR=/tmp
O=gpurun_out/prof
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_quad -o s -- \
    python bench.py --steps 300 --warmup 30 --no-cpu-baseline > $O/bench_under_rocprof_quad.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_lane -o s -- \
    python bench.py --steps 300 --warmup 30 --no-cpu-baseline --lanes 1 --batch 65536 > $O/bench_under_rocprof_lane.log 2>&1
for L in 1 4; do
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU \
      SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_sq_l$L -o c -- python profiles/tools/gpu_pmc_target.py $L > /dev/null 2>&1
done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o c -- python profiles/tools/gpu_pmc_target.py 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o c -- python profiles/tools/gpu_pmc_target.py 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES \
    SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/pmc_policy -o c -- python profiles/tools/gpu_pmc_policy.py iiwa 8192 > /dev/null 2>&1
python bench.py 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench.json
ls -R $O | head -60
cut -c1-600 $O/bench.json
