# the float32 soaks once more on states and actions of another seed (23 instead of 11): canonical chart 8 and 4 lanes,
# reference chart 8 lanes (the default mappings at 8192 environments)
cd /root/repo
O=gpurun_out/soak23; rm -rf $O; mkdir -p $O
for l in 8 4; do MB_SEED=23 MB_CHART=canonical python profiles/tools/gpu_sens_probe.py $l 8192 40 2>&1 | grep -v amdgpu.ids > $O/sens_soak_canonical_l${l}_seed23.log; done
MB_SEED=23 python profiles/tools/gpu_sens_probe.py 8 8192 40 2>&1 | grep -v amdgpu.ids > $O/sens_soak_reference_l8_seed23.log
grep -h "verdict\|^==" $O/*.log | cut -c1-330
