"""Per-kernel register / scratch / occupancy table of one translation unit, from hipcc's own resource remarks.

    python profiles/tools/kernel_resources.py rl_on_manifold_amd/csrc/atacom_chart_iiwa.hip [filter]
"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
r = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c', src, '-o', '/dev/null',
                    '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True)
blocks = re.split(r'remark: [^\n]*Function Name: ', r.stderr)[1:]
names = [b.split('\n')[0].strip(' []') for b in blocks]
dem = subprocess.run(['c++filt'] + names, capture_output=True, text=True).stdout.strip().split('\n')


def g(b, k):
    m = re.search(k + r': (\d+)', b)
    return m.group(1) if m else '?'


for b, dn in zip(blocks, dem):
    dn = re.sub(r'atacom::', '', dn).split('(')[0]
    if flt in dn:
        print('%-70s VGPR %3s AGPR %3s scratch %5s occ %s' % (dn[:70], g(b, 'VGPRs'), g(b, 'AGPRs'),
                                                           g(b, r'ScratchSize \[bytes/lane\]'),
                                                           g(b, r'Occupancy \[waves/SIMD\]')))
