"""Kernel-by-kernel comparison of two builds of libatacom_hip.so (registers, scratch, LDS, code size), read from the code
objects inside the libraries -- no GPU needed.  The A/B evidence behind "option X costs nothing while it is off".

    python profiles/tools/kernel_table_diff.py build/ab/libatacom_nonoise.so rl_on_manifold_amd/libatacom_hip.so [filter]
"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_kernel_resources import _kernels      # noqa: E402

a, b = sys.argv[1], sys.argv[2]
flt = sys.argv[3] if len(sys.argv) > 3 else ''
with tempfile.TemporaryDirectory() as t1, tempfile.TemporaryDirectory() as t2:
    A = {k[0]: k[1:] for k in _kernels(t1, a)}
    B = {k[0]: k[1:] for k in _kernels(t2, b)}
same = diff = 0
print('%-62s %-28s %-28s' % ('kernel', os.path.basename(a), os.path.basename(b)))
print('%-62s %-28s %-28s' % ('', 'LDS scratch VGPR AGPR code', 'LDS scratch VGPR AGPR code'))
for name in sorted(set(A) | set(B)):
    if flt not in name:
        continue
    ra, rb = A.get(name), B.get(name)
    if ra == rb:
        same += 1
        continue
    diff += 1
    f = lambda r: '-' if r is None else '%3d %5d %4d %4d %6d' % r      # noqa: E731
    print('%-62s %-28s %-28s' % (name[:62], f(ra), f(rb)))
print('%d kernels identical in every column, %d differ' % (same, diff))
