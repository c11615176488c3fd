"""Kernel-tuning helper: time the stand-alone null-space primitive (bidiag+P, and +rref) per mapping."""
import sys, ctypes as C
import torch
sys.path.insert(0, '.')
from rl_on_manifold_amd import _lib
from rl_on_manifold_amd.engine import _ptr
lib = _lib.load()
dev = 'cuda:0'
shapes = {0: (2, 3, 1), 1: (6, 9, 3), 2: (12, 17, 5)}
ns = [int(a) for a in sys.argv[1:]] or [8192, 65536]
for env_id in (1, 2):
  c, nn, k = shapes[env_id]
  for n in ns:
    Jc = torch.randn((n, c, nn), device=dev)
    rhs = torch.randn((n, c), device=dev)
    x = torch.empty((n, nn), device=dev); nb = torch.empty((n, nn, k), device=dev); rr = torch.empty((n, nn, k), device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for lanes in (1, 4):
        for with_rref in (0, 1):
            def run():
                _lib.check(lib.atacom_nullspace(env_id, 0, lanes, n, _ptr(Jc), _ptr(rhs), 0.05, _ptr(x), _ptr(nb), _ptr(rr) if with_rref else None, st))
            for _ in range(5): run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            print('env=%d n=%d lanes=%d rref(xk)=%d: %.1f us' % (env_id, n, lanes, with_rref, e0.elapsed_time(e1) / 20 * 1e3), flush=True)
