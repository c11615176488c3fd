"""Build libatacom_hip.so in-tree with hipcc for gfx950 (no GPU needed: hipcc cross-compiles).

    python -m rl_on_manifold_amd.build [--force]

One translation unit per environment (they compile in parallel) + the C-ABI host file, linked into
rl_on_manifold_amd/libatacom_hip.so.  The .so is git-ignored but travels to the GPU box with gpurun.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.environ.get('ATACOM_LIB_OUT') or os.path.join(HERE, 'libatacom_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
ARCH = 'gfx950'
FLAGS = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function'] + \
    os.environ.get('ATACOM_HIPCC_FLAGS', '').split()
UNITS = ['atacom_iiwa.hip', 'atacom_iiwa_group.hip', 'atacom_iiwa_f64.hip', 'atacom_noise_iiwa.hip', 'atacom_noise_iiwa_f64.hip', 'atacom_iiwa_dyn.hip',
         'atacom_iiwa_dyn_f64.hip', 'atacom_iiwa_dyn_chart.hip', 'atacom_chart_iiwa.hip', 'atacom_planar.hip',
         'atacom_noise_planar.hip', 'atacom_chart.hip', 'atacom_circle.hip',
         'atacom_capi.cpp']           # longest first: the pool runs min(cores, units) compilers
# per-unit extra flags.  atacom_iiwa_group.hip holds the 4- / 8-lane float32 iiwa kernels alone so that they can take the
# iterative-ilp scheduler (-1.0 % on the headline step, profiles/r05_ab_sched.log) without the lane kernels of the same source
# paying for it (+29 % on the lane rollout kernel, r04_ab_sched_iterative_ilp.log; the option also crashes the compiler on
# float64 instantiations).  (-amdgpu-sched-strategy=max-ilp was tried per unit in round 1 -- planar step kernel -4 %, planar
# policy-rollout kernel +19 %, iiwa quad kernel +8 % -- and dropped, profiles/r01_lanes_vs_batch.md; round 5: +0.7 % on the headline)
UNIT_FLAGS = {'atacom_iiwa_group.hip': ['-mllvm', '-amdgpu-sched-strategy=iterative-ilp']}


def _sources():
    out = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    out.append(os.path.join(os.path.dirname(HERE), 'include', 'atacom_hip.h'))
    return out


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in _sources())


# kernel-tuning builds: ATACOM_KEEP_OBJ=1 keeps the objects of a build; ATACOM_ONLY_UNITS=a.hip,b.hip then recompiles only
# those units (with ATACOM_HIPCC_FLAGS applied to them alone) and links against the kept objects of the others
ONLY = [u for u in os.environ.get('ATACOM_ONLY_UNITS', '').split(',') if u]


def _compile(unit):
    src = os.path.join(CSRC, unit)
    tag = os.environ.get('ATACOM_OBJ_TAG', '')
    obj = os.path.join(CSRC, os.path.splitext(unit)[0] + tag + '.o')
    if ONLY and unit not in ONLY:
        kept = os.path.join(CSRC, os.path.splitext(unit)[0] + os.environ.get('ATACOM_BASE_TAG', '_keep') + '.o')
        if not os.path.exists(kept):
            raise RuntimeError('ATACOM_ONLY_UNITS needs the kept object %s (build once with ATACOM_KEEP_OBJ=1 ATACOM_OBJ_TAG=_keep)' % kept)
        return kept
    cmd = [HIPCC] + FLAGS + UNIT_FLAGS.get(unit, []) + (['-x', 'hip'] if unit.endswith('.cpp') else []) + ['-c', src, '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (unit, ' '.join(cmd), r.stderr[-4000:]))
    return obj


# The kernels run at the edge of the register file and two compiler defects have been met there (a miscompiled float64
# instantiation in round 1; register copies under a narrowed exec mask, profiles/r06_exec_mask_copies.md).  The shipped code is
# validated with THIS compiler; another one is not refused -- tests/test_kernel_resources.py audits whatever it produced --
# but it is named.
VALIDATED_HIPCC = 'HIP version: 7.2'


def _hipcc_version():
    try:
        out = subprocess.run([HIPCC, '--version'], capture_output=True, text=True).stdout
        return next((ln.strip() for ln in out.splitlines() if ln.startswith('HIP version')), out.strip()[:60])
    except OSError as e:
        return 'unavailable (%s)' % e


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    ver = _hipcc_version()
    if not ver.startswith(VALIDATED_HIPCC):
        print('[atacom] WARNING: building with "%s"; the kernels were validated with hipcc 7.2 -- run the code-object audits '
              '(python -m pytest tests/test_kernel_resources.py) and the GPU suite before trusting this build' % ver, flush=True)
    if verbose:
        print('[atacom] building %s for %s ...' % (os.path.basename(LIB), ARCH), flush=True)
    with ThreadPoolExecutor(max_workers=min(len(UNITS), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(_compile, UNITS))
    cmd = [HIPCC, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n%s\n%s' % (' '.join(cmd), r.stderr[-4000:]))
    if not os.environ.get('ATACOM_KEEP_OBJ'):
        for o in objs:
            if not (ONLY and o.endswith(os.environ.get('ATACOM_BASE_TAG', '_keep') + '.o')):
                os.remove(o)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
