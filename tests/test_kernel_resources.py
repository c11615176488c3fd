"""What the built library's kernels occupy, read from the code objects inside libatacom_hip.so (no GPU needed).

DESIGN.md section 6 rests on "matrices live in registers: no scratch, no LDS" for the float32 production kernels.  Both
have been lost to the optimiser without a functional symptom: a select chain rewritten into a dynamically indexed private
array went to SCRATCH in round 1 (planar 31 -> 17 us once gone) and was PROMOTED TO LDS in round 3, where addressing it
made every wave read the dispatch packet in host memory (2 - 26 us per launch).  This test pins the property."""
import os
import re
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'


def _kernels(tmp, so=None):
    """[(demangled name, LDS bytes, scratch bytes per lane, VGPRs, AGPRs, code bytes)] of every gfx950 kernel in the library."""
    if so is None:
        from rl_on_manifold_amd import build
        so = build.build(verbose=False)
    fat = os.path.join(tmp, 'fat.bin')
    subprocess.check_call([os.path.join(LLVM, 'llvm-objcopy'), '--dump-section', '.hip_fatbin=' + fat, so,
                           os.path.join(tmp, 'copy.so')])
    data = open(fat, 'rb').read()
    rows = []
    for m in re.finditer(MAGIC, data):                       # one bundle per translation unit
        p = m.start()
        n_entries = struct.unpack_from('<Q', data, p + 24)[0]
        off = p + 32
        for _ in range(n_entries):
            o, size, id_len = struct.unpack_from('<QQQ', data, off)
            ident = data[off + 24:off + 24 + id_len].decode()
            off += 24 + id_len
            if 'gfx950' not in ident or size == 0:
                continue
            elf = os.path.join(tmp, 'dev%d.elf' % p)
            open(elf, 'wb').write(data[p + o:p + o + size])
            notes = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', elf], capture_output=True, text=True,
                                   check=True).stdout
            syms = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '-sW', elf], capture_output=True, text=True,
                                  check=True).stdout
            size = {ln.split()[-1]: int(ln.split()[2]) for ln in syms.split('\n') if ' FUNC ' in ln}
            for blk in notes.split('  - .agpr_count:')[1:]:
                g = lambda k: re.search(r'\.' + k + r':\s+(\S+)', blk).group(1)      # noqa: E731
                rows.append((g('name'), int(g('group_segment_fixed_size')), int(g('private_segment_fixed_size')),
                             int(g('vgpr_count')), int(blk.split()[0]), size.get(g('name'), 0)))
    names = subprocess.run(['c++filt'] + [r[0] for r in rows], capture_output=True, text=True).stdout.strip().split('\n')
    short = [re.sub(r'\(.*', '', n).replace('atacom::', '').replace('void ', '') for n in names]
    return [(s,) + r[1:] for s, r in zip(short, rows)]


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, 'llvm-readelf')), reason='needs the ROCm LLVM binutils')
def test_no_promoted_arrays_and_no_scratch_in_the_production_kernels(tmp_path):
    ks = _kernels(str(tmp_path))
    # the census (round 6): 3 tasks + 2 circle baselines, 2 dtypes, 2 charts, ... on the mappings
    # csrc/atacom_ops_impl.h: has_mapping instantiates -- it was 678 kernels / 46 MB with every (dtype, task, mapping) product
    assert 400 < len(ks) < 450, len(ks)
    from rl_on_manifold_amd import build
    assert os.path.getsize(build.LIB) < 32 * 2 ** 20
    names = {k[0] for k in ks}
    for gone in ('k_step<double, Circle, 8, true, false, 0, false>', 'k_step<float, Circle, 4, true, false, 0, false>',
                 'k_step<double, Iiwa, 2, true, false, 0, false>', 'k_step<double, Planar, 8, true, false, 0, false>',
                 'k_rollout_mlp<double, Iiwa, 2, true, 64, false, 0, false>', 'k_rollout_mlp<double, Iiwa, 4, true, 64, false, 1, false>'):
        assert gone not in names, gone
    assert any(k[0].startswith('k_step<float, Iiwa, 4, true, false, 0, false>') for k in ks), [k[0] for k in ks][:5]
    # static LDS: the two-stage statistics reduction, and the float32 rigid-body kernels of the reference chart, which park
    # the held solver state across the dynamics (84 values per lane: 21 float4 x threads per workgroup; atacom_kernels.h) --
    # nothing else (the policy kernels' LDS is dynamic)
    def args(name):
        return [a.strip() for a in name[name.index('<') + 1:name.rindex('>')].split(',')]

    def parked(name):
        if not name.startswith(('k_step<float, Iiwa', 'k_rollout<float, Iiwa', 'k_rollout_mlp<float, Iiwa')):
            return False
        a = args(name)
        dyn, chart = (a[4], a[5]) if not name.startswith('k_rollout_mlp<') else (a[5], a[6])
        return dyn == 'true' and chart == '0' and a[2] != '1'    # (one environment per lane: measured slower with it, not parked)
    lds = [k for k in ks if k[1] != 0]
    assert lds and all(k[0].startswith('k_stats<') or parked(k[0]) for k in lds), [k for k in lds if not parked(k[0])][:5]
    for k in lds:
        if parked(k[0]):
            assert k[1] == 21 * 16 * 256, k                  # 84 values per lane, 256 threads per workgroup
    # float32, kinematic mode (DYN = false), every task, mapping and chart: the single-step kernels never touch scratch,
    # and neither do the T-step kernels of the lane-group mappings
    bad, n_noise = [], 0
    for name, _, scratch, *_ in ks:
        if not name.startswith(('k_step<float', 'k_rollout<float')):
            continue
        a = args(name)                                       # T, E, LANES, HOLD, DYN, CHART, NOISE
        if a[4] == 'true':
            # rigid-body mode (opt-in).  Quad mapping, reference chart, held q (the mode's default configuration):
            # NO scratch since the solver state is parked in LDS across the dynamics (round 4; it had been 32 - 208 bytes per
            # lane); the other instantiations (one environment per lane, refreshed q, canonical chart) are bounded
            if scratch > (0 if (a[2] == '4' and a[3] == 'true' and a[5] == '0') else 700):
                bad.append((name, scratch))
            continue
        noise = a[6] == 'true'                               # the kernels with the domain-randomisation options compiled in
        n_noise += noise
        if noise and a[3] == 'false' and name.startswith('k_rollout<'):
            continue                                         # ... refreshed q (hold_q = 0) in a T-step kernel: two of them spill
        if noise and a[5] == '1' and a[1] == 'Planar' and scratch <= 32:
            continue                                         # ... the planar canonical-chart step kernel (second form): 20 bytes
        if name.startswith('k_step<') or int(a[2]) > 1:
            if scratch:
                bad.append((name, scratch))
    assert not bad, bad
    assert n_noise == 2 * 2 * 4 * 2 * 2                      # planar + iiwa, step + rollout, 4 mappings, HOLD, 2 charts (float32)


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, 'llvm-readelf')), reason='needs the ROCm LLVM binutils')
def test_float64_kernels_hold_their_arrays_in_registers(tmp_path):
    """rounds 1 - 5 compiled the lane-group solver of the float64 kernels as real functions; arrays handed
    over by reference lived in scratch (1.6 - 2.6 KB per lane, 183 of 331 double kernels, 112 us per step on the headline
    workload).  With the solver inlined (csrc/atacom_quad.h) what is left is genuine register spilling in the iiwa kernels
    (256 double-width values do not fit where 256 floats do); pinned here:
      * the float64 step kernel of the mapping the policy picks for the headline workload -- 8 lanes, reference chart, held
        q -- uses NO scratch;
      * no float64 kernel outside the iiwa task uses scratch, except the planar one-lane policy kernel;
      * the lane-group iiwa step / T-step kernels in kinematic mode stay below 800 bytes per lane (spills), i.e. no private
        array has come back (the smallest array of the solver is 12 doubles x 2 slots = 192 bytes ON TOP of the spills --
        the outlined build sat at 1584 - 2392)."""
    ks = _kernels(str(tmp_path))

    def args(name):
        return [a.strip() for a in name[name.index('<') + 1:name.rindex('>')].split(',')]
    by = {k[0]: k for k in ks}
    head = by['k_step<double, Iiwa, 8, true, false, 0, false>']
    assert head[2] == 0 and head[1] == 0, head
    bad = []
    for name, lds, scratch, *_ in ks:
        if 'double' not in name or not scratch:
            continue
        a = args(name)
        if len(a) < 2 or a[1] != 'Iiwa':
            if not (name.startswith('k_rollout_mlp<double, Planar, 1,') or (a[1:3] == ['Planar', '1'] and scratch <= 32)):
                bad.append((name, scratch))
            continue
        if name.startswith(('k_step<', 'k_rollout<')) and int(a[2]) > 1 and a[4] == 'false' and scratch > 800:
            bad.append((name, scratch))
    assert not bad, bad


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, 'llvm-objdump')), reason='needs the ROCm LLVM binutils')
def test_no_register_copies_under_a_narrowed_exec_mask(tmp_path):
    """ADVICE r5 (medium), isolated in round 6 (profiles/r06_exec_mask_copies.md): hipcc 7.2 can place the register allocator's
    live-range copies (v_accvgpr_write ...) at the top of the JOIN block of a lane-0-only store region, in front of the
    `s_or_b64 exec, exec, ...` that reopens the mask -- only the lanes that ran the region get their copy, the rest read
    stale registers afterwards.  That is what fed the in-kernel network of the rigid-body policy kernel a wrong bias in three
    of four lanes when built without the LDS parking (the flagged kernel is exactly the one that fails on hardware,
    profiles/r06_nopark_policy_tests.log).  The defect is silent and moves with code generation, so EVERY kernel of the built
    library is disassembled and audited for the pattern (profiles/tools/exec_restore_audit.py): a join label followed by
    nothing but vector copies up to the exec restore."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'profiles', 'tools'))
    from exec_restore_audit import audit_library
    from rl_on_manifold_amd import build
    found, n_objects = audit_library(build.build(verbose=False), str(tmp_path), LLVM)
    assert n_objects >= 13
    assert not found, [(f[1], [t for _, t in f[4]]) for f in found]
