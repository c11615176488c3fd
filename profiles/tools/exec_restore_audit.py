"""Flag vector-register writes that sit directly in front of an exec-mask RESTORE (`s_or_b64 exec, exec, s[..]`) at a join label.

    python profiles/tools/exec_restore_audit.py kernel.s [more.s ...]      (hipcc -S output or llvm-objdump -d --symbolize-operands)
    python profiles/tools/exec_restore_audit.py --so rl_on_manifold_amd/libatacom_hip.so      (every kernel of a built library)

The defect of hipcc 7.2 behind ADVICE r5 (medium) / profiles/r06_exec_mask_copies.md: in a kernel at the register ceiling the
register allocator's live-range copies (v_accvgpr_write / v_accvgpr_read / v_mov) for values that are live THROUGH a
lane-0-only store block were placed at the top of the block's join label, i.e. BEFORE the `s_or_b64 exec, exec, ...` that
reopens the mask -- so only the lanes that ran the store block got their copy; the others read stale registers after the join.
A vector write between a label and the exec restore that follows it executes under the narrowed mask of the region being
closed; wave-wide values must not be copied there.  Flagged: a join label followed by NOTHING BUT register copies (and scalar
bookkeeping) up to the `s_or_b64 exec, exec, ...` -- the body of a region (real work behind a label) is not.  Exit status 1 if any is found."""
import re
import sys

LABEL = re.compile(r'^(\.LBB\d+_\d+:|[0-9a-f]+ <L\d+>:)')            # hipcc -S text / llvm-objdump --symbolize-operands
KERNEL = re.compile(r'^(_Z\w+):|^[0-9a-f]+ <(_Z\w+)>:')
COPY = re.compile(r'(v_accvgpr_write|v_accvgpr_read|v_mov_b32|v_mov_b64|v_accvgpr_mov)')
RESTORE = re.compile(r'\s*s_or_b64 exec, exec, (s\[\d+:\d+\])')
ENDS_BLOCK = ('s_cbranch', 's_branch', 's_endpgm', 's_setpc', 's_swappc')


def audit(lines):
    """[(kernel, label line, restore line, [(line, text), ...])] for every exec restore with vector copies between it and the
    label that opens its block."""
    out = []
    kernel = None
    for i, ln in enumerate(lines):
        k = KERNEL.match(ln)
        if k:
            kernel = k.group(1) or k.group(2)
        if not RESTORE.match(ln):
            continue
        j = i - 1
        writes = []
        while j >= 0 and not LABEL.match(lines[j]) and not KERNEL.match(lines[j]):
            t = lines[j].strip()
            if t.startswith(ENDS_BLOCK):
                writes = None
                break
            if COPY.match(t) and '_dpp' not in t:
                writes.append((j + 1, t.split('//')[0].strip()))
            elif t and not t.startswith((';', '//', 's_', 'v_readlane', 'v_nop')) and not t.endswith(':'):
                writes = None                 # real work between the label and the restore: the body of a region, not a join
                break
            j -= 1
        if writes and j >= 0 and LABEL.match(lines[j]):
            out.append((kernel, j + 1, i + 1, list(reversed(writes))))
    return out


def audit_library(so, tmp, llvm='/opt/rocm/lib/llvm/bin'):
    """Every gfx950 code object inside a built libatacom_hip.so, disassembled with labels: [(unit, kernel, label, restore, copies)]."""
    import os
    import struct
    import subprocess
    fat = os.path.join(tmp, 'fat.bin')
    subprocess.check_call([os.path.join(llvm, 'llvm-objcopy'), '--dump-section', '.hip_fatbin=' + fat, so, os.path.join(tmp, 'copy.so')])
    data = open(fat, 'rb').read()
    out, n = [], 0
    for m in re.finditer(b'__CLANG_OFFLOAD_BUNDLE__', data):
        p0 = m.start()
        n_entries = struct.unpack_from('<Q', data, p0 + 24)[0]
        off = p0 + 32
        for _ in range(n_entries):
            o, size, id_len = struct.unpack_from('<QQQ', data, off)
            ident = data[off + 24:off + 24 + id_len].decode()
            off += 24 + id_len
            if 'gfx950' not in ident or size == 0:
                continue
            elf = os.path.join(tmp, 'dev%d.elf' % n)
            open(elf, 'wb').write(data[p0 + o:p0 + o + size])
            text = subprocess.run([os.path.join(llvm, 'llvm-objdump'), '-d', '--symbolize-operands', '--no-show-raw-insn', elf],
                                  capture_output=True, text=True, check=True).stdout
            out += [(n,) + f for f in audit(text.split('\n'))]
            n += 1
    return out, n


def main():
    found = 0
    if len(sys.argv) > 2 and sys.argv[1] == '--so':
        import tempfile
        with tempfile.TemporaryDirectory() as tmp:
            res, n = audit_library(sys.argv[2], tmp)
        for unit, kernel, lab, rest, writes in res:
            found += len(writes)
            print('code object %d: %s: %d vector copies between label @%d and the exec restore @%d: %s'
                  % (unit, (kernel or '?')[:90], len(writes), lab, rest, '; '.join(t for _, t in writes[:9])))
        print('%d code objects, %d copies under a narrowed mask in front of an exec restore' % (n, found))
        return 1 if found else 0
    for path in sys.argv[1:]:
        for kernel, lab, res, writes in audit(open(path).read().split('\n')):
            found += len(writes)
            print('%s: %s: %d vector copies between label @%d and the exec restore @%d'
                  % (path, (kernel or '?')[:70], len(writes), lab, res))
            for n, t in writes[:12]:
                print('    @%d  %s' % (n, t[:80]))
    print('%d copies under a narrowed mask in front of an exec restore' % found)
    return 1 if found else 0


if __name__ == '__main__':
    sys.exit(main())
