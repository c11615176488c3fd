"""Audit of a gfx950 kernel's assembly for VECTOR REGISTERS DEFINED UNDER A PARTIAL EXEC MASK AND READ AFTER THE MASK IS RESTORED.

    python profiles/tools/exec_region_audit.py kernel.s [kernel-name-substring]

Why (profiles/r05_dyn_mlp_park.md): the rigid-body policy kernel built without the LDS parking fed its
in-kernel network a wrong bias in the lanes with lane % 4 != 0 -- the signature of a register copy / reload executed inside a
lane-0-only region (`if (lq == 0 && valid) { stores }`) whose value is then consumed by all lanes.  In the SOURCE those regions
contain stores only, so any vector register written inside one and read outside is the compiler's doing.  The audit walks
the straight-line text of one kernel, tracks `s_and_saveexec_b64 ... / s_or_b64 exec, exec, ...` nesting (the structured
control flow hipcc emits), records v / a registers written while the mask is narrowed and reports those that are read after
the region closes before being rewritten.  (Linear scan: a first-order check, not a data-flow proof; loops are followed once.)
"""
import re
import sys

REG = re.compile(r'\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]')
NO_DEST = ('global_store', 'buffer_store', 'ds_write', 'scratch_store', 'flat_store', 's_', 'v_cmp', 'v_cmpx', 'v_readlane',
           'v_readfirstlane', 'v_nop', 'buffer_wbl2', 'buffer_inv', 'ds_gws', 'global_atomic', 'ds_add', 'v_writelane')


def regs(tok):
    out = []
    for m in REG.finditer(tok):
        if m.group(1):
            out.append((m.group(1), int(m.group(2))))
        else:
            out += [(m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1)]
    return out


def split_ops(line):
    body = line.split(';')[0].strip()
    if not body or body.endswith(':') or body.startswith('.'):
        return None, []
    parts = body.split(None, 1)
    op = parts[0]
    ops = []
    if len(parts) > 1:
        depth, cur = 0, ''
        for ch in parts[1]:
            if ch == '[':
                depth += 1
            elif ch == ']':
                depth -= 1
            if ch == ',' and depth == 0:
                ops.append(cur.strip()); cur = ''
            else:
                cur += ch
        if cur.strip():
            ops.append(cur.strip())
    return op, ops


MAX_REGION = 400     # instructions: the lane-0 store blocks are tens of instructions; whole-kernel guards (b < B) are not audited


def audit(lines):
    stack = []          # open regions: [saved-exec sgpr operand, open line, {reg: (line, text)}]
    pending = {}        # reg -> (def line, text, region open line): defs made inside a SMALL closed region, not yet rewritten
    findings = []
    regions = 0
    for n, raw in enumerate(lines, 1):
        op, ops = split_ops(raw)
        if op is None:
            continue
        if op in ('s_and_saveexec_b64', 's_or_saveexec_b64'):
            stack.append([ops[0], n, {}])
            continue
        if op in ('s_or_b64', 's_mov_b64', 's_xor_b64') and ops and ops[0] == 'exec':
            saved = ops[-1]
            for k in range(len(stack) - 1, -1, -1):
                if stack[k][0] == saved:
                    reg_open, defs = stack[k][1], stack[k][2]
                    for inner in stack[k + 1:]:
                        defs.update(inner[2])
                    del stack[k:]
                    if n - reg_open <= MAX_REGION:
                        regions += 1
                        if stack:
                            stack[-1][2].update(defs)            # still narrowed by an outer SMALL-or-large region: its problem
                        for r, (dl, dt) in defs.items():
                            pending[r] = (dl, dt, reg_open)
                    elif stack:
                        stack[-1][2].update(defs)
                    break
            continue
        dest, srcs = [], []
        if op.startswith(NO_DEST):
            for o in ops:
                srcs += regs(o)
        else:
            if ops:
                dest = regs(ops[0])
            for o in ops[1:]:
                srcs += regs(o)
            if op.startswith(('v_fmac', 'v_mac', 'v_pk_fmac', 'v_dot')) or ('dpp' in op and 'bound_ctrl' not in raw):
                srcs += dest
        for r in srcs:
            if r in pending:
                findings.append((pending[r], (n, raw.strip()), r))
                del pending[r]
        for r in dest:
            pending.pop(r, None)
            if stack:
                stack[-1][2][r] = (n, raw.strip())
    return findings, regions


def main():
    text = open(sys.argv[1]).read().split('\n')
    want = sys.argv[2] if len(sys.argv) > 2 else None
    # cut kernels: from a label ending ':' that starts with _Z to s_endpgm
    i = 0
    total = 0
    while i < len(text):
        if text[i].startswith('_Z') and text[i].rstrip().endswith(':') or (text[i].startswith('_Z') and ':' in text[i][:400]):
            name = text[i].split(':')[0]
            j = i
            while j < len(text) and 's_endpgm' not in text[j]:
                j += 1
            if want is None or want in name:
                f, nreg = audit(text[i:j + 1])
                print('%s: %d lines, %d small exec-narrowed regions, %d vector registers written inside one and read after it closes'
                      % (name[:90], j - i, nreg, len(f)))
                for (dl, dtxt, ropen), (ul, utxt), r in f[:40]:
                    print('   %s%d  def @%d [region opened @%d]: %s\n          use @%d: %s' % (r[0], r[1], dl + i, ropen + i, dtxt[:90], ul + i, utxt[:90]))
                total += len(f)
            i = j
        i += 1
    return total


if __name__ == '__main__':
    main()
