#!/usr/bin/env python3
"""isa_lint.py -- wait-state hazards the compiler cannot see inside inline asm (gfx9 / CDNA, gfx950).

hipcc's hazard recogniser pads the instructions it emits itself; the text of an `asm volatile`
is opaque to it.  This script reads the device assembly of every object of the library
(`hipcc ... --offload-device-only -S`) and applies the manual-wait-state rules of the ISA to ALL
vector-memory instructions, the ones from inline asm included:

  R1  VALU writes an SGPR (v_readlane / v_readfirstlane / v_cmp / carry-out ...)  ->  a VMEM
      instruction reads that SGPR: 5 wait states.  (A spilled store base restored by v_readlane
      directly in front of an asm `global_store ... s[n:n+1]` is a store through a half-written
      64-bit address: a wild write.)
  R2  a VMEM store of more than 64 bits of data  ->  a VALU write of its data VGPRs: 2 wait states
      on gfx940+ (the store still reads them).
  R3  SALU writes M0  ->  an LDS-DMA VMEM instruction (reads M0): 1 wait state.
      A VALU write of M0 (v_readfirstlane m0) falls under R1.

An instruction is one wait state, `s_nop k` is k + 1.  The rule for R1 is the conservative one LLVM
uses for its own code: ANY VALU write of the register within the window counts, whatever was
written to it in between.  The search follows the control flow (labels + s_branch / s_cbranch):
the state at a block's head is the worst case over its predecessors.

Compiler-emitted VMEM must pass too (it does: that is the lint's self-check); findings say whether
the instruction sits inside an asm block.

  isa_lint.py file.s [file.s ...]      exit status 1 when anything is found
"""
import re
import sys
from collections import defaultdict

R1_WAIT = 5
R2_WAIT = 2
R3_WAIT = 1
CAP = 8          # ages saturate here (older than every rule's window)

_label = re.compile(r'^([.\w$@]+):')
_reg_range = re.compile(r'^([sv])\[(\d+):(\d+)\]$')
_reg_one = re.compile(r'^([sv])(\d+)$')

VMEM_PREFIX = ('global_', 'buffer_', 'flat_', 'scratch_', 'tbuffer_', 'image_')
# VALU instructions whose SECOND operand is an SGPR result (carry / scale flag) as well
_second_dst = re.compile(r'^v_(add_co|sub_co|subrev_co|addc_co|subb_co|subbrev_co|div_scale|mad_u64_u32|mad_i64_i32)')


def parse_regs(op):
    """operand text -> list of ('s'|'v', index) or special names ('vcc_lo', 'vcc_hi', 'm0')."""
    op = op.strip()
    op = re.sub(r'^[-|]+|[|]+$', '', op)           # neg / abs decorations
    m = _reg_range.match(op)
    if m:
        return [(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)]
    m = _reg_one.match(op)
    if m:
        return [(m.group(1), int(m.group(2)))]
    if op == 'vcc':
        return [('x', 'vcc_lo'), ('x', 'vcc_hi')]
    if op in ('vcc_lo', 'vcc_hi', 'm0'):
        return [('x', op)]
    return []


def split_operands(rest):
    out, depth, cur = [], 0, ''
    for ch in rest:
        if ch == '[':
            depth += 1
        elif ch == ']':
            depth -= 1
        if ch == ',' and depth == 0:
            out.append(cur)
            cur = ''
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    # trailing modifiers ("off offset:16 nt", "s[0:1] offset:8") ride on the last operands
    res = []
    for o in out:
        toks = o.strip().split()
        res.append(toks[0] if toks else '')
        for t in toks[1:]:
            res.append(t)
    return res


class Ins:
    __slots__ = ('line', 'text', 'mn', 'ops', 'in_asm')

    def __init__(self, line, text, in_asm):
        self.line, self.text, self.in_asm = line, text, in_asm
        parts = text.split(None, 1)
        self.mn = parts[0]
        self.ops = split_operands(parts[1]) if len(parts) > 1 else []

    def wait_states(self):
        if self.mn == 's_nop':
            try:
                return int(self.ops[0], 0) + 1
            except (ValueError, IndexError):
                return 1
        return 1

    def is_valu(self):
        return self.mn.startswith('v_')

    def is_vmem(self):
        return self.mn.startswith(VMEM_PREFIX)

    def is_lds_dma(self):
        return self.is_vmem() and ('_lds_' in self.mn or 'lds' in self.ops)

    def valu_sgpr_defs(self):
        """SGPRs (and vcc / m0) a VALU instruction writes."""
        if not self.is_valu() or not self.ops:
            return []
        regs = [r for r in parse_regs(self.ops[0]) if r[0] in 'sx']
        if _second_dst.match(self.mn) and len(self.ops) > 1:
            regs += [r for r in parse_regs(self.ops[1]) if r[0] in 'sx']
        # the e32 compares write vcc without naming it
        if self.mn.startswith('v_cmp') and not regs:
            regs = [('x', 'vcc_lo'), ('x', 'vcc_hi')]
        return regs

    def valu_vgpr_defs(self):
        if not self.is_valu() or not self.ops:
            return []
        return [r for r in parse_regs(self.ops[0]) if r[0] == 'v']

    def scalar_reads(self):
        regs = []
        for o in self.ops:
            regs += [r for r in parse_regs(o) if r[0] in 'sx']
        return regs

    def wide_store_data(self):
        """data VGPRs of a store of more than 64 bits."""
        if not self.is_vmem() or 'store' not in self.mn:
            return []
        if not re.search(r'(dwordx3|dwordx4|b96|b128)', self.mn):
            return []
        # global/flat/scratch: addr, data, saddr ; buffer: data, vaddr, rsrc, soffset
        idx = 0 if self.mn.startswith(('buffer_', 'tbuffer_')) else 1
        if len(self.ops) <= idx:
            return []
        return [r for r in parse_regs(self.ops[idx]) if r[0] == 'v']

    def salu_m0_def(self):
        return self.mn.startswith('s_') and self.ops and self.ops[0] == 'm0' and not self.mn.startswith('s_cmp')


def read_blocks(path):
    """-> list of functions: (name, [block]) with block = dict(label, ins[], succ labels, fall)"""
    funcs = []
    cur_blocks = None
    blk = None
    in_asm = False
    fname = None
    with open(path) as f:
        for ln, raw in enumerate(f, 1):
            s = raw.rstrip('\n')
            st = s.strip()
            if st.startswith(';;#ASMSTART'):
                in_asm = True
                continue
            if st.startswith(';;#ASMEND'):
                in_asm = False
                continue
            if not st or st.startswith(';') or st.startswith('//'):
                continue
            m = _label.match(s)
            if m and not s[0].isspace():
                lab = m.group(1)
                if not lab.startswith('.L') and not lab.startswith('.'):
                    fname = lab                          # a function (or object) starts
                    cur_blocks = []
                    funcs.append((fname, cur_blocks))
                if cur_blocks is None:
                    continue
                blk = {'label': lab, 'ins': [], 'succ': [], 'fall': True}
                cur_blocks.append(blk)
                continue
            if st.startswith('.'):
                continue
            if blk is None:
                continue
            text = st.split(';')[0].strip()
            if not text:
                continue
            for piece in text.split('\n'):
                ins = Ins(ln, piece.strip(), in_asm)
                blk['ins'].append(ins)
                mn = ins.mn
                term = False
                if mn == 's_branch':
                    blk['succ'].append(ins.ops[0]); blk['fall'] = False; term = True
                elif mn.startswith('s_cbranch'):
                    blk['succ'].append(ins.ops[-1]); term = True
                elif mn in ('s_endpgm', 's_setpc_b64', 's_trap'):
                    blk['fall'] = False; term = True
                if term:
                    nb = {'label': None, 'ins': [], 'succ': [], 'fall': True}
                    cur_blocks.append(nb)
                    blk = nb
    return funcs


def run_block(blk, state, findings, fname, path):
    st = dict(state)
    for ins in blk['ins']:
        where = 'inline asm' if ins.in_asm else 'compiler-emitted'
        if ins.is_vmem():
            for r in ins.scalar_reads():
                age = st.get(('S', r))
                if age is not None and age < R1_WAIT:
                    findings.add((path, fname, ins.line, 'R1',
                                  f'{ins.text}   <- reads {fmt(r)} {age} wait state(s) after a VALU wrote it '
                                  f'(needs {R1_WAIT}) [{where}]'))
            if ins.is_lds_dma():
                age = st.get(('M', 'm0'))
                if age is not None and age < R3_WAIT:
                    findings.add((path, fname, ins.line, 'R3',
                                  f'{ins.text}   <- LDS-DMA {age} wait state(s) after SALU wrote m0 '
                                  f'(needs {R3_WAIT}) [{where}]'))
                age = st.get(('S', ('x', 'm0')))
                if age is not None and age < R1_WAIT:
                    findings.add((path, fname, ins.line, 'R1',
                                  f'{ins.text}   <- LDS-DMA {age} wait state(s) after VALU wrote m0 [{where}]'))
        if ins.is_valu():
            for r in ins.valu_vgpr_defs():
                age = st.get(('W', r))
                if age is not None and age < R2_WAIT:
                    findings.add((path, fname, ins.line, 'R2',
                                  f'{ins.text}   <- overwrites {fmt(r)} {age} wait state(s) after a >64-bit store '
                                  f'that reads it (needs {R2_WAIT}) [store in {st.get(("Wsrc", r), "?")}]'))
        ws = ins.wait_states()
        for k in list(st):
            if k[0] == 'Wsrc':
                continue
            st[k] = min(CAP, st[k] + ws)
            if st[k] >= CAP:
                del st[k]
                if k[0] == 'W':
                    st.pop(('Wsrc', k[1]), None)
        for r in ins.valu_sgpr_defs():
            st[('S', r)] = 0
        wide = ins.wide_store_data()
        for r in wide:
            st[('W', r)] = 0
            st[('Wsrc', r)] = 'inline asm' if ins.in_asm else 'compiler-emitted'
        if ins.salu_m0_def():
            st[('M', 'm0')] = 0
    return st


def fmt(r):
    return f'{r[0]}{r[1]}' if r[0] in 'sv' else str(r[1])


def lint_file(path):
    findings = set()
    for fname, blocks in read_blocks(path):
        by_label = {b['label']: i for i, b in enumerate(blocks) if b['label']}
        n = len(blocks)
        in_state = [None] * n
        if n == 0:
            continue
        in_state[0] = {}
        work = [0]
        guard = 0
        while work:
            guard += 1
            if guard > 200000:
                raise RuntimeError(f'{path}:{fname}: dataflow does not settle')
            i = work.pop()
            scratch = set()
            out = run_block(blocks[i], in_state[i], scratch, fname, path)
            out = {k: v for k, v in out.items() if k[0] != 'Wsrc' or ('W', k[1]) in out}
            succ = []
            if blocks[i]['fall'] and i + 1 < n:
                succ.append(i + 1)
            for lab in blocks[i]['succ']:
                if lab in by_label:
                    succ.append(by_label[lab])
            for j in succ:
                if in_state[j] is None:
                    in_state[j] = dict(out)
                    work.append(j)
                else:
                    merged, changed = merge_states(in_state[j], out)
                    if changed:
                        in_state[j] = merged
                        work.append(j)
        for i, b in enumerate(blocks):
            if in_state[i] is not None:
                run_block(b, in_state[i], findings, fname, path)
    return sorted(findings, key=lambda f: (f[0], f[2]))


def merge_states(a, b):
    out = dict(a)
    changed = False
    for k, v in b.items():
        if k[0] == 'Wsrc':
            out.setdefault(k, v)
            continue
        if k not in out or v < out[k]:
            out[k] = v
            changed = True
    return out, changed


def main(argv):
    if len(argv) < 2:
        print(__doc__)
        return 2
    total = 0
    per_rule = defaultdict(int)
    for p in argv[1:]:
        fs = lint_file(p)
        for path, fname, line, rule, msg in fs:
            print(f'{path}:{line}: {rule} in {fname}:\n    {msg}')
            per_rule[rule] += 1
        total += len(fs)
    if total:
        print(f'isa_lint: {total} hazard(s): ' + ', '.join(f'{k}={v}' for k, v in sorted(per_rule.items())))
        return 1
    print(f'isa_lint: {len(argv) - 1} file(s) clean')
    return 0


if __name__ == '__main__':
    sys.exit(main(sys.argv))
