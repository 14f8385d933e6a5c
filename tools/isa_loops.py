#!/usr/bin/env python3
"""Loops of one kernel in a hipcc -S dump, with their instruction mix:
   python tools/isa_loops.py file.s kernel-name-substring [min-instructions]
A loop is a backward branch to a label; nested loops are listed separately (outer ones include inner ones).
Half-rate VALU opcodes are the ones profiles/r02_valu_ops_microbench.txt measured at ~4.4 cycles."""
import collections
import re
import sys

FULL = re.compile(r'^v_(add|sub|subrev)_(u32|f32|u16|i16|f16)|^v_(and|or|xor|not)_b32|^v_(lshrrev_b32|ashrrev_i32|mov_b32)'
                  r'|^v_(fma|fmac|mul)_f32|^v_(max|min)_(u16|i16|f16)|^v_cvt_f32_ubyte')


def classify(op):
    if op.startswith('v_'):
        if op.endswith('_dpp') or op.endswith('_sdwa'):
            return 'valu_half'
        return 'valu_full' if FULL.match(op) else 'valu_half'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_')):
        return 'vmem'
    if op.startswith('s_waitcnt'):
        return 'wait'
    if op.startswith('s_'):
        return 'salu'
    return 'other'


def main():
    txt = open(sys.argv[1]).read().split('\n')
    flt = sys.argv[2]
    min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    start = None
    for i, line in enumerate(txt):
        m = re.match(r'^(_Z\w+):', line)
        if m and flt in m.group(1):
            start = i
            name = m.group(1)
            break
    if start is None:
        sys.exit('kernel not found')
    instrs = []  # (op, text)
    labels = {}
    for line in txt[start + 1:]:
        if line.startswith('.Lfunc_end'):
            break
        m = re.match(r'^(\.LBB\w+):', line)
        if m:
            labels[m.group(1)] = len(instrs)
            continue
        m = re.match(r'^\s+([a-z]+_[a-z0-9_]+)\s*(.*)', line)
        if m:
            instrs.append((m.group(1), m.group(2)))
    print(name, 'instructions:', len(instrs))
    for idx, (op, rest) in enumerate(instrs):
        if op.startswith('s_cbranch') or op == 's_branch':
            tgt = rest.split()[0] if rest else ''
            if tgt in labels and labels[tgt] <= idx:
                body = instrs[labels[tgt]:idx + 1]
                if len(body) < min_n:
                    continue
                cat = collections.Counter(classify(o) for o, _ in body)
                ops = collections.Counter(o for o, _ in body)
                print(f'loop {tgt} .. +{idx}: {len(body)} instr  ' + '  '.join(f'{k}={v}' for k, v in sorted(cat.items())))
                print('   ' + ', '.join(f'{k}:{v}' for k, v in ops.most_common(28)))


main()
