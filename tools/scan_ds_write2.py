#!/usr/bin/env python3
"""Scans hipcc -S output for a multi-dword LDS store whose data VGPRs are overwritten by one of the next few
instructions (observed on gfx950: ds_write2_b32 followed at once by a VALU write of its second data register stored
the NEW value; hipcc pads only stores of more than 64 bits of data).  python tools/scan_ds_write2.py file.s [window]"""
import re, sys
lines = [l.rstrip() for l in open(sys.argv[1])]
win = int(sys.argv[2]) if len(sys.argv) > 2 else 2
def regs(tok):
    tok = tok.strip()
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()
ins = [(i, l.strip()) for i, l in enumerate(lines) if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
hits = 0
for idx, (i, l) in enumerate(ins):
    m = re.match(r'(ds_write2\w*|ds_write_b64|ds_write_b96|ds_write_b128)\s+(.*)', l)
    if not m: continue
    ops = [o for o in m.group(2).split(',')]
    data = set()
    for o in ops[1:]:
        data |= regs(o.split(' offset')[0])
    for (i2, l2) in ins[idx + 1: idx + 1 + win]:
        m2 = re.match(r'(v_\w+|ds_read\w*|global_load\w*)\s+([^,]+),', l2)
        if m2 and regs(m2.group(2)) & data:
            hits += 1
            print(f'{sys.argv[1]}:{i+1}: {l}\n      -> {i2+1}: {l2}')
            break
print('hits:', hits)
