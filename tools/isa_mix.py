#!/usr/bin/env python3
"""Instruction mix per kernel of a hipcc -S dump: python tools/isa_mix.py file.s [name-filter]"""
import collections, re, sys
txt = open(sys.argv[1]).read().split('\n')
flt = sys.argv[2] if len(sys.argv) > 2 else ''
cur = None; ops = None
def flush():
    if cur and flt in cur:
        tot = sum(ops.values())
        top = ', '.join(f'{k}:{v}' for k, v in ops.most_common(22))
        print(f'{cur[:60]} total={tot}\n   {top}')
for line in txt:
    m = re.match(r'^(_Z\w+):', line)
    if m:
        flush(); cur = m.group(1); ops = collections.Counter(); continue
    if cur is None: continue
    if line.startswith('.Lfunc_end'):
        flush(); cur = None; continue
    m = re.match(r'^\s+([a-z]+_[a-z0-9_]+)', line)
    if m: ops[m.group(1)] += 1
