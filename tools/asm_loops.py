#!/usr/bin/env python3
"""Instruction mix of every loop of one kernel in a `make asm` listing (VALU / LDS / memory /
SALU / waits / barriers between a back edge and its target).  Usage: asm_loops.py build/k_eq.s SYMBOL"""
import re
import sys

src = open(sys.argv[1]).read().split('\n')
name = sys.argv[2]
start = [i for i, l in enumerate(src) if l.startswith(name + ':')][0]
end = [i for i, l in enumerate(src) if i > start and l.strip().startswith('.Lfunc_end')][0]
blocks, cur = [], None
for l in src[start:end]:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        cur = [m.group(1), []]
        blocks.append(cur)
        continue
    s = l.strip()
    if not s or s.startswith(';') or s.startswith('.'):
        continue
    if cur is None:
        cur = ['entry', []]
        blocks.append(cur)
    cur[1].append(s)
idx = {b[0]: i for i, b in enumerate(blocks)}


def cls(s):
    op = s.split()[0]
    for p, c in (('v_', 'V'), ('ds_', 'L'), ('global_', 'M'), ('buffer_', 'M'), ('flat_', 'M'),
                 ('s_barrier', 'B'), ('s_waitcnt', 'W'), ('s_', 'S')):
        if op.startswith(p):
            return c
    return '?'


for i, b in enumerate(blocks):
    for s in b[1]:
        m = re.match(r's_c?branch\S*\s+(\.LBB\d+_\d+)', s)
        if m and m.group(1) in idx and idx[m.group(1)] <= i:
            j = idx[m.group(1)]
            cnt = {}
            for bb in blocks[j:i + 1]:
                for t in bb[1]:
                    cnt[cls(t)] = cnt.get(cls(t), 0) + 1
            print(f"loop {blocks[j][0]}..{b[0]} blocks={i - j + 1}", dict(sorted(cnt.items())))
