#!/usr/bin/env python3
"""Stress of the dB-finish helper pool: many collects of 4096 windows, every result compared."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
cm = ge.load_package()
S, C, T = 4096, 2, 2048
b = cm.Batch(S, C, T, flags=cm.VU)
b.set_gain(-1, 2, 1000, [750, 1250])
b.generate(cm.GEN_NOISE, 7, T)
b.run(T)
ref, rcs = b.vu_results()
ref = [r.as_dict() for r in ref]
bad = 0
for it in range(300):
    b.run(T)
    b.vu_snapshot()
    b.run(T)                       # a second window in flight while the first is finished
    b.vu_snapshot()
    r1, c1 = b.vu_collect()
    r2, c2 = b.vu_collect()
    for r in (r1, r2):
        if any(r[s].as_dict() != ref[s] for s in range(0, S, 1)):
            bad += 1
print("pool stress:", "ok" if not bad else f"{bad} bad collects")
sys.exit(1 if bad else 0)
