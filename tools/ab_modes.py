#!/usr/bin/env python3
"""Read-only (VU window only) kernel time by gain form: general ({750,1250}/1000 or 1001/1000), every
gain below the scale (900/1000), no gain.  usage: ab_modes.py [c2|c4]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

cm = ge.load_package()
shape = sys.argv[1] if len(sys.argv) > 1 else "c2"
S, C = (4096, 2) if shape == "c2" else (8192, 1)
T = 65536
forms = {"general": [750, 1250][:C] if C == 2 else [1001], "below the scale": [900, 800][:C], "no gain": None}
bs = {}
for name, g in forms.items():
    b = cm.Batch(S, C, T, flags=cm.VU)
    if g is not None:
        b.set_gain(-1, C, 1000, g)
    b.generate(cm.GEN_NOISE, 12345, T)
    bs[name] = (b, [])
for rnd in range(7):
    for name, (b, v) in bs.items():
        b.run(T)
        b.sync()
        b.timing(True)
        b.timing_read()
        for _ in range(20):
            b.run(T)
        ms, n = b.timing_read()
        b.timing(False)
        b.vu_reset(-1)
        v.append(ms / n)
for name, (b, v) in bs.items():
    m = statistics.median(v)
    print(f"{shape} VU only, {name:16s}: median {m:.4f} ms  min {min(v):.4f}  -> {S*C*T*2/(m*1e-3)/1e9:.0f} GB/s")
