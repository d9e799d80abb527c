#!/usr/bin/env python3
"""config-2 kernel time with the PCM written to a second array and written in place (as the
reference's transform works), interleaved rounds in one process.  usage: ab_inplace.py [c2|c4]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

cm = ge.load_package()
shape = sys.argv[1] if len(sys.argv) > 1 else "c2"
S, C = (4096, 2) if shape == "c2" else (8192, 1)
T = 65536
bs = {}
for name, fl in (("two arrays", cm.OUT_PCM | cm.VU), ("in place", cm.OUT_PCM | cm.VU | cm.INPLACE)):
    b = cm.Batch(S, C, T, flags=fl)
    b.set_gain(-1, C, 1000, [750, 1250][:C] if C == 2 else [900])
    if C == 2:
        b.set_chmap(-1, [1, 0])
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
    print(f"{shape} {name:10s}: median {m:.4f} ms  min {min(v):.4f}  -> {S*C*T*4/(m*1e-3)/1e9:.0f} GB/s")
