#!/usr/bin/env python3
"""Kernel time of one shape with whatever library COOLMIC_HIP_LIB points at (run it for two
builds back to back on one GPU).  usage: ab_libs.py [c2|c4] [vu|pcm]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

cm = ge.load_package()
shape = sys.argv[1] if len(sys.argv) > 1 else "c2"
mode = sys.argv[2] if len(sys.argv) > 2 else "vu"
S, C = (4096, 2) if shape == "c2" else (8192, 1)
T = 65536
b = cm.Batch(S, C, T, flags=cm.VU if mode == "vu" else cm.VU | cm.OUT_PCM)
b.set_gain(-1, C, 1000, [750, 1250][:C] if C == 2 else [900])
if C == 2:
    b.set_chmap(-1, [1, 0])
b.generate(cm.GEN_NOISE, 12345, T)
v = []
for rnd in range(9):
    b.run(T)
    b.sync()
    b.timing(True)
    b.timing_read()
    for _ in range(10):
        b.run(T)
    ms, n = b.timing_read()
    b.timing(False)
    b.vu_reset(-1)
    v.append(ms / n)
bps = 2 if mode == "vu" else 4
print(f"{os.path.basename(cm.LIB_PATH):32s} {shape} {mode}: median {statistics.median(v):.4f} ms min {min(v):.4f} "
      f"-> {S*C*T*bps/(statistics.median(v)*1e-3)/1e9:.0f} GB/s")
