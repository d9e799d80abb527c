#!/usr/bin/env python3
"""Planar-float output of the many-channel kernels (k_run_wide / k_run_rows): 2 B in + 4 B out."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
cm = ge.load_package()
T = 16384
for C in (3, 4, 6, 8, 16):
    S = (1 << 28) // (T * C)
    for flags, name, bps in ((cm.OUT_F32, "f32 only", 6), (cm.OUT_F32 | cm.VU, "f32 + VU", 6)):
        b = cm.Batch(S, C, T, flags=flags)
        b.set_gain(-1, 1, 1000, [900])
        b.generate(cm.GEN_NOISE, 1, T)
        for _ in range(300):      # ~0.1 s of load: the clocks the chip then holds
            b.run(T)
        b.sync(); b.timing(True); b.timing_read()
        for _ in range(100):
            b.run(T)
        ms, n = b.timing_read()
        print(f"C={C:2d} S={S:5d} {name:10s} {ms/n:8.3f} ms  {S*C*T*bps/(ms/n*1e-3)/1e9:7.0f} GB/s")
        b.close()
