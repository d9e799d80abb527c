#!/usr/bin/env python3
"""Kernel time of the many-channel path (k_run_generic) for a few channel counts."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

cm = ge.load_package()
T = 16384
for C in (3, 6, 4, 8, 16):
    S = (1 << 28) // (T * C)          # ~0.5 GB of PCM
    for flags, name, bps in ((cm.OUT_PCM | cm.VU, "pcm+vu", 4), (cm.VU, "vu only", 2)):
        b = cm.Batch(S, C, T, flags=flags)
        b.set_gain(-1, 1, 1000, [900])
        b.generate(cm.GEN_NOISE, 1, T)
        for _ in range(2):
            b.run(T)
        b.sync()
        b.timing(True)
        b.timing_read()
        for _ in range(5):
            b.run(T)
        ms, n = b.timing_read()
        gbs = S * C * T * bps / (ms / n * 1e-3) / 1e9
        print(f"C={C:2d} S={S:5d} {name:8s} {ms/n:8.3f} ms  {gbs:7.0f} GB/s")
        b.close()
