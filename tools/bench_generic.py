#!/usr/bin/env python3
"""Kernel time of the many-channel kernels (k_run_wide, k_run_rows) for a few channel counts,
with identity maps and with a channel map (rotation by one channel)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

cm = ge.load_package()
T = 16384
for C in (3, 5, 6, 7, 12, 15, 4, 8, 16):
    S = (1 << 28) // (T * C)          # ~0.5 GB of PCM
    for flags, name, bps, mapped in ((cm.OUT_PCM | cm.VU, "pcm+vu", 4, False), (cm.VU, "vu only", 2, False),
                                     (cm.OUT_PCM | cm.VU, "pcm+vu, mapped", 4, True)):
        b = cm.Batch(S, C, T, flags=flags)
        b.set_gain(-1, 1, 1000, [900])
        if mapped:
            b.set_chmap(-1, [(c + 1) % C for c in range(C)])
        b.generate(cm.GEN_NOISE, 1, T)
        for _ in range(300):                   # ~0.1 s: the clocks the chip then holds
            b.run(T)
        b.sync()
        b.timing(True)
        b.timing_read()
        for _ in range(100):
            b.run(T)
        ms, n = b.timing_read()
        gbs = S * C * T * bps / (ms / n * 1e-3) / 1e9
        print(f"C={C:2d} S={S:5d} {name:15s} {ms/n:8.3f} ms  {gbs:7.0f} GB/s")
        b.close()
