#!/usr/bin/env python3
"""Kernel time of the planar-float output variants (the Vorbis feeder's format)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

cm = ge.load_package()
T = 65536
for name, S, C in (("stereo", 2048, 2), ("mono", 4096, 1)):
    for flags, what, bps in ((cm.OUT_F32 | cm.VU, "f32 + VU", 6), (cm.OUT_F32 | cm.OUT_PCM | cm.VU, "f32 + pcm + VU", 8),
                             (cm.OUT_F32, "f32 only", 6)):
        b = cm.Batch(S, C, T, flags=flags)
        b.set_gain(-1, C, 1000, [750, 1250][:C] if C == 2 else [900])
        b.generate(cm.GEN_NOISE, 1, T)
        for _ in range(300):      # ~0.1 s of load: the clocks the chip then holds
            b.run(T)
        b.sync()
        b.timing(True)
        b.timing_read()
        for _ in range(100):
            b.run(T)
        ms, n = b.timing_read()
        print(f"{name:6s} S={S} {what:15s} {ms/n:7.3f} ms  {S*C*T*bps/(ms/n*1e-3)/1e9:7.0f} GB/s ({bps} B/sample)")
        b.close()
