#!/usr/bin/env python3
"""A/B in one process, interleaved, at sustained clocks: the EQ kernel's T-in waves with the general gain
form ($CMHIP_EQ_GENERAL_GAIN, read when a batch is created) against the short form for gains below
the scale.  Config 3 shape, float planes, and the stereo int16 + VU form."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

cm = ge.load_package()
T = 65536
for name, S, C, flags, bps in (("c3 float planes", 8192, 1, cm.EQ | cm.OUT_F32, 6),
                               ("stereo int16+VU", 4096, 2, cm.EQ | cm.OUT_PCM | cm.VU, 4)):
    bs = {}
    for variant in ("general", "short"):
        if variant == "general":
            os.environ["CMHIP_EQ_GENERAL_GAIN"] = "1"
        else:
            os.environ.pop("CMHIP_EQ_GENERAL_GAIN", None)
        b = cm.Batch(S, C, T, flags=flags)
        b.set_eq(-1, cm.eq3())
        b.set_gain(-1, 1, 1000, [900])
        b.generate(cm.GEN_NOISE, 12345, T)
        bs[variant] = b
    for b in bs.values():
        for _ in range(100):
            b.run(T)
        b.sync()
    res = {k: [] for k in bs}
    for rnd in range(4):
        for k, b in bs.items():
            b.timing(True)
            b.timing_read()
            for _ in range(60):
                b.run(T)
            ms, n = b.timing_read()
            b.timing(False)
            res[k].append(ms / n)
    for k, v in res.items():
        best = min(v)
        print(f"{name:18s} {k:8s}: " + " ".join(f"{x:.4f}" for x in v) +
              f" ms   best {best:.4f} ms = {S * C * T * bps / best / 1e6:7.1f} GB/s", flush=True)
    for b in bs.values():
        b.close()
