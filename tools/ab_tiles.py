#!/usr/bin/env python3
"""Interleaved A/B of the VU-only tile size in ONE process (cdna_hip_programming.md rule 24):
rounds x variants, median and min of the kernel time from HIP events on the batch's stream."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

cm = ge.load_package()
T = 65536
for name, S, C in (("stereo 4096x2", 4096, 2), ("mono 8192x1", 8192, 1)):
    b = cm.Batch(S, C, T, flags=cm.VU)
    if os.environ.get("AB_TILES_GAIN", "1") != "0":
        b.set_gain(-1, C, 1000, [750, 1250][:C] if C == 2 else [1100])
    if C == 2:
        b.set_chmap(-1, [1, 0])
    b.generate(cm.GEN_NOISE, 12345, T)
    res = {4: [], 8: [], 16: []}
    for _ in range(400):                      # ~0.1 s of load: the clocks the chip then holds
        b.run(T)
    for rnd in range(9):
        for tile in (4, 8, 16):
            os.environ["CMHIP_VU_TILE"] = str(tile)
            b.run(T)
            b.sync()
            b.timing(True)
            b.timing_read()
            for _ in range(40):
                b.run(T)
            ms, n = b.timing_read()
            b.timing(False)
            b.vu_reset(-1)
            res[tile].append(ms / n)
    for tile in (4, 8, 16):
        v = res[tile]
        gbs = S * C * T * 2 / (statistics.median(v) * 1e-3) / 1e9
        print(f"{name} tile {tile:2d} x 1 KiB/lane-vector: median {statistics.median(v):.4f} ms "
              f"min {min(v):.4f} ms  -> {gbs:.0f} GB/s")
    b.close()
