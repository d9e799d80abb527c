#!/usr/bin/env python3
"""PCIe-inclusive rate with the PCM slots in pinned, device-mapped host memory (CMHIP_HOSTPCM): the fused
kernel reads the block over PCIe and writes the result back over PCIe in the same launch -- both
directions at once, no copy engine.  Compare tools/bench_pcie.py (copies + device-resident slots)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

cm = ge.load_package()
S, C, T = 4096, 2, 16384           # 256 MiB in + 256 MiB out per block
for flags, name in ((cm.OUT_PCM | cm.VU | cm.HOSTPCM, "two host arrays"),
                    (cm.OUT_PCM | cm.VU | cm.HOSTPCM | cm.INPLACE, "in place"),
                    (cm.VU | cm.HOSTPCM, "VU only (read)")):
    b = cm.Batch(S, C, T, flags=flags)
    b.set_gain(-1, 2, 1000, [750, 1250])
    b.set_chmap(-1, [1, 0])
    x = np.random.default_rng(1).integers(-32768, 32768, size=T * C, dtype=np.int64).astype(np.int16)
    for s in range(0, S, 512):
        b.upload(s, x)
    for _ in range(2):
        b.run(T)
    b.sync()
    steps = 6
    t0 = time.perf_counter()
    for _ in range(steps):
        b.run(T)
    b.sync()
    dt = time.perf_counter() - t0
    samples = S * C * T * steps
    print(f"zero copy, {name:16s}: {samples / dt / 1e6:8.0f} Msamples/s  ({samples * 2 / dt / 1e9:.1f} GB/s "
          f"per direction used, {dt / steps * 1e3:.2f} ms per block)")
    b.close()
