#!/usr/bin/env python3
"""EQ kernel time against the number of biquad sections (config 3 shape, float planes), gain on / off:
what the T-in + S waves cost alone (1 section) and what every further section adds."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge

cm = ge.load_package()
S, T = 8192, 65536
coef = np.concatenate([cm.eq3(48000.0), cm.design_biquad(1, 48000.0, 3000.0, 4.0, 2.0)])
for gain in (True, False):
    for nsec in (1, 2, 3, 4):
        b = cm.Batch(S, 1, T, flags=cm.EQ | cm.OUT_F32)
        b.set_eq(-1, coef[:5 * nsec])
        if gain:
            b.set_gain(-1, 1, 1000, [900])
        b.generate(cm.GEN_NOISE, 12345, T)
        for _ in range(100):
            b.run(T)
        b.sync()
        b.timing(True)
        b.timing_read()
        for _ in range(100):
            b.run(T)
        ms, n = b.timing_read()
        print(f"sections {nsec} gain {'on ' if gain else 'off'}: {ms / n:.4f} ms  {S * T * 6 / (ms / n) / 1e6:7.1f} GB/s", flush=True)
        b.close()
