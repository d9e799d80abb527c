#!/usr/bin/env python3
"""EQ kernel time for one shape with the library $COOLMIC_HIP_LIB points at (A/B of two builds in
one gpurun call).  Usage: ab_eq.py C [f32|pcmvu] [sections]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import __graft_entry__ as ge

cm = ge.load_package()
C = int(sys.argv[1]) if len(sys.argv) > 1 else 2
mode = sys.argv[2] if len(sys.argv) > 2 else "f32"
nsec = int(sys.argv[3]) if len(sys.argv) > 3 else 3
flags = cm.EQ | (cm.OUT_F32 if mode == "f32" else cm.OUT_PCM | cm.VU)
S, T = (8192 // C) // 32 * 32 if C > 2 else 8192 // C, 65536
b = cm.Batch(S, C, T, flags=flags)
import numpy as np
coef = np.concatenate([cm.eq3(), cm.design_biquad(1, 48000.0, 3000.0, 4.0, 2.0)])[: 5 * nsec]
b.set_eq(-1, coef)
b.set_gain(-1, 1, 1000, [900])
b.generate(cm.GEN_NOISE, 12345, T)
for _ in range(100):      # the chip needs ~100 ms of load to reach its clocks (20 launches: 0.88 ms, 1000: 0.78)
    b.run(T)
b.sync()
b.timing(True)
for _ in range(100):
    b.run(T)
b.sync()
ms, n = b.timing_read()
print(f"{os.path.basename(os.environ.get('COOLMIC_HIP_LIB', 'default')):32s} C={C} S={S} {mode} nsec={nsec}: {ms / n:.4f} ms")
