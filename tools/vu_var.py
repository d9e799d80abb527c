#!/usr/bin/env python3
"""Per-launch time of the read-only stereo kernel on an otherwise idle chip: it is VALU-bound and follows the
GPU's clock (first launches ~0.158 ms, then up to 0.25 ms, settling at 0.18-0.20 ms).  No arguments."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
cm = ge.load_package()
S, C, T = 4096, 2, 65536
b = cm.Batch(S, C, T, flags=cm.VU)
b.set_gain(-1, 2, 1000, [750, 1250]); b.set_chmap(-1, [1, 0])
b.generate(cm.GEN_NOISE, 12345, T)
for _ in range(3): b.run(T)
b.sync()
b.timing(True); b.timing_read()
out = []
for i in range(40):
    b.run(T)
    ms, n = b.timing_read()
    out.append(ms / n)
    if i % 10 == 9: b.vu_reset(-1)
print(" ".join(f"{v*1000:.0f}" for v in out))
b.timing(False)
# back to back without reading in between
b.timing(True); b.timing_read()
for i in range(40): b.run(T)
ms, n = b.timing_read()
print("40 back to back:", ms / n)
