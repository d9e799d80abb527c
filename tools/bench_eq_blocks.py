#!/usr/bin/env python3
"""EQ kernel time by block length (8192 mono streams, 3 sections, float planes): the pipeline needs
2 * sections steps to fill, which short blocks pay in full."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

cm = ge.load_package()
S = 8192
for T in (64, 128, 480, 512, 1024, 4096, 16384, 65536):
    b = cm.Batch(S, 1, T, flags=cm.EQ | cm.OUT_F32)
    b.set_eq(-1, cm.eq3())
    b.set_gain(-1, 1, 1000, [900])
    b.generate(cm.GEN_NOISE, 12345, T)
    n_warm = max(20, int(0.1 / (T * 1.2e-8 + 5e-6)))
    for _ in range(n_warm):
        b.run(T)
    b.sync()
    b.timing(True)
    b.timing_read()
    for _ in range(n_warm):
        b.run(T)
    ms, n = b.timing_read()
    blocks = (T + 63) // 64
    print(f"T={T:6d}: {ms / n * 1e3:9.1f} us per launch, {ms / n * 1e3 / (blocks + 6):6.2f} us per pipeline step "
          f"({blocks} blocks + 6), {S * T * 6 / (ms / n * 1e-3) / 1e9:7.0f} GB/s")
    b.close()
