#!/usr/bin/env python3
"""PCIe-inclusive rate of the batch path (never bench.py's `value`): pinned host PCM ->
upload_all -> fused kernel -> download_all, double buffered over two batches so that the
copies of one block overlap the kernel and copies of the other."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

cm = ge.load_package()
S, C, T = 4096, 2, 16384           # 256 MiB in + 256 MiB out per block
steps = 12
bs, hin, hout = [], [], []
for i in range(2):
    b = cm.Batch(S, C, T, flags=cm.OUT_PCM | cm.VU)
    b.set_gain(-1, 2, 1000, [750, 1250])
    b.set_chmap(-1, [1, 0])
    bs.append(b)
    hin.append(cm.PinnedPcm(b))
    hout.append(cm.PinnedPcm(b))
rng = np.random.default_rng(1)
for h in hin:
    h.array[:] = rng.integers(-32768, 32768, size=h.shape, dtype=np.int64).astype(np.int16)

def step(i):
    b = bs[i & 1]
    b.sync()                        # the previous use of this buffer pair has finished
    b.upload_all(hin[i & 1].ptr, T)
    b.run(T)
    b.download_all(hout[i & 1].ptr, T)

for i in range(4):
    step(i)
for b in bs:
    b.sync()
t0 = time.perf_counter()
for i in range(steps):
    step(i)
for b in bs:
    b.sync()
dt = time.perf_counter() - t0
samples = S * C * T * steps
print(f"PCIe-inclusive: {samples / dt / 1e6:.0f} Msamples/s  "
      f"({samples * 2 / dt / 1e9:.1f} GB/s each way, {dt / steps * 1e3:.2f} ms per {S}x{C}x{T} block)")
# (parity of this path: tests/test_gpu_parity.py::test_host_resident_slots and the upload_all / download_all
# tests -- the oracle is the tests' checker, tools do not load it)
