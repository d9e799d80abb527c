#!/usr/bin/env python3
"""Throughput of the per-stream operator chain (config 1: sine -> transform -> vumeter) with
the reference's 1024-byte pulls: what a single pipeline pays per read (launch + copies)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import __graft_entry__ as ge

cm = ge.load_package()
for gain in (None, (1, 1000, [900])):
    dev = cm.Snddev("sine", 48000, 1)
    tr = cm.Transform(48000, 1)
    h0 = dev.get_iohandle()
    tr.attach(h0); h0.unref(); dev.unref()
    h = tr.get_iohandle()
    vu = cm.Vumeter(48000, 1)
    vu.attach(h)
    if gain:
        tr.set_master_gain(*gain)
    for _ in range(50):
        vu.read(-1)
    n = 2000
    t0 = time.perf_counter()
    for _ in range(n):
        vu.read(-1)
    dt = time.perf_counter() - t0
    rc, r = vu.result()
    print(f"gain {'on ' if gain else 'off'}: {dt / n * 1e6:7.1f} us per 1024-byte read, {n * 512 / dt / 1e6:6.2f} Msamples/s "
          f"(rc {rc}, frames {r.frames})")
    h.unref(); vu.unref(); tr.unref()
