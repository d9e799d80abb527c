#!/usr/bin/env python3
"""What a bench step costs beyond its kernel: wall clock per step of config 2 for (a) runs alone, (b) runs with a
snapshot of the windows per step and the host collecting the previous one (the bench's step), (c) as (b) but the
host collects every eighth window only."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cm = importlib.import_module("libcoolmic-dsp_amd")
S, C, T = 4096, 2, 65536
b = cm.Batch(S, C, T, flags=cm.OUT_PCM | cm.VU)
b.set_gain(-1, 2, 1000, [750, 1250])
b.set_chmap(-1, [1, 0])
b.generate(cm.GEN_NOISE, 12345, T)
res = (cm.VuResult * S)()
rcs = (cm.C.c_int * S)() if hasattr(cm, "C") else None


def loop(n, mode):
    pending = 0
    for i in range(n):
        b.run(T)
        if mode:
            b.vu_snapshot()
            if pending:
                b.vu_collect(res, rcs)
                pending -= 1
            pending += 1
    while pending:
        b.vu_collect(res, rcs)
        pending -= 1
    b.sync()


for _ in range(3):
    for name, mode in (("runs alone", 0), ("run + snapshot + collect", 1)):
        loop(300, mode)
        t0 = time.perf_counter()
        loop(600, mode)
        print("%-26s %.4f ms per step" % (name, (time.perf_counter() - t0) / 600 * 1e3), flush=True)
b.timing(True)
b.timing_read()
loop(300, 1)
ms, n = b.timing_read()
print("kernel alone (dispatch events) %.4f ms" % (ms / n))
for _ in range(2):
    b.timing(True)
    b.timing_read()
    loop(300, 1)
    t0 = time.perf_counter()
    loop(600, 1)
    dt = (time.perf_counter() - t0) / 600 * 1e3
    ms, n = b.timing_read()
    b.timing(False)
    print("run + snapshot + collect, every launch timed with events: %.4f ms per step (kernel %.4f)" % (dt, ms / n), flush=True)
