#!/usr/bin/env python3
"""bench.py's timed loop in isolation, to find what makes vu_snapshot() slow at Python level."""
import os, sys, time, ctypes as C
mods = sys.argv[1:]
if "numpy" in mods:
    import numpy as np
if "torch" in mods:
    import torch
if "dist" in mods:
    import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
cm = ge.load_package()
if "torch" in mods:
    torch.cuda.set_device(0)
S, Cn, T = 4096, 2, 65536
b = cm.Batch(S, Cn, T, flags=cm.OUT_PCM | cm.VU, device=0)
b.set_gain(-1, 2, 1000, [750, 1250]); b.set_chmap(-1, [1, 0])
b.generate(cm.GEN_NOISE, 12345, T); b.sync()
results = (cm.VuResult * S)(); rcs = (C.c_int * S)()

fn = cm.lib.cmhip_batch_vu_snapshot
gaps = []
hh = b.h

def run_steps(n):
    pending = False
    acc = [0.0, 0.0, 0.0]
    global accd
    accd = [0.0]
    for _ in range(n):
        tp0 = time.perf_counter()
        b.run(T)
        tp1 = time.perf_counter()
        if "dummy" in mods:
            cm.lib.cmhip_version()
            tpd = time.perf_counter()
            accd[0] += tpd - tp1
            tp1 = tpd
        if "pyspin" in mods:
            x = 0
            for _i in range(200):
                x += _i
            tpd = time.perf_counter()
            accd[0] += tpd - tp1
            tp1 = tpd
        if "raw" in mods:
            ta = time.clock_gettime_ns(time.CLOCK_MONOTONIC)
            fn(hh)
            tb = time.clock_gettime_ns(time.CLOCK_MONOTONIC)
            e1, e2 = C.c_longlong(), C.c_longlong()
            cm.lib.cmhip_debug_times(C.byref(e1), C.byref(e2))
            gaps.append((e1.value - ta, e2.value - e1.value, tb - e2.value))
        else:
            b.vu_snapshot()
        tp2 = time.perf_counter()
        if pending:
            b.vu_collect(results, rcs)
        pending = True
        tp3 = time.perf_counter()
        acc[0] += tp1 - tp0; acc[1] += tp2 - tp1; acc[2] += tp3 - tp2
    if pending:
        b.vu_collect(results, rcs)
    b.sync()
    return [v / n * 1e6 for v in acc]

run_steps(3)
if "timing" in mods:
    b.timing(True); b.timing_read()
if "sync" in mods:
    torch.cuda.synchronize()
t0 = time.perf_counter()
acc = run_steps(20)
dt = time.perf_counter() - t0
import threading
print("python threads:", len(threading.enumerate()), "os threads:", len(os.listdir("/proc/self/task")), "dummy/pyspin us:", round(accd[0] / 20 * 1e6))
print(" ".join(mods) or "-", f": {dt / 20 * 1e3:.4f} ms/step; run %.0f snapshot %.0f collect %.0f us" % tuple(acc))
if gaps:
    g = gaps[-10:]
    print("before entry / inside / after exit (us):", [tuple(round(v / 1000) for v in t) for t in g])
