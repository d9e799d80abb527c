#!/usr/bin/env python3
"""Classes of placement inside one slab (placement_slab): inputs at several offsets x outputs every STEP GiB.
A pair is slow when both arrays are in the same class; the rows show which stretches of the slab share one."""
import ctypes as C
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cm = importlib.import_module("libcoolmic-dsp_amd")
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

S, Cn, T = 4096, 2, 65536
BYTES = S * Cn * T * 2
GB = 1 << 30
SLAB = int(sys.argv[1]) if len(sys.argv) > 1 else 200
STEP = int(sys.argv[2]) if len(sys.argv) > 2 else 4
INS = [int(x) for x in (sys.argv[3:] or "0 20 40 60 90 120 156 176".split())]
p = C.c_void_p()
assert hip.hipMalloc(C.byref(p), SLAB * GB) == 0
slab = p.value
b = cm.Batch(S, Cn, T, flags=cm.OUT_PCM | cm.VU | cm.EXTSLOTS)
b.set_gain(-1, 2, 1000, [750, 1250])
b.set_chmap(-1, [1, 0])
host = np.random.default_rng(1).integers(-32768, 32767, size=BYTES // 2, dtype=np.int16)


def timed(pi, po, warm=5, n=24):
    for _ in range(warm):
        b.run_slots(T, pi, po)
    b.sync()
    b.timing(True)
    b.timing_read()
    for _ in range(n):
        b.run_slots(T, pi, po)
    ms, k = b.timing_read()
    b.timing(False)
    return ms / k


hip.hipMemcpy(slab, host.ctypes.data, BYTES, 1)
timed(slab, slab + BYTES, 400, 10)
outs = list(range(0, SLAB - 1, STEP))
print("slab %012x; columns: output at GiB " % slab + " ".join("%3d" % o for o in outs))
for i in INS:
    src = slab + i * GB
    hip.hipMemcpy(src, host.ctypes.data, BYTES, 1)
    row = []
    for o in outs:
        row.append(float("nan") if abs(o - i) < 1 else timed(src, slab + o * GB))
    lo = np.nanmin(row)
    print("in at %3d GiB (fastest %.4f): " % (i, lo) + " ".join("  ." if v != v else ("  x" if v > lo * 1.035 else "  -") for v in row), flush=True)
