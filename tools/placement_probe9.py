#!/usr/bin/env python3
"""Does the run without the VU window see the placement classes of placement_probe3?  N input x N output arrays,
every pair timed with the config-2 kernel with and without its window (same arrays, same process)."""
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
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8


def dmalloc(n):
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), n) == 0
    return p.value


def make(flags):
    b = cm.Batch(S, Cn, T, flags=flags | cm.EXTSLOTS)
    b.set_gain(-1, 2, 1000, [750, 1250])
    b.set_chmap(-1, [1, 0])
    return b


forms = [("PCM + VU", make(cm.OUT_PCM | cm.VU)), ("PCM only", make(cm.OUT_PCM)), ("VU only", make(cm.VU))]
host = np.random.default_rng(1).integers(-32768, 32767, size=BYTES // 2, dtype=np.int16)
arr = [dmalloc(BYTES) for _ in range(2 * N)]
hip.hipMemcpy(arr[0], host.ctypes.data, BYTES, 1)
for a in arr[1:]:
    hip.hipMemcpy(a, arr[0], BYTES, 3)
ins, outs = arr[0::2], arr[1::2]


def timed(b, pi, po, warm=6, n=30):
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


timed(forms[0][1], ins[0], outs[0], 400, 10)
for name, b in forms:
    print(name)
    for i in range(N):
        if name == "VU only":
            print("in%-2d " % i + "%.4f" % timed(b, ins[i], None), flush=True)
        else:
            print("in%-2d " % i + "  ".join("%.4f" % timed(b, ins[i], outs[j]) for j in range(N)), flush=True)
