#!/usr/bin/env python3
"""Which of many 1 GiB hipMalloc arrays are in the class that makes the config-2 kernel faster (placement_probe3)?
N arrays; array REF is the input and every other array the output, then the other way round."""
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
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
REF = N // 2


def dmalloc(n):
    p = C.c_void_p()
    rc = hip.hipMalloc(C.byref(p), n)
    assert rc == 0, rc
    return p.value


b = cm.Batch(S, Cn, T, flags=cm.OUT_PCM | cm.VU | cm.EXTSLOTS)
b.set_gain(-1, 2, 1000, [750, 1250])
b.set_chmap(-1, [1, 0])
host = np.random.default_rng(1).integers(-32768, 32767, size=BYTES // 2, dtype=np.int16)
arr = [dmalloc(BYTES) for _ in range(N)]
hip.hipMemcpy(arr[REF], host.ctypes.data, BYTES, 1)
for a in arr:
    if a != arr[REF]:
        hip.hipMemcpy(a, arr[REF], BYTES, 3)
for _ in range(400):
    b.run_slots(T, arr[REF], arr[0])
b.sync()


def timed(pi, po):
    for _ in range(6):
        b.run_slots(T, pi, po)
    b.sync()
    b.timing(True)
    b.timing_read()
    for _ in range(30):
        b.run_slots(T, pi, po)
    ms, n = b.timing_read()
    b.timing(False)
    return ms / n


fw = [timed(arr[REF], arr[i]) if i != REF else float("nan") for i in range(N)]
bw = [timed(arr[i], arr[REF]) if i != REF else float("nan") for i in range(N)]
for i in range(N):
    print("%3d va %012x  ref->x %.4f  x->ref %.4f" % (i, arr[i], fw[i], bw[i]), flush=True)
