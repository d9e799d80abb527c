#!/usr/bin/env python3
"""Does a temporary spacer allocation between the input and the output array put them into different placement
classes (placement_probe3/4)?  For every spacer size: four (input, output) pairs, output allocated while the
spacer is held, spacer freed afterwards; time of the config-2 kernel on each pair."""
import ctypes as C
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cm = importlib.import_module("libcoolmic-dsp_amd")
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

S, Cn, T = 4096, 2, 65536
BYTES = S * Cn * T * 2
GB = 1 << 30


def dmalloc(n):
    p = C.c_void_p()
    rc = hip.hipMalloc(C.byref(p), n)
    assert rc == 0, rc
    return p.value


b = cm.Batch(S, Cn, T, flags=cm.OUT_PCM | cm.VU | cm.EXTSLOTS)
b.set_gain(-1, 2, 1000, [750, 1250])
b.set_chmap(-1, [1, 0])
host = np.random.default_rng(1).integers(-32768, 32767, size=BYTES // 2, dtype=np.int16)


def timed(pi, po, warm=10, n=60):
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


pairs = []
for spacer in [int(x) for x in (sys.argv[1:] or "0 8 16 32 48 0 32".split())]:
    for k in range(4):
        pi = dmalloc(BYTES)
        hip.hipMemcpy(pi, host.ctypes.data, BYTES, 1)
        sp = dmalloc(spacer * GB) if spacer else None
        po = dmalloc(BYTES)
        if sp:
            hip.hipFree(sp)
        pairs.append((spacer, pi, po))
timed(pairs[0][1], pairs[0][2], 400, 10)
for r in range(2):
    for spacer, pi, po in pairs:
        print("round %d spacer %2d GiB  in %012x out %012x  %.4f ms" % (r, spacer, pi, po, timed(pi, po)), flush=True)
