#!/usr/bin/env python3
"""One large slab, input at its start, output k GiB further on: is the placement class (placement_matrix/4)
periodic in the distance?  usage: placement_slab.py [slab GiB] [step MiB]"""
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
GB, MB = 1 << 30, 1 << 20
SLAB = int(sys.argv[1]) if len(sys.argv) > 1 else 64
STEP = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
IN_AT = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # input at this many GiB into the slab
p = C.c_void_p()
assert hip.hipMalloc(C.byref(p), SLAB * GB) == 0
slab = p.value
b = cm.Batch(S, Cn, T, flags=cm.OUT_PCM | cm.VU | cm.EXTSLOTS)
b.set_gain(-1, 2, 1000, [750, 1250])
b.set_chmap(-1, [1, 0])
host = np.random.default_rng(1).integers(-32768, 32767, size=BYTES // 2, dtype=np.int16)
src = slab + IN_AT * GB
hip.hipMemcpy(src, host.ctypes.data, BYTES, 1)


def timed(pi, po, warm=6, n=30):
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


timed(src, slab + (BYTES if IN_AT == 0 else 0), 400, 10)
print("slab %012x, %d GiB, input at +%d GiB" % (slab, SLAB, IN_AT))
off = 0
while off + BYTES <= SLAB * GB:
    if off + BYTES <= IN_AT * GB or off >= IN_AT * GB + BYTES:
        print("out at +%6d MiB  %.4f ms" % (off // MB, timed(src, slab + off)), flush=True)
    off += STEP * MB
