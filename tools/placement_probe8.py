#!/usr/bin/env python3
"""Which forms of the stereo run see the placement class (placement_probe6)?  One slab; input at its start;
for every form the time with the output 16 GiB and 40 GiB further on, interleaved."""
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
p = C.c_void_p()
assert hip.hipMalloc(C.byref(p), 64 * GB) == 0
slab = p.value
host = np.random.default_rng(1).integers(-32768, 32767, size=BYTES // 2, dtype=np.int16)
hip.hipMemcpy(slab, host.ctypes.data, BYTES, 1)


def timed(b, pi, po, warm=10, n=60):
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


forms = []
for name, flags, gain, swap in (("gain + swap, PCM + VU (config 2)", cm.OUT_PCM | cm.VU, [750, 1250], True),
                                ("identity, PCM + VU", cm.OUT_PCM | cm.VU, None, False),
                                ("identity, PCM only", cm.OUT_PCM, None, False),
                                ("gain, PCM only", cm.OUT_PCM, [750, 1250], False)):
    b = cm.Batch(S, Cn, T, flags=flags | cm.EXTSLOTS)
    if gain:
        b.set_gain(-1, 2, 1000, gain)
    if swap:
        b.set_chmap(-1, [1, 0])
    forms.append((name, b))
timed(forms[0][1], slab, slab + 16 * GB, 400, 10)
# find a near and a far position of different speed for the first form
scan = {o: timed(forms[0][1], slab, slab + o * GB, 5, 20) for o in range(4, 61, 4)}
near, far = max(scan, key=scan.get), min(scan, key=scan.get)
print("scan: " + "  ".join("%d:%.4f" % kv for kv in scan.items()))
for name, b in forms:
    ts = [(timed(b, slab, slab + near * GB), timed(b, slab, slab + far * GB)) for _ in range(3)]
    a = sum(t[0] for t in ts) / 3
    f = sum(t[1] for t in ts) / 3
    print("%-36s out +%d GiB %.4f ms   out +%d GiB %.4f ms   %+.1f %%" % (name, near, a, far, f, (f / a - 1) * 100), flush=True)
