#!/usr/bin/env python3
"""Where the arrays lie against the speed of the config-2 kernel, as a matrix: N input arrays x N output arrays (separate hipMallocs), the config-2
kernel on every pair.  Does the time follow the input array, the output array, or the pair?"""
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
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5
WARM, TIMED = (20, 100) if N <= 6 else (8, 40)
FLAGS = cm.OUT_PCM | cm.VU if len(sys.argv) <= 2 else int(sys.argv[2], 0)


def dmalloc(n):
    p = C.c_void_p()
    rc = hip.hipMalloc(C.byref(p), n)
    assert rc == 0, rc
    return p.value


b = cm.Batch(S, Cn, T, flags=FLAGS | cm.EXTSLOTS)
b.set_gain(-1, 2, 1000, [750, 1250])
b.set_chmap(-1, [1, 0])
host = np.random.default_rng(1).integers(-32768, 32767, size=BYTES // 2, dtype=np.int16)
ins, outs = [], []
POOL = os.environ.get("PROBE_POOL")     # one pool of arrays, each used as input and as output
for i in range(N):                      # alternating, as a batch allocates them
    ins.append(dmalloc(BYTES))
    hip.hipMemcpy(ins[-1], host.ctypes.data, BYTES, 1)
    if not POOL:
        outs.append(dmalloc(BYTES))
if POOL:
    outs = ins
for _ in range(400):
    b.run_slots(T, ins[0], outs[0])
b.sync()
t = np.zeros((N, N))
for r in range(2):
    for i in range(N):
        for j in range(N):
            if ins[i] == outs[j]:
                t[i, j] = float("nan")
                continue
            for _ in range(WARM):
                b.run_slots(T, ins[i], outs[j])
            b.sync()
            b.timing(True)
            b.timing_read()
            for _ in range(TIMED):
                b.run_slots(T, ins[i], outs[j])
            ms, n = b.timing_read()
            b.timing(False)
            t[i, j] += ms / n / 2
print("rows: input array, columns: output array; ms per launch")
print("            " + "  ".join("out%d %05x" % (j, (outs[j] >> 21) & 0xfffff) for j in range(N)))
for i in range(N):
    print("in%d %05x   " % (i, (ins[i] >> 21) & 0xfffff) + "  ".join("%10.4f" % v for v in t[i]))
thr = (np.nanmin(t) + np.nanmax(t)) / 2
print("fast pairs (x) at threshold %.4f:" % thr)
for i in range(N):
    print("in%-2d " % i + " ".join("x" if v < thr else "." for v in t[i]))
print("row means   " + "  ".join("%.4f" % v for v in np.nanmean(t, axis=1)))
print("col means   " + "  ".join("%.4f" % v for v in np.nanmean(t, axis=0)))
