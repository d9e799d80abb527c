#!/usr/bin/env python3
"""Where the config-2 kernel's arrays lie against its speed: one EXTSLOTS batch, PCM arrays named per run
(cmhip_batch_run_slots) inside slabs this script allocates with hipMalloc, input and output at chosen
offsets.  Kernel time from the dispatch's own events, blocks of launches interleaved over the placements."""
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


def dmalloc(n):
    p = C.c_void_p()
    rc = hip.hipMalloc(C.byref(p), n)
    assert rc == 0, rc
    return p.value


b = cm.Batch(S, Cn, T, flags=cm.OUT_PCM | cm.VU | cm.EXTSLOTS)
b.set_gain(-1, 2, 1000, [750, 1250])
b.set_chmap(-1, [1, 0])
rng = np.random.default_rng(1)
host = rng.integers(-32768, 32767, size=BYTES // 2, dtype=np.int16)

MB = 1 << 20
placements = []            # (label, in_ptr, out_ptr)
slab = dmalloc(3 * BYTES + 64 * MB)
for name, d in (("slab out=in+1G", 0), ("+256B", 256), ("+4K", 4096), ("+64K", 65536), ("+1M", MB), ("+2M", 2 * MB),
                ("+3M", 3 * MB), ("+16M", 16 * MB), ("+1G", BYTES)):
    placements.append((name, slab, slab + BYTES + d))
for i in range(4):
    pi, po = dmalloc(BYTES), dmalloc(BYTES)
    placements.append(("malloc pair %d" % i, pi, po))
done = set()
for _, pi, _ in placements:
    if pi not in done:
        hip.hipMemcpy(pi, host.ctypes.data, BYTES, 1)
        done.add(pi)

for _ in range(400):
    b.run_slots(T, placements[0][1], placements[0][2])
b.sync()
res = {p[0]: [] for p in placements}
for r in range(3):
    for name, pi, po in placements:
        for _ in range(30):
            b.run_slots(T, pi, po)
        b.sync()
        b.timing(True)
        b.timing_read()
        for _ in range(150):
            b.run_slots(T, pi, po)
        ms, n = b.timing_read()
        b.timing(False)
        res[name].append(ms / n)
for name, pi, po in placements:
    print("%-16s in %012x out %012x  d=%+d MiB rem %7d  %s" % (name, pi, po, (po - pi) // MB, (po - pi) % MB,
                                                             "  ".join("%.4f" % v for v in res[name])), flush=True)
