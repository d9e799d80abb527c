#!/usr/bin/env python3
"""{window merged by workgroups of 1, 4, 8, 16 waves; no window} x {input and output arrays of the same placement class, of different
ones}: the config-2 kernel in one process, on the slowest and the fastest of N x N array pairs.
PF_SNAP=1: a VU window per launch (snapshot + collect every step, as bench.py runs); PF_NOVU=1: the other builds
named on the command line run without a window (PCM only)."""
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
S, Cn, T = [int(x) for x in os.environ.get("PF_SHAPE", "4096,2,65536").split(",")]    # e.g. PF_SHAPE=2730,6,16384
BYTES = S * Cn * T * 2
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6


def dmalloc(n):
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), n) == 0
    return p.value


def make(flags, env=None):
    if env:
        os.environ[env[0]] = env[1]
    b = cm.Batch(S, Cn, T, flags=flags | cm.EXTSLOTS)
    if env:
        os.environ.pop(env[0])
    if Cn == 2:
        b.set_gain(-1, 2, 1000, [750, 1250])
        b.set_chmap(-1, [1, 0])
    else:
        b.set_gain(-1, 1, 1000, [900])
    return b


SNAP = bool(os.environ.get("PF_SNAP"))          # a VU window per launch (snapshot + collect every step), as bench.py runs
RO = len(sys.argv) > 2 and sys.argv[2] == "ro"          # the read-only runs (VU only): no output array, no pairs
F32 = cm.OUT_F32 if os.environ.get("PF_F32") else 0            # float planes beside the PCM result
forms = [("NW=%s" % n, make(cm.VU if RO else cm.OUT_PCM | cm.VU | F32, ("CMHIP_FAST_NW", n))) for n in (("1", "4", "8") if Cn <= 2 else ("1",))]
if not RO:
    forms.insert(1, ("no window", make(cm.OUT_PCM | F32)))
for extra in sys.argv[3:]:                # other builds of the library (timing-only variants), on the same arrays
    import importlib.util
    os.environ["COOLMIC_HIP_LIB"] = os.path.abspath(extra)
    tag = "cm_x%d" % len(forms)
    spec = importlib.util.spec_from_file_location(tag, os.path.join(os.path.dirname(cm.__file__), "__init__.py"),
                                                  submodule_search_locations=[os.path.dirname(cm.__file__)])
    cm2 = importlib.util.module_from_spec(spec)
    sys.modules[tag] = cm2
    spec.loader.exec_module(cm2)
    b2 = cm2.Batch(S, Cn, T, flags=(cm2.VU if RO else (cm2.OUT_PCM if os.environ.get("PF_NOVU") else cm2.OUT_PCM | cm2.VU)) | cm2.EXTSLOTS)
    if Cn == 2:
        b2.set_gain(-1, 2, 1000, [750, 1250])
        b2.set_chmap(-1, [1, 0])
    else:
        b2.set_gain(-1, 1, 1000, [900])
    forms.append((os.path.basename(extra)[-24:], b2))
host = np.random.default_rng(1).integers(-32768, 32767, size=BYTES // 2, dtype=np.int16)
arr = [dmalloc(BYTES) for _ in range(2 * N)]
hip.hipMemcpy(arr[0], host.ctypes.data, BYTES, 1)
for a in arr[1:]:
    hip.hipMemcpy(a, arr[0], BYTES, 3)
ins, outs = arr[0::2], arr[1::2]


def step(b, pi, po):
    b.run_slots(T, pi, po)
    if SNAP and (b.flags & cm.VU):
        b.vu_snapshot()
        b.vu_collect()


def timed(b, pi, po, warm=6, n=30):
    for _ in range(warm):
        step(b, pi, po)
    b.sync()
    b.timing(True)
    b.timing_read()
    for _ in range(n):
        step(b, pi, po)
    ms, k = b.timing_read()
    b.timing(False)
    return ms / k


import time


def wall(b, pi, po, n=300):
    for _ in range(20):
        b.run_slots(T, pi, po)
    b.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        b.run_slots(T, pi, po)
    b.sync()
    return (time.perf_counter() - t0) / n * 1e3


if RO:
    timed(forms[0][1], ins[0], None, 400, 10)
    for name, b in forms:
        r = [[timed(b, ins[i], None, 10, 60) for i in range(3)] for _ in range(3)]
        print("%-8s kernel, three input arrays: " % name + "  ".join("%.4f" % (sum(x[i] for x in r) / 3) for i in range(3)), flush=True)
    sys.exit(0)
timed(forms[0][1], ins[0], outs[0], 400, 10)
import statistics
for name, b in forms:
    ts = [timed(b, ins[i], outs[j], 5, 30) for i in range(N) for j in range(N)]
    print("%-10s over %d array pairs: mean %.4f  median %.4f  min %.4f  max %.4f ms" % (name, len(ts), statistics.mean(ts), statistics.median(ts), min(ts), max(ts)), flush=True)
