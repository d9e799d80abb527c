#!/usr/bin/env python3
"""Per-role busy cycles of the pipelined EQ kernel (diagnostic build, `make stamps`).
Run with COOLMIC_HIP_LIB=libcoolmic-dsp_amd/lib/libcoolmic-dsp-hip-stamps.so.  usage: eq_stamps.py [SHAPE]
SHAPE: eq3 (config 3: mono float planes, default) | eq3vu1 (mono int16 + VU) | eq3vu (stereo int16 + VU) |
eq3f6 (5.1 float planes) | eq3vu6 (5.1 int16 + VU)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

cm = ge.load_package()
SHAPES = {"eq3": (8192, 1, cm.OUT_F32), "eq3vu1": (8192, 1, cm.OUT_PCM | cm.VU), "eq3vu": (4096, 2, cm.OUT_PCM | cm.VU),
          "eq3f6": (1365, 6, cm.OUT_F32), "eq3vu6": (1365, 6, cm.OUT_PCM | cm.VU),
          "eq3f4": (2048, 4, cm.OUT_F32), "eq3f3": (2730, 3, cm.OUT_F32), "eq3f16": (512, 16, cm.OUT_F32)}
shape = sys.argv[1] if len(sys.argv) > 1 else "eq3"
S, Cn, out_flags = SHAPES[shape]
T = 65536
b = cm.Batch(S, Cn, T, flags=cm.EQ | out_flags)
b.set_eq(-1, cm.eq3())
b.set_gain(-1, 1, 1000, [900])
b.generate(cm.GEN_NOISE, 12345, T)
for _ in range(int(os.environ.get("EQ_STAMP_RUNS", "60"))):      # (sustained clocks; the stamps are the last launch's)
    b.run(T)
b.sync()
out = (C.c_uint64 * 64)()
cm.lib.cmhip_debug_read.argtypes = [C.c_void_p, C.c_void_p]
assert cm.lib.cmhip_debug_read(b.h, out) == 0
nsteps = out[48]
print("shape", shape, "(%d streams x %d ch)" % (S, Cn), "steps", nsteps)
for w in range(12):
    if out[2 * w + 1] > nsteps:
        hw = out[36 + w]
        role = {0: "R", 1: "Tin", 2: "Tff", 3: "S", 4: "Tff+S"}[(out[24 + w] >> 4) & 7] + str(out[24 + w] & 15)
        print(f"wave {w:2d} {role:6s}: busy {out[2*w]/nsteps:8.1f} clk/step   total {out[2*w+1]/nsteps:8.1f} clk/step   "
              f"SIMD {(hw >> 4) & 3}  CU {(hw >> 8) & 15}  wave slot {hw & 15}")
for i in range(2):
    p = [out[50 + 3 * i + j] / nsteps for j in range(3)]
    if any(p):
        print(f"T wave {i}: PCM arrived at {p[0]:7.1f}, converted at {p[1]:7.1f}, F_0 queued at {p[2]:7.1f} clk into the step")
last = []
for w in range(12):
    word = out[56 + w // 2]
    last.append((word >> (32 * (w & 1))) & 0xffffffff)
if sum(last):
    print("last at the barrier, share of the steps: " + "  ".join("w%d %.0f%%" % (w, 100.0 * c / sum(last)) for w, c in enumerate(last) if c))
    print("first to last arrival at the barrier: %.0f clk per step on average" % (out[62] / max(sum(last), 1)))
