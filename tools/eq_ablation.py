#!/usr/bin/env python3
"""Launch time of the pipelined EQ kernel with one part cut out (`make -C libcoolmic-dsp_amd abl`).
Timing only: the ablated builds compute wrong results by construction."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BITS = {1: "store wave idle", 2: "no global loads (synthetic PCM)", 4: "recurrence waves copy rows (no FMAs)",
        8: "recurrence waves write 1 of 16 vectors", 16: "no feed-forward of sections 1..",
        32: "recurrence waves idle", 64: "no input stage (load, gain, section 0 feed-forward)",
        128: "T waves idle"}
CHILD = r'''
import sys, os
sys.path.insert(0, %r)
import __graft_entry__ as ge
cm = ge.load_package()
S, T = 8192, 65536
b = cm.Batch(S, 1, T, flags=cm.EQ | cm.OUT_F32)
b.set_eq(-1, cm.eq3())
b.set_gain(-1, 1, 1000, [900])
b.generate(cm.GEN_NOISE, 12345, T)
for _ in range(3):
    b.run(T)
b.sync()
b.timing(True)
for _ in range(10):
    b.run(T)
b.sync()
ms, n = b.timing_read(); print("MS %%.4f ms per launch" %% (ms / n))
''' % ROOT
masks = [int(x, 0) for x in sys.argv[1:]] or [0, 1, 2, 4, 8, 16, 32, 64, 128]
for n in masks:
    lib = os.path.join(ROOT, "libcoolmic-dsp_amd", "lib",
                       "libcoolmic-dsp-hip.so" if n == 0 else f"libcoolmic-dsp-hip-abl{n}.so")
    env = dict(os.environ, COOLMIC_HIP_LIB=lib)
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=120)
    ms = [l for l in out.stdout.splitlines() if l.startswith("MS")]
    name = " + ".join(v for k, v in BITS.items() if n & k) or "full pipeline"
    print(f"{n:3d}: {ms[0] if ms else out.stderr[-300:]}   {name}", flush=True)
