#!/usr/bin/env python3
"""Where the host time of a bench step goes: per-call wall time of run / vu_snapshot / vu_collect,
with and without torch imported first (torch brings its own HIP runtime)."""
import os
import sys
import time
import ctypes as C

if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch  # noqa: F401
    torch.cuda.set_device(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

cm = ge.load_package()
S, Cn, T = 4096, 2, 65536
b = cm.Batch(S, Cn, T, flags=cm.OUT_PCM | cm.VU)
b.set_gain(-1, 2, 1000, [750, 1250]); b.set_chmap(-1, [1, 0])
b.generate(cm.GEN_NOISE, 12345, T); b.sync()
res = (cm.VuResult * S)(); rcs = (C.c_int * S)()
b.timing(True)
acc = {"run": 0.0, "snapshot": 0.0, "collect": 0.0}
def step(pending):
    t0 = time.perf_counter(); b.run(T)
    t1 = time.perf_counter(); b.vu_snapshot()
    t2 = time.perf_counter()
    if pending:
        b.vu_collect(res, rcs)
    t3 = time.perf_counter()
    acc["run"] += t1 - t0; acc["snapshot"] += t2 - t1; acc["collect"] += t3 - t2
for _ in range(5):
    step(True)
b.sync()
if "sync" in sys.argv:
    torch.cuda.synchronize()
if "alloc" in sys.argv:
    _t = torch.zeros(16, device="cuda")
if "reads" in sys.argv:
    b.timing_read()
for k in acc: acc[k] = 0.0
n = 40
t0 = time.perf_counter()
for i in range(n):
    step(True)
b.sync()
dt = time.perf_counter() - t0
ms, k = b.timing_read()
print(("torch first" if "torch" in sys.argv else "no torch   "), f"{dt / n * 1e3:.4f} ms/step, kernel {ms / k:.4f};",
      ", ".join(f"{k} {v / n * 1e6:.0f} us" for k, v in acc.items()))
