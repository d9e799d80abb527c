import os, sys
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
cm = ge.load_package()
T = 16384
for C in (3, 6, 8, 16):
    S = (1 << 28) // (T * C * 2)
    for name, gain in (("gains", True), ("no gain", False)):
        b = cm.Batch(S, C, T, flags=cm.VU)
        if gain:
            b.set_gain(-1, 1, 1000, [900])
        b.generate(cm.GEN_NOISE, 5, T)
        for _ in range(3): b.run(T)
        b.sync(); b.timing(True); b.timing_read()
        for _ in range(20): b.run(T)
        ms, n = b.timing_read()
        print(f"C={C:2d} VU only, {name:8s}: {ms/n:.4f} ms  {S*C*T*2/(ms/n*1e-3)/1e9:6.0f} GB/s")
        b.close()
