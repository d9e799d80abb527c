#!/usr/bin/env python3
"""EQ kernel by channel count (the CH = 0 form of k_eq_pipe: T-in staging through LDS, staged int16 result), three
sections, ~0.5 G samples per launch: float planes / int16 + VU / all three outputs.  With library paths on the
command line: every build on the same shapes, interleaved (as tools/ab_two_libs.py).
usage: bench_eq_channels.py [lib.so ...]"""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "libcoolmic-dsp_amd")


def load(path, tag):
    if path:
        os.environ["COOLMIC_HIP_LIB"] = os.path.abspath(path)
    spec = importlib.util.spec_from_file_location("cm_" + tag, os.path.join(PKG, "__init__.py"), submodule_search_locations=[PKG])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["cm_" + tag] = mod
    spec.loader.exec_module(mod)
    return mod


libs = sys.argv[1:] or [None]
mods = [(os.path.basename(p) if p else "product", load(p, str(i))) for i, p in enumerate(libs)]
T = 65536
PAD = int(os.environ.get("EQCH_PAD", "0"))          # slot capacity T + PAD frames: other strides between slots and planes
for C in [int(c) for c in os.environ.get("EQCH_C", "3,4,5,6,8,12,16").split(",")]:
    S = 8192 // C
    for name, fl, bps in (("float planes", "OUT_F32", 6), ("int16 + VU", "OUT_PCM|VU", 4), ("float + int16 + VU", "OUT_F32|OUT_PCM|VU", 8)):
        row = []
        for tag, cm in mods:
            flags = cm.EQ
            for f in fl.split("|"):
                flags |= getattr(cm, f)
            b = cm.Batch(S, C, T + PAD, flags=flags)
            b.set_eq(-1, cm.eq3())
            b.set_gain(-1, 1, 1000, [900])
            b.generate(cm.GEN_NOISE, 12345, T)
            for _ in range(60):
                b.run(T)
            b.sync()
            b.timing(True); b.timing_read()
            for _ in range(40):
                b.run(T)
            ms, n = b.timing_read()
            b.close()
            row.append((tag, ms / n))
        print("C=%2d %-20s " % (C, name) + "   ".join("%s %.4f ms %6.1f GB/s" % (t, m, S * C * T * bps / m / 1e6) for t, m in row), flush=True)
