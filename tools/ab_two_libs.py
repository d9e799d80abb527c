#!/usr/bin/env python3
"""A/B of several builds of the library in ONE process, interleaved, at sustained clocks: every library
is loaded under a module name of its own, each gets its own batch of the same shape, and timed
blocks of launches alternate between them (process-to-process differences -- placement of the
arrays, clocks -- cancel).  usage: ab_two_libs.py SHAPE lib1.so lib2.so ...   SHAPE: eq3 | eq3vu | eq3vu1 | eq3all | eq3vu6 | c2 | c2s | c4 | vu1 | vu2 | vu6"""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "libcoolmic-dsp_amd")


def load(lib_path, tag):
    os.environ["COOLMIC_HIP_LIB"] = lib_path
    name = "cm_" + tag
    spec = importlib.util.spec_from_file_location(name, os.path.join(PKG, "__init__.py"),
                                                  submodule_search_locations=[PKG])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def make(cm, shape):
    T = 65536
    if shape in ("eq3", "eq3vu", "eq3vu1", "eq3all", "eq3vu6", "eq3f2"):
        S, C = {"eq3": (8192, 1), "eq3vu": (4096, 2), "eq3vu1": (8192, 1), "eq3all": (4096, 2), "eq3vu6": (1365, 6), "eq3f2": (4096, 2)}[shape]
        flags = cm.EQ | {"eq3": cm.OUT_F32, "eq3f2": cm.OUT_F32, "eq3all": cm.OUT_F32 | cm.OUT_PCM | cm.VU}.get(shape, cm.OUT_PCM | cm.VU)
        b = cm.Batch(S, C, T, flags=flags)
        b.set_eq(-1, cm.eq3())
        b.set_gain(-1, 1, 1000, [900])
        bps = {"eq3": 6, "eq3f2": 6, "eq3all": 8}.get(shape, 4)
    elif shape in ("c4", "vu2", "vu1", "c2s"):       # mono PCM + VU; stereo / mono VU only; config 2 at 4096 frames
        S, C, T = {"c4": (8192, 1, T), "vu2": (4096, 2, T), "vu1": (8192, 1, T), "c2s": (4096, 2, 4096)}[shape]
        pcm = shape in ("c4", "c2s")
        b = cm.Batch(S, C, T, flags=(cm.OUT_PCM | cm.VU) if pcm else cm.VU)
        b.set_gain(-1, C, 1000, [750, 1250][:C])
        bps = 4 if pcm else 2
    elif shape == "c2":
        S, C, bps = 4096, 2, 4
        b = cm.Batch(S, C, T, flags=cm.OUT_PCM | cm.VU)
        b.set_gain(-1, 2, 1000, [750, 1250])
        b.set_chmap(-1, [1, 0])
    else:
        S, C, T, bps = 2730, 6, 16384, 2
        b = cm.Batch(S, C, T, flags=cm.VU)
        b.set_gain(-1, 1, 1000, [900])
    b.generate(cm.GEN_NOISE, 12345, T)
    return b, T, S * C * T * bps


shape = sys.argv[1]
libs = sys.argv[2:]
items = []
for i, spec in enumerate(libs):            # "lib.so" or "lib.so+NAME=VALUE" (environment while its batch is made)
    path, _, env = spec.partition("+")
    cm = load(os.path.abspath(path), str(i))
    if env:
        k, _, v = env.partition("=")
        os.environ[k] = v
    b, T, nbytes = make(cm, shape)
    if env:
        os.environ.pop(k, None)
    items.append((os.path.basename(spec), b, T, nbytes))
for _, b, T, _ in items:
    for _ in range(80):
        b.run(T)
    b.sync()
res = {name: [] for name, *_ in items}
for rnd in range(5):
    for name, b, T, _ in items:
        b.timing(True)
        b.timing_read()
        for _ in range(40):
            b.run(T)
        ms, n = b.timing_read()
        b.timing(False)
        if shape in ("c2", "vu6"):
            b.vu_reset(-1)
        res[name].append(ms / n)
base = None
for name, b, T, nbytes in items:
    v = sorted(res[name])
    med = v[len(v) // 2]
    base = base or med
    print(f"{shape} {name:40s} median {med:.4f} ms  min {v[0]:.4f}  max {v[-1]:.4f}  {nbytes / med / 1e6:7.1f} GB/s  "
          f"{(med / base - 1) * 100:+5.1f} % vs first", flush=True)
