#!/usr/bin/env python3
"""Random parity sweep of the block kernels against the oracle (beyond the fixed seeds of tests/):
channel counts 1..16, maps, gains, ragged lengths, every output set, windows over several launches.
Usage: python tests/fuzz_parity.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
from oracle import oracle_ffi as of

cm = ge.load_package()
oracle = of.Oracle()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)


def expect(x, C, ga, cmap):
    y = x
    if cmap is not None:
        y = oracle.chmap(cmap, y, C)
    if ga is not None:
        rc, g = oracle.gain(C, *ga)
        assert rc == 0
        y = oracle.gain_apply(g, y, C)
    return y


bad = 0
for case in range(cases):
    C = int(rng.integers(1, 17))
    S = int(rng.integers(1, 12))
    T = int(rng.choice([1, 7, 64, 65, 513, 1000, 4097, 20000]))
    flags = int(rng.choice([cm.OUT_PCM | cm.VU, cm.OUT_PCM | cm.VU | cm.INPLACE, cm.VU, cm.OUT_PCM,
                            cm.OUT_F32 | cm.OUT_PCM | cm.VU, cm.OUT_F32, cm.OUT_F32 | cm.VU]))
    if rng.random() < 0.2:
        flags |= cm.HOSTPCM
    any_map = rng.random() < 0.5
    b = cm.Batch(S, C, T, flags=flags)
    gas, maps = [], []
    whole = rng.random()                     # some batches: no gain anywhere / every gain below its scale
    for s in range(S):
        mode = int(rng.integers(0, 4))
        ga = None if mode == 0 else (C, int(rng.integers(1, 65536)), [int(v) for v in rng.integers(0, 65536, C)]) \
            if mode < 3 else (1, int(rng.choice([1, 1000, 65535])), [int(rng.integers(0, 65536))])
        if whole < 0.15:
            ga = None if rng.random() < 0.7 else (C, 4242, [4242] * C)              # disabled or unity
        elif whole < 0.3:
            sc = int(rng.integers(2, 65536))
            ga = (C, sc, [int(v) for v in rng.integers(0, sc, C)])                  # all below the scale
        m = [int(v) for v in rng.integers(0, C, C)] if any_map and rng.random() < 0.7 else None
        if ga:
            assert b.set_gain(s, *ga) == 0
        if m:
            assert b.set_chmap(s, m) == 0
        gas.append(ga)
        maps.append(m)
    wants = [[] for _ in range(S)]
    launches = int(rng.integers(1, 4))
    ok = True
    for k in range(launches):
        lens = [int(rng.integers(0, T + 1)) if rng.random() < 0.7 else T for _ in range(S)]
        xs = []
        for s in range(S):
            kind = rng.random()
            x = rng.integers(-32768, 32768, lens[s] * C).astype(np.int16)
            if kind < 0.2:
                x = rng.choice(np.array([-32768, -32767, -1, 0, 1, 32767], dtype=np.int16), lens[s] * C)
            xs.append(x)
            if lens[s]:
                b.upload(s, x)
        b.run(T, frames_per_stream=lens)
        for s in range(S):
            want = expect(xs[s], C, gas[s], maps[s])
            wants[s].append(want)
            if flags & cm.OUT_PCM:
                got = b.download(s, lens[s]) if lens[s] else np.zeros(0, np.int16)
                ok &= np.array_equal(got, want)
            if flags & cm.OUT_F32 and lens[s]:
                planar = oracle.to_f32_planar(want, C)
                for c in range(C):
                    ok &= np.array_equal(b.download_f32(s, c, lens[s]).view(np.uint32), planar[c].view(np.uint32))
    if flags & cm.VU:
        for s in range(S):
            v = oracle.vu_new(C)
            for w in wants[s]:
                oracle.vu_accumulate(v, w)
            rc_o, r_o = oracle.vu_result(v)
            rc_g, r_g = b.vu_result(s)
            ok &= rc_g == rc_o and (rc_o != 0 or r_g.as_dict() == of.vu_result_dict(r_o))
    b.close()
    if not ok:
        bad += 1
        print(f"MISMATCH case {case}: C={C} S={S} T={T} flags={flags:#x} maps={any_map} launches={launches}", flush=True)
print(f"{cases} cases, seed {seed}: {bad} mismatches")
sys.exit(1 if bad else 0)
