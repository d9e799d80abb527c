#!/usr/bin/env python3
"""Random parity sweep of the EQ path against the oracle: 0..4 sections, 1..16 channels, maps,
gains, ragged lengths, every output set, state and VU windows carried over several launches.
Usage: python tests/fuzz_eq.py [cases] [seed]"""
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
bad = 0
for case in range(cases):
    C = int(rng.choice([1, 1, 2, 2, 3, 5, 6, 8, 16]))
    S = int(rng.integers(1, 40))
    T = int(rng.choice([1, 63, 64, 65, 300, 1000]))
    nsec = int(rng.integers(0, 5))
    flags = cm.EQ | int(rng.choice([cm.OUT_F32, cm.OUT_PCM | cm.VU, cm.OUT_PCM | cm.VU | cm.INPLACE,
                                    cm.OUT_F32 | cm.OUT_PCM | cm.VU, cm.OUT_PCM, cm.VU | cm.OUT_F32]))
    coef = np.concatenate([cm.eq3(48000.0), cm.design_biquad(1, 48000.0, 3000.0, 4.0, 2.0)])[: 5 * nsec]
    b = cm.Batch(S, C, T, flags=flags)
    assert b.set_eq(-1, coef if nsec else None) == 0
    gas, maps = [], []
    for s in range(S):
        ga = None if rng.random() < 0.3 else (int(rng.integers(1, 4000)), [int(v) for v in rng.integers(0, 5000, C)])
        m = None if rng.random() < 0.5 else [int(v) for v in rng.integers(0, C, C)]
        if ga:
            assert b.set_gain(s, C, ga[0], ga[1]) == 0
        if m:
            assert b.set_chmap(s, m) == 0
        gas.append(ga)
        maps.append(m)
    q = (of.Biquad * max(nsec, 1))()
    for i in range(nsec):
        q[i].b0, q[i].b1, q[i].b2, q[i].a1, q[i].a2 = [float(v) for v in coef[5 * i:5 * i + 5]]
    states = [[np.zeros(4 * max(nsec, 1), dtype=np.float32) for _ in range(C)] for _ in range(S)]
    vus = [oracle.vu_new(C) for _ in range(S)]
    ok = True
    for k in range(int(rng.integers(1, 4))):
        lens = [int(rng.integers(0, T + 1)) if rng.random() < 0.6 else T for _ in range(S)]
        xs = [rng.integers(-32768, 32768, lens[s] * C).astype(np.int16) for s in range(S)]
        for s in range(S):
            if lens[s]:
                b.upload(s, xs[s])
        b.run(T, frames_per_stream=lens)
        for s in range(S):
            n = lens[s]
            want_i = np.empty((n, C), dtype=np.int16)
            want_f = []
            for c in range(C):
                src = c if maps[s] is None else maps[s][c]
                g = None
                if gas[s] is not None:
                    rc, g = oracle.gain(1, 1, gas[s][0], [gas[s][1][c]])
                    assert rc == 0
                wf, wi = oracle.eq_run_mono(g, q, nsec, states[s][c], xs[s].reshape(-1, C)[:, src].copy())
                want_i[:, c] = wi
                want_f.append(wf)
            want_i = want_i.reshape(-1)
            oracle.vu_accumulate(vus[s], want_i)
            if flags & cm.OUT_PCM:
                got = b.download(s, n) if n else np.zeros(0, np.int16)
                ok &= np.array_equal(got, want_i)
            if flags & cm.OUT_F32 and n:
                for c in range(C):
                    ok &= np.array_equal(b.download_f32(s, c, n).view(np.uint32), want_f[c].view(np.uint32))
    if flags & cm.VU:
        for s in range(S):
            rc_o, r_o = oracle.vu_result(vus[s])
            rc_g, r_g = b.vu_result(s)
            ok &= rc_g == rc_o and (rc_o != 0 or r_g.as_dict() == of.vu_result_dict(r_o))
    b.close()
    if not ok:
        bad += 1
        print(f"MISMATCH case {case}: C={C} S={S} T={T} nsec={nsec} flags={flags:#x}", flush=True)
print(f"{cases} EQ cases, seed {seed}: {bad} mismatches")
sys.exit(1 if bad else 0)
