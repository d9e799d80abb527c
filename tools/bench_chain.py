#!/usr/bin/env python3
"""What ONE pipeline behind the reference's operator API pays per 1024-byte pull (launch + wait), in the
two wirings that exist:

  direct   sine -> transform -> vumeter            (BASELINE config 1; the meter shares the transform's launch)
  tee      sine -> transform -> tee -> {reader pulling 1024 bytes, vumeter}
                                                   (the product's wiring, ref: src/simple.c:212-229; the meter
                                                    shares the launch through window records)

A result() every 20 reads, as the product takes them (ref: src/simple.c:370).  Usage: bench_chain.py [--tee]
(without the flag both wirings run, and the ratio is printed)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

cm = ge.load_package()
N, WARM = 4000, 200


def chain(tee, gain):
    dev = cm.Snddev("sine", 48000, 1)
    tr = cm.Transform(48000, 1)
    h0 = dev.get_iohandle()
    tr.attach(h0); h0.unref(); dev.unref()
    h = tr.get_iohandle()
    vu = cm.Vumeter(48000, 1)
    enc = t = None
    if tee:
        t = cm.Tee(2)
        t.attach(h); h.unref()
        enc = t.get_iohandle(0)
        h = t.get_iohandle(1)
    vu.attach(h)
    if gain:
        tr.set_master_gain(*gain)

    def pull(i):
        if enc is not None:
            enc.read(1024)
        vu.read(-1)
        if i % 20 == 19:
            vu.result()

    for i in range(WARM):
        pull(i)
    runs0 = cm.lib.cmhip_debug_run_count()
    t0 = time.perf_counter()
    for i in range(N):
        pull(i)
    dt = time.perf_counter() - t0
    runs = cm.lib.cmhip_debug_run_count() - runs0
    mode = vu.mode()
    for o in (h, vu, tr) + ((enc, t) if tee else ()):
        o.unref()
    return dt / N * 1e6, runs / N, mode


only_tee = "--tee" in sys.argv
out = {}
for gain in (None, (1, 1000, [900])):
    for tee in ((True,) if only_tee else (False, True)):
        us, runs, mode = chain(tee, gain)
        out[(tee, bool(gain))] = us
        print(f"{'tee   ' if tee else 'direct'}  gain {'on ' if gain else 'off'}: {us:7.2f} us per 1024-byte pull, "
              f"{512 / us:6.2f} Msamples/s, {runs:.2f} launches per pull (meter mode {mode})")
if not only_tee:
    for g in (False, True):
        print(f"tee / direct, gain {'on ' if g else 'off'}: {out[(True, g)] / out[(False, g)]:.3f}")
