#!/usr/bin/env python3
"""coolmic_group_t: N pipelines behind the operator API, one upload / launch / download per block.
Host-side cost per block (sine sources through coolmic_iohandle_t, queues, readers)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
cm = ge.load_package()
for N, block in ((256, 4096), (1024, 4096), (4096, 1024)):
    C = 2
    grp = cm.Group(C, N, block, queue_blocks=2)
    hs = []
    for i in range(N):
        dev = cm.Snddev("null", 48000, C)
        h = dev.get_iohandle()
        slot = grp.add_stream(h)
        h.unref(); dev.unref()
        grp.set_master_gain(slot, C, 1000, [900, 1100])
        hs.append(grp.get_iohandle(slot))
    nbytes = block * 2 * C
    for _ in range(2):                         # warm up: pump + drain
        grp.pump()
        for h in hs:
            h.read(nbytes)
    t_pump = t_read = 0.0
    rounds = 8
    for _ in range(rounds):
        t0 = time.perf_counter(); grp.pump(); t1 = time.perf_counter()
        for h in hs:
            n, _d = h.read(nbytes)
        t2 = time.perf_counter()
        t_pump += t1 - t0; t_read += t2 - t1
    samples = rounds * N * block * C
    print(f"N={N:5d} block={block:5d}: pump {t_pump / rounds * 1e3:7.2f} ms, readers {t_read / rounds * 1e3:7.2f} ms per block  "
          f"-> {samples / (t_pump + t_read) / 1e6:8.1f} Msamples/s ({samples / t_pump / 1e6:8.1f} pump only)")
    for h in hs:
        h.unref()
    grp.unref()
