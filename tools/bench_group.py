#!/usr/bin/env python3
"""coolmic_group_t: N pipelines behind the operator API, one upload / launch / download per block.
Host-side cost per block (null sources through coolmic_iohandle_t, pinned staging, queues): `rounds`
pumps back to back -- each overlaps its pull with the previous block's upload, kernel and download --
then the readers drain the queues (through ctypes, so their time says nothing about a C host)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
cm = ge.load_package()
rounds = 8
THREADS = [int(t) for t in os.environ.get("GROUP_PULL_THREADS", "1,4,8").split(",")]
for N, block, threads in [(n, b, t) for (n, b) in ((256, 4096), (1024, 4096), (4096, 1024), (4096, 4096)) for t in THREADS]:
    C = 2
    grp = cm.Group(C, N, block, queue_blocks=rounds + 2)
    grp.set_pull_threads(threads)
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
    t0 = time.perf_counter()
    for _ in range(rounds):
        grp.pump()
    n, _d = hs[0].read(nbytes)                 # brings the last block home
    t1 = time.perf_counter()
    assert n == nbytes
    got = nbytes
    for i, h in enumerate(hs):
        for r in range(rounds if i else rounds - 1):
            n, _d = h.read(nbytes)
            got += n
    t2 = time.perf_counter()
    assert got == rounds * N * nbytes, (got, rounds * N * nbytes)
    samples = rounds * N * block * C
    print(f"N={N:5d} block={block:5d} pull threads={threads}: pump {(t1 - t0) / rounds * 1e3:7.3f} ms per block -> "
          f"{samples / (t1 - t0) / 1e6:8.1f} Msamples/s;  readers (ctypes) {(t2 - t1) / rounds * 1e3:7.2f} ms per block")
    for h in hs:
        h.unref()
    grp.unref()
