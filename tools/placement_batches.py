#!/usr/bin/env python3
"""Is the 6 % spread of the config-2 kernel between processes a property of where its arrays lie?
Several batches of the config-2 shape in ONE process, timed in interleaved blocks of launches (kernel time
from the dispatch's own events).  usage: placement_batches.py [batches] [rounds]"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cm = importlib.import_module("libcoolmic-dsp_amd")

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
S, C, T = 4096, 2, 65536
import time
batches = []
for i in range(NB):
    t0 = time.time()
    b = cm.Batch(S, C, T, flags=cm.OUT_PCM | cm.VU)
    print("batch %d created in %.0f ms" % (i, (time.time() - t0) * 1e3), flush=True)
    b.set_gain(-1, 2, 1000, [750, 1250])
    b.set_chmap(-1, [1, 0])
    b.generate(cm.GEN_NOISE, 12345, T)
    batches.append(b)
for _ in range(400):                      # sustained clocks
    batches[0].run(T)
batches[0].sync()
for r in range(ROUNDS):
    line = []
    for b in batches:
        for _ in range(50):
            b.run(T)
        b.sync()
        b.timing(True)
        b.timing_read()
        for _ in range(200):
            b.run(T)
        ms, n = b.timing_read()
        b.timing(False)
        line.append(ms / n)
    print("round %d: " % r + "  ".join("%.4f" % v for v in line), flush=True)
