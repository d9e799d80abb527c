#!/usr/bin/env python3
"""Five launches of the config-3 EQ kernel: the program rocprofv3 --pmc passes profile
(`tools/eq_pmc.sh`).  Put `python3 tools/eq_pmc_target.py` directly after `--`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

cm = ge.load_package()
S, T = 8192, 65536
b = cm.Batch(S, 1, T, flags=cm.EQ | cm.OUT_F32)
b.set_eq(-1, cm.eq3())
b.set_gain(-1, 1, 1000, [900])
b.generate(cm.GEN_NOISE, 12345, T)
for _ in range(5):
    b.run(T)
b.sync()
b.close()
