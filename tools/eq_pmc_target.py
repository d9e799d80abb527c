#!/usr/bin/env python3
"""Five launches of the EQ kernel (config 3, or the shape named on the command line: eq3vu1, eq3vu, eq3f6,
eq3vu6 as in tools/eq_stamps.py): the program rocprofv3 --pmc passes profile (`tools/eq_pmc.sh [SHAPE]`).  Put
`python3 tools/eq_pmc_target.py [SHAPE]` directly after `--`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

cm = ge.load_package()
SHAPES = {"eq3": (8192, 1, cm.OUT_F32), "eq3vu1": (8192, 1, cm.OUT_PCM | cm.VU), "eq3vu": (4096, 2, cm.OUT_PCM | cm.VU),
          "eq3f6": (1365, 6, cm.OUT_F32), "eq3vu6": (1365, 6, cm.OUT_PCM | cm.VU),
          "eq3f4": (2048, 4, cm.OUT_F32), "eq3f3": (2730, 3, cm.OUT_F32), "eq3f16": (512, 16, cm.OUT_F32)}
S, Cn, out_flags = SHAPES[sys.argv[1] if len(sys.argv) > 1 else "eq3"]
T = 65536
b = cm.Batch(S, Cn, T, flags=cm.EQ | out_flags)
b.set_eq(-1, cm.eq3())
b.set_gain(-1, 1, 1000, [900])
b.generate(cm.GEN_NOISE, 12345, T)
for _ in range(5):
    b.run(T)
b.sync()
b.close()
