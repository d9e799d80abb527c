#!/usr/bin/env python3
"""Kernel time of the EQ path (config 3 shape) by output set and gain setting."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

cm = ge.load_package()
T = 65536
for C in (1, 2, 6):
  S = 8192 // C
  for name, flags, bps in (("float planes", cm.EQ | cm.OUT_F32, 6),
                           ("int16 + VU", cm.EQ | cm.OUT_PCM | cm.VU, 4),
                           ("int16 in place + VU", cm.EQ | cm.OUT_PCM | cm.VU | cm.INPLACE, 4),
                           ("float + int16 + VU", cm.EQ | cm.OUT_F32 | cm.OUT_PCM | cm.VU, 8)):
      for gain in (True, False):
          b = cm.Batch(S, C, T, flags=flags)
          b.set_eq(-1, cm.eq3())
          if gain:
              b.set_gain(-1, 1, 1000, [900])
          b.generate(cm.GEN_NOISE, 12345, T)
          for _ in range(60):       # warm: the clocks ramp over the first ~100 ms
              b.run(T)
          b.sync()
          b.timing(True)
          for _ in range(60):
              b.run(T)
          b.sync()
          ms, n = b.timing_read()
          ms /= n
          print(f"C={C} {name:24s} gain {'on ' if gain else 'off'}: {ms:.4f} ms  {S * C * T * bps / ms / 1e6:7.1f} GB/s "
                f"({bps} B/sample)  {S * C * T / ms / 1e3:9.1f} Msamples/s", flush=True)
          b.close()
