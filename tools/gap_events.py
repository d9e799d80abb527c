"""Per-step host gap with and without timing events / window snapshots (config 2 shape; needs a GPU).
Run as a script: python tools/gap_events.py"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.getcwd())


def main():
    import __graft_entry__ as ge
    cm = ge.load_package()
    S, Cn, T = 4096, 2, 65536
    b = cm.Batch(S, Cn, T, flags=cm.OUT_PCM | cm.VU)
    b.set_gain(-1, 2, 1000, [750, 1250]); b.set_chmap(-1, [1, 0])
    b.generate(cm.GEN_NOISE, 12345, T); b.sync()
    res = (cm.VuResult * S)(); rcs = (C.c_int * S)()
    def loop(n, timing, snap):
        b.timing(timing); b.timing_read()
        pending = False
        b.sync(); t0 = time.perf_counter()
        for _ in range(n):
            b.run(T)
            if snap:
                b.vu_snapshot()
                if pending: b.vu_collect(res, rcs)
                pending = True
        if pending: b.vu_collect(res, rcs)
        b.sync(); dt = time.perf_counter() - t0
        ms, k = b.timing_read() if timing else (0, 1)
        b.vu_reset(-1)
        return dt / n * 1e3, ms / max(k, 1)
    loop(5, True, True)
    for timing in (True, False):
        for snap in (True, False):
            r = [loop(30, timing, snap) for _ in range(3)]
            print("timing", timing, "snapshots", snap, "ms/step", [round(x[0], 4) for x in r], "kernel", [round(x[1], 4) for x in r])


if __name__ == "__main__":
    main()
