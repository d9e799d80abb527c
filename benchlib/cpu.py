"""cpu_baseline: the CPU oracle (test infrastructure, here as the thing timed BESIDE the GPU path -- never the
product) on the box's host cores, on a bounded sample of the benchmarked workload."""
import ctypes as C
import math
import os


def cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max), or None when unlimited / unknown"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else round(int(q) / int(per), 2)
    except Exception:
        return None


def cpu_baseline(workload, channels, gains, swap):
    """The CPU oracle (kind "port": scalar restatement of src/transform.c:101-124 and src/vumeter.c:161-218) on
    this host, on a bounded sample of the same workload.  Threads: one per CPU the process may actually use --
    min(visible hardware threads, ceil(cgroup CPU-time quota)): round 3 started one per VISIBLE thread (256 under
    a quota of 16 CPUs) and read the baseline a third low.  `value` is that run; the oversubscribed figure stays
    beside it (`all_visible_threads_Msamples_s`)."""
    from oracle import oracle_ffi
    lib = oracle_ffi.load()
    visible = os.cpu_count() or 1
    try:
        visible = len(os.sched_getaffinity(0))
    except Exception:
        pass
    quota = cpu_quota()                  # a container's CPU-time quota may be far below the threads it can see
    cores = visible if quota is None else max(1, min(visible, int(math.ceil(quota))))
    frames = 65536
    gain = (C.c_uint16 * channels)(*((gains * channels)[:channels] if gains else [0] * channels))
    scale = 1000 if gains else 0         # (scale 0: the reference's disabled gain)
    cmap_p = None
    if swap and channels == 2:
        cmap = (C.c_uint8 * 2)(1, 0)
        cmap_p = C.cast(cmap, C.c_void_p)
    chk = C.c_uint64()

    def run(threads, per_thread):
        streams = threads * per_thread
        secs = lib.oracle_bench_block(threads, streams, channels, frames, cmap_p, scale, gain, 12345, C.byref(chk))
        return streams, streams * frames * channels / secs / 1e6

    # ~12-15 s of CPU work in all (the timed arithmetic plus the untimed generation of every stream), spread over
    # the threads: about a second of wall time on a 16-CPU quota
    per_thread = max(48, int(2.4e9 / (cores * frames * channels)))
    streams, rate = run(cores, per_thread)
    _, rate_one = run(1, max(48, per_thread // 8))
    res = {
        "value": round(rate, 1), "unit": "Msamples/s", "cores": cores, "cpu_quota_cpus": quota,
        "visible_hardware_threads": visible, "kind": "port",
        "sample": "%d of the workload's streams (%d per thread) x %d frames x %d ch, same generator and "
                  "parameters, block-at-once, one thread per usable CPU" % (streams, per_thread, frames, channels),
        "one_thread_Msamples_s": round(rate_one, 1),
    }
    if visible > cores:
        _, rate_all = run(visible, 48)
        res["all_visible_threads_Msamples_s"] = round(rate_all, 1)
        res["all_visible_threads_note"] = "%d threads under the quota of %s CPUs (round 3's `value`)" % (visible, quota)
    chain_frames = 20_000_000
    secs_chain = lib.oracle_bench_chain(chain_frames, 1000, 900, C.byref(chk))
    res["pull_chain_1024B_one_thread_Msamples_s"] = round(chain_frames / secs_chain / 1e6, 1)
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                res["cpu_model"] = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    return res
