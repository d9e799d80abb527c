#!/usr/bin/env python3
"""bench.py -- Msamples/s through transform -> vumeter on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload c2|c4|c3|c5]

A step is one pass of the hot path over one batch that is already resident in HBM:
one fused gain(+channel map) -> VU launch over every stream of the rank's shard, the
asynchronous snapshot of all VU windows to the host, and the host-side dB finish
(double, as the reference) of the previous step's windows, overlapped with the GPU.

Workloads (per GPU; stream s of the node lives on rank s % N -- round-robin shards,
no data-path collective, "weak" scaling):
  c2  4096 stereo streams x 65536 frames, channel swap + gains {750,1250}/1000, PCM
      materialised (2 B read + 2 B written per sample)             [default, configs[1]]
  c4  8192 mono streams x 65536 frames, gain 900/1000, PCM materialised  [configs[3]]
  c5  c4 + node-global VU: one RCCL all-reduce pair per step             [configs[4]]
  c3  8192 mono streams, int16 -> float + 3-band EQ, float out           [configs[2]]

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel, timed with HIP
events on the stream it is launched on; `cpu_baseline` is the CPU oracle (the scalar
restatement of the reference loops) timed on this host, N=1 only.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

WORKLOADS = {
    # name: (streams/GPU, channels, frames, bytes per sample, description)
    "c2": (4096, 2, 65536, 4, "4096 stereo int16 48 kHz streams x 65536 frames per GPU, "
                              "channel swap + gains {750,1250}/1000 -> VU, PCM materialised"),
    "c4": (8192, 1, 65536, 4, "8192 mono int16 streams x 65536 frames per GPU (65536 streams "
                              "round-robin over 8 GPUs), gain 900/1000 -> VU, PCM materialised"),
    "c5": (8192, 1, 65536, 4, "c4 + node-global VU via RCCL all-reduce each step"),
    "c3": (8192, 1, 65536, 6, "8192 mono streams x 65536 frames per GPU, int16 -> float + "
                              "3-band biquad EQ, planar float out"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 0.1 s of warm-up and 0.35 s of timed steps -- the chip needs ~100 ms of load to reach the
    # clocks it then holds (config 2: 0.355 ms per step over 100 steps from idle, 0.349 over 1000)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--frames", type=int, default=0, help="override frames per launch")
    ap.add_argument("--streams", type=int, default=0, help="override streams per GPU")
    ap.add_argument("--strong", action="store_true",
                    help="fixed total: the workload's streams are divided among the ranks (default: "
                         "weak scaling, the workload's streams per GPU)")
    ap.add_argument("--node-batch", type=int, default=8,
                    help="config 5: blocks whose node-global VU records travel in one all-gather")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extras", action="store_true", help="skip ceilings / VU-only line")
    args = ap.parse_args()

    # stdout carries exactly one line, rank 0's JSON: whatever libraries print on the way (gloo
    # announces its connections on stdout) goes to stderr with everything else
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py: --gpus %d needs one process per GPU: launch with "
                     "python -m torch.distributed.run --nproc-per-node %d ..." % (args.gpus, args.gpus))
        args.gpus = world

    # torch is plumbing for the multi-rank run only (rendezvous, the barrier and the clock over
    # ranks; RCCL for the one workload with an exchange step, config 5).  It brings its own HIP
    # runtime (ROCm 7.0 inside the wheel), and under that runtime's default direct dispatch the
    # submitting thread of this step loop was seen stalling for a kernel's length per step on
    # busy hosts (0.7 ms per step instead of 0.37).  So:
    #   * one rank: torch is not loaded at all;
    #   * several ranks without a data-path collective (c2, c3, c4): the engine is loaded FIRST
    #     and keeps the system HIP runtime, torch comes second and only runs gloo on the CPU for
    #     the barrier and the max-over-ranks clock -- torch.cuda is never touched;
    #   * config 5 (node-global VU over RCCL): torch first, with AMD_DIRECT_DISPATCH=0 unless the
    #     environment says otherwise (0.40 ms per step).
    force_node = os.environ.get("COOLMIC_BENCH_FORCE_NODE") == "1"       # single-rank test of the reduce path
    # Rehearsal knob for a 1-GPU box (never set by the driver): all ranks share device 0 and any
    # GPU collective is replaced by gloo on host copies, so the N>1 code can run without N GPUs.
    rehearsal = os.environ.get("COOLMIC_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    gpu_torch = args.workload == "c5" and (world > 1 or force_node)    # torch.cuda is needed
    need_torch = world > 1 or force_node
    torch = dist = None
    if gpu_torch:
        os.environ.setdefault("AMD_DIRECT_DISPATCH", "0")
        import torch
        import torch.distributed as dist

    import __graft_entry__ as ge
    cm = ge.load_package()
    from libcoolmic_dsp_amd import shard

    if cm.device_count() < 1:
        sys.exit("bench.py: no HIP device; this path has no CPU fallback")
    if need_torch and not gpu_torch:
        import torch
        import torch.distributed as dist
    if need_torch:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29599")
        if gpu_torch and not rehearsal:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            if gpu_torch:
                torch.cuda.set_device(local_rank)
            dist.init_process_group("gloo", rank=rank, world_size=world)
    coll_device = "cuda" if gpu_torch and not rehearsal else "cpu"

    S, Cn, T, bps, desc = WORKLOADS[args.workload]
    if args.frames:
        T = args.frames
    if args.streams:
        S = args.streams
    if args.strong:
        if S % world:
            raise SystemExit("--strong: %d streams do not divide among %d ranks" % (S, world))
        S //= world
    eq = args.workload == "c3"
    node_vu = args.workload == "c5"

    if eq:
        flags = cm.EQ | cm.OUT_F32
    else:
        flags = cm.OUT_PCM | cm.VU
    b = cm.Batch(S, Cn, T, flags=flags, device=local_rank)
    if args.workload == "c2":
        assert b.set_gain(-1, 2, 1000, [750, 1250]) == 0
        assert b.set_chmap(-1, [1, 0]) == 0
    elif os.environ.get("COOLMIC_BENCH_GAIN", "1") != "0":
        assert b.set_gain(-1, 1, 1000, [900]) == 0
    if eq:
        assert b.set_eq(-1, cm.eq3(48000.0)) == 0
    # global stream id of local stream s is rank + s*world (round-robin sharding)
    n_local, first_global, global_step = shard.shard(S * world, world, rank)
    assert n_local == S
    b.generate(cm.GEN_NOISE, 12345, T, first_global=first_global, global_step=global_step)
    b.sync()

    has_vu = bool(flags & cm.VU)
    results = (cm.VuResult * S)()
    rcs = (C.c_int * S)()
    # node-global VU (config 5): every block leaves one 34-word record per rank; the records of
    # --node-batch blocks travel in ONE all-gather (the exchange is latency bound) and are combined
    # on the device.  Two sets of record buffers alternate so that a set's exchange runs beside the
    # next blocks' kernels; the batch's stream and torch's stream order themselves with events,
    # the host never waits inside the loop.
    node_on = node_vu and (world > 1 or force_node)
    NB = max(1, args.node_batch)
    node_sets = [torch.zeros(NB, cm.NODE_WORDS, dtype=torch.int64, device="cuda") for _ in range(2)] if node_on else None
    node_scratch = torch.empty(world, NB, cm.NODE_WORDS, dtype=torch.int64,
                               device="cpu" if rehearsal else "cuda") if node_on else None
    node_host = torch.zeros(NB, cm.NODE_WORDS, dtype=torch.int64) if node_on and rehearsal else None
    ext_stream = torch.cuda.ExternalStream(b.hip_stream()) if node_on and not rehearsal else None
    ev_ready = [torch.cuda.Event() for _ in range(2)] if ext_stream is not None else None
    ev_done = [torch.cuda.Event() for _ in range(2)] if ext_stream is not None else None
    node_step = [0]
    node_result = [None]                   # combined records of the last exchanged set

    def node_exchange(k):
        words = node_sets[k]
        if ext_stream is None:             # rehearsal (gloo): through a host copy
            b.sync()
            node_host.copy_(words)
            node_result[0] = shard.gather_node_records(dist, node_host, node_scratch)
        else:
            ev_ready[k].record(ext_stream)
            cur = torch.cuda.current_stream()
            cur.wait_event(ev_ready[k])                # RCCL runs after the records are written
            node_result[0] = shard.gather_node_records(dist, words, node_scratch)
            ev_done[k].record(cur)

    def device_sync():
        cm.device_synchronize(local_rank)       # hipDeviceSynchronize: every stream of this rank's GPU
        if gpu_torch:
            torch.cuda.synchronize()

    def barrier():
        device_sync()
        if world > 1:
            dist.barrier()
        device_sync()

    def run_steps(n):
        # order per step: launch, snapshot of this window (async), then the host finishes the
        # PREVIOUS window -- the next launch is always queued before the host blocks
        pending = False
        probe = os.environ.get("COOLMIC_BENCH_PROBE") == "1"
        acc = [0.0, 0.0, 0.0]
        for _ in range(n):
            tp0 = time.perf_counter()
            b.run(T)
            tp1 = time.perf_counter()
            if node_on:
                i = node_step[0]
                node_step[0] += 1
                k, slot = (i // NB) & 1, i % NB
                if slot == 0 and ext_stream is not None:
                    ext_stream.wait_event(ev_done[k])      # the exchange that last used this set
                b.node_partial(node_sets[k][slot].data_ptr(), first_global=rank, global_step=world)
                if slot == NB - 1:
                    node_exchange(k)
            if has_vu:
                b.vu_snapshot()                      # async D2H of all windows + reset
                tp2 = time.perf_counter()
                if pending:
                    b.vu_collect(results, rcs)       # dB finish of the previous window (host)
                pending = True
                if probe:
                    tp3 = time.perf_counter()
                    acc[0] += tp1 - tp0; acc[1] += tp2 - tp1; acc[2] += tp3 - tp2
        if probe and n:
            print("probe: per step run %.0f us, snapshot %.0f us, collect %.0f us" %
                  tuple(v / n * 1e6 for v in acc), file=sys.stderr)
        if pending:
            b.vu_collect(results, rcs)
        if node_on and node_step[0] % NB:              # records of a partly filled set
            node_exchange((node_step[0] // NB) & 1)
            node_step[0] += NB - node_step[0] % NB
        b.sync()
        if node_on and ext_stream is not None:
            torch.cuda.current_stream().synchronize()

    run_steps(args.warmup)
    b.timing(True)
    b.timing_read()
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    kern_ms, launches = b.timing_read()
    b.timing(False)

    if world > 1:
        dt = shard.max_over_ranks(dist, dt, device=coll_device)

    samples_per_step_rank = S * Cn * T
    total_samples = samples_per_step_rank * world * args.steps
    value = total_samples / dt / 1e6

    kern_avg_ms = kern_ms / max(launches, 1)
    achieved = samples_per_step_rank * bps / (kern_avg_ms * 1e-3) / 1e9 if kern_avg_ms > 0 else 0.0
    traffic = None
    # HBM bytes per launch from the PMC passes of tools/hbm_pmc.sh, kept per workload
    pmc_path = os.path.join(ROOT, "profiles", "pmc_%s.json" % args.workload)
    if not os.path.exists(pmc_path):
        pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc_path):
        try:
            pmc = json.load(open(pmc_path))
            if pmc.get("workload") == args.workload and pmc.get("frames") == T and \
                    pmc.get("streams") == S:
                traffic = pmc.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
        "kernel": "k_eq_pipe" if eq else "k_run_fast", "kernel_avg_ms": round(kern_avg_ms, 4),
        "launches": launches, "algorithmic_bytes_per_sample": bps,
        "algorithmic_bytes_per_launch": samples_per_step_rank * bps,
    }

    out = {
        "metric": "Msamples/s transform->vumeter", "value": round(value, 1), "unit": "Msamples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "int16" if not eq else "f32",
        "arithmetic": "int16 PCM, exact int32 products / division, int64 VU accumulation, dB in f64 on the host"
        if not eq else "int16 in, exact integer gain, f32 biquads (fixed fmaf order), f32 out",
        "data": "synthetic (per-stream LCG noise generated on device, seed 12345 + stream id)",
        "config": {"workload": "%s: %s" % (args.workload, desc), "streams_per_gpu": S,
                   "channels": Cn, "frames_per_launch": T, "sharding": "stream s -> rank s %% %d" % world,
                   "collective": ("RCCL all-gather of %d blocks' node records (34 x int64 each), combined on the "
                                  "device" % NB) if node_vu else "none"},
        "roofline": roofline,
    }

    if rank == 0 and not args.no_extras and not eq:
        extras = {}
        try:
            extras["hbm_read_ceiling_GBs"] = round(b.ceiling(0, 10), 1)
            extras["hbm_copy_ceiling_GBs"] = round(b.ceiling(1, 10), 1)
        except Exception as e:           # measurement extras must not break the line
            extras["ceiling_error"] = str(e)
        out["measured_ceilings"] = extras
        # the small-block regime, same batch, fewer frames per launch (kernel only)
        sweep = {}
        try:
            for frames in (512, 4096):
                if frames >= T:
                    continue
                b.vu_reset(-1)
                for _ in range(3):
                    b.run(frames)
                b.sync()
                b.timing(True)
                b.timing_read()
                for _ in range(20):
                    b.run(frames)
                ms, n = b.timing_read()
                b.timing(False)
                sweep[str(frames)] = {"kernel_avg_ms": round(ms / n, 4),
                                      "achieved_GBs": round(S * Cn * frames * bps / (ms / n * 1e-3) / 1e9, 1)}
        except Exception as e:
            sweep["error"] = str(e)
        out["small_blocks_kernel_only"] = sweep
    b.close()

    if rank == 0 and not args.no_extras and not eq:
        # second line of SURVEY 8(d): VU only, 2 B/sample read -- never mixed with the above
        v = cm.Batch(S, Cn, T, flags=cm.VU, device=local_rank)
        if args.workload == "c2":
            v.set_gain(-1, 2, 1000, [750, 1250])
            v.set_chmap(-1, [1, 0])
        else:
            v.set_gain(-1, 1, 1000, [900])
        v.generate(cm.GEN_NOISE, 12345, T, first_global=rank, global_step=world)
        for _ in range(100):
            v.run(T)
        v.sync()
        v.timing(True)
        v.timing_read()
        for _ in range(100):
            v.run(T)
        ms, n = v.timing_read()
        v.close()
        gbs = samples_per_step_rank * 2 / (ms / n * 1e-3) / 1e9
        out["vu_only"] = {"kernel_avg_ms": round(ms / n, 4), "achieved_GBs": round(gbs, 1),
                          "frac_of_peak": round(gbs / HBM_PEAK_GBS, 4),
                          "Msamples_per_s_kernel": round(samples_per_step_rank / (ms / n * 1e-3) / 1e6, 1),
                          "algorithmic_bytes_per_sample": 2}
        # the same read-only run with the transform as the reference creates it (gain disabled,
        # ref: src/transform.c:107-108) and with every gain below the scale: shorter arithmetic
        for key, g in (("vu_only_gain_disabled", None), ("vu_only_gains_below_scale", [900, 800][:Cn])):
            v = cm.Batch(S, Cn, T, flags=cm.VU, device=local_rank)
            if g is not None:
                v.set_gain(-1, Cn, 1000, g)
            if args.workload == "c2":
                v.set_chmap(-1, [1, 0])
            v.generate(cm.GEN_NOISE, 12345, T, first_global=rank, global_step=world)
            for _ in range(100):
                v.run(T)
            v.sync()
            v.timing(True)
            v.timing_read()
            for _ in range(100):
                v.run(T)
            ms, n = v.timing_read()
            v.close()
            gbs = samples_per_step_rank * 2 / (ms / n * 1e-3) / 1e9
            out[key] = {"kernel_avg_ms": round(ms / n, 4), "achieved_GBs": round(gbs, 1),
                        "frac_of_peak": round(gbs / HBM_PEAK_GBS, 4)}

    if rank == 0 and world == 1 and not args.no_extras and args.workload == "c2":
        # the other kernels of the path, kernel time only: never part of `value`
        other = {}
        try:
            for name, (s_, c_, t_, fl, bps_, eqz) in {
                    "c3_eq_float_planes": (8192, 1, 65536, cm.EQ | cm.OUT_F32, 6, True),
                    "eq_stereo_int16_vu": (4096, 2, 65536, cm.EQ | cm.OUT_PCM | cm.VU, 4, True),
                    "six_channels_pcm_vu": (2730, 6, 16384, cm.OUT_PCM | cm.VU, 4, False)}.items():
                o = cm.Batch(s_, c_, t_, flags=fl, device=local_rank)
                o.set_gain(-1, 1, 1000, [900])
                if eqz:
                    o.set_eq(-1, cm.eq3(48000.0))
                o.generate(cm.GEN_NOISE, 12345, t_)
                for _ in range(100):                 # ~0.1 s: the clocks the chip then holds
                    o.run(t_)
                o.sync()
                o.timing(True)
                o.timing_read()
                for _ in range(100):
                    o.run(t_)
                ms, n = o.timing_read()
                o.close()
                gbs = s_ * c_ * t_ * bps_ / (ms / n * 1e-3) / 1e9
                other[name] = {"streams": s_, "channels": c_, "frames": t_, "kernel_avg_ms": round(ms / n, 4),
                               "algorithmic_bytes_per_sample": bps_, "achieved_GBs": round(gbs, 1),
                               "frac_of_peak": round(gbs / HBM_PEAK_GBS, 4)}
        except Exception as e:
            other["error"] = str(e)
        out["other_kernels"] = other

    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(args.workload, Cn)

    if need_torch:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)


def cpu_baseline(workload, channels):
    """The CPU oracle (kind "port": scalar restatement of src/transform.c:101-124 and
    src/vumeter.c:161-218) on this host, on a bounded sample of the same workload."""
    from oracle import oracle_ffi
    lib = oracle_ffi.load()
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    frames = 65536
    per_thread = 48                      # streams per thread: ~10-25 s of CPU work in total
    streams = cores * per_thread
    if channels == 2:
        gain = (C.c_uint16 * 2)(750, 1250)
        cmap = (C.c_uint8 * 2)(1, 0)
        cmap_p = C.cast(cmap, C.c_void_p)
    else:
        gain = (C.c_uint16 * 1)(900)
        cmap_p = None
    chk = C.c_uint64()
    secs = lib.oracle_bench_block(cores, streams, channels, frames, cmap_p, 1000, gain, 12345,
                                  C.byref(chk))
    n_all = streams * frames * channels
    secs1 = lib.oracle_bench_block(1, per_thread, channels, frames, cmap_p, 1000, gain, 12345,
                                   C.byref(chk))
    n_one = per_thread * frames * channels
    chain_frames = 20_000_000
    secs_chain = lib.oracle_bench_chain(chain_frames, 1000, 900, C.byref(chk))
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    return {
        "value": round(n_all / secs / 1e6, 1), "unit": "Msamples/s", "cores": cores, "kind": "port",
        "sample": "%d of the workload's streams (%d per thread) x %d frames x %d ch, same "
                  "generator and parameters, block-at-once" % (streams, per_thread, frames, channels),
        "one_thread_Msamples_s": round(n_one / secs1 / 1e6, 1),
        "pull_chain_1024B_one_thread_Msamples_s": round(chain_frames / secs_chain / 1e6, 1),
        "cpu_model": model,
    }


if __name__ == "__main__":
    main()
