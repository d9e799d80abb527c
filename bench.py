#!/usr/bin/env python3
"""bench.py -- Msamples/s through transform -> vumeter on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload c2|c2ro|c4|c3|c5] [--gain general|below|off]

A step is one pass of the hot path over one batch that is already resident in HBM:
one fused gain(+channel map) -> VU launch over every stream of the rank's shard, the
asynchronous snapshot of all VU windows to the host, and the host-side dB finish
(double, as the reference) of the previous step's windows, overlapped with the GPU.

One process per GPU.  `python bench.py --gpus N` with N > 1 starts the N rank processes
itself (before anything touches HIP) and relays rank 0's line; under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the ranks are
already there (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment).

Workloads (per GPU; stream s of the node lives on rank s % N -- round-robin shards,
no data-path collective, "weak" scaling):
  c2  4096 stereo streams x 65536 frames, channel swap + gains {750,1250}/1000, PCM
      materialised (2 B read + 2 B written per sample)             [default, configs[1]]
  c2ro the same streams, VU only: nothing written, 2 B read per sample -- the north star's HBM-READ
      roofline; --gain picks the arithmetic form (general {750,1250} | below {900,800} | off)
  c4  8192 mono streams x 65536 frames, gain 900/1000, PCM materialised  [configs[3]]
  c5  c4 + node-global VU: RCCL all-reduce of the blocks' records         [configs[4]]
  c3  8192 mono streams, int16 -> float + 3-band EQ, float out           [configs[2]]
  x6  2730 six-channel streams x 16384 frames, PCM + VU (the many-channel kernel; for profiles/)

Rank 0 prints ONE JSON line.  `value` and `roofline` come from a batch created the library's
DEFAULT way (two plain allocations; the engine's opt-in placement search is a leg of its own,
`setup.place_search`, after the benchmarked batch is gone).  `roofline` is for the dominant
kernel, timed with HIP events on the stream it is launched on; `cpu_baseline` is the CPU oracle
(the scalar restatement of the reference loops) timed on this host, N=1 only.

The process plumbing, the measurement legs beside `value` and the CPU baseline live in benchlib/.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchlib import launch                      # noqa: E402  (no HIP, no torch: safe before the ranks are started)
from benchlib.launch import MIN_WARMUP_S         # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

WORKLOADS = {
    # name: (streams/GPU, channels, frames, bytes per sample, description)
    "c2": (4096, 2, 65536, 4, "4096 stereo int16 48 kHz streams x 65536 frames per GPU, "
                              "channel swap + gains {750,1250}/1000 -> VU, PCM materialised"),
    "c2ro": (4096, 2, 65536, 2, "4096 stereo int16 48 kHz streams x 65536 frames per GPU, channel swap + gain "
                                "-> VU only: nothing written, 2 B read per sample (the HBM-read roofline)"),
    "c4": (8192, 1, 65536, 4, "8192 mono int16 streams x 65536 frames per GPU (65536 streams "
                              "round-robin over 8 GPUs), gain 900/1000 -> VU, PCM materialised"),
    "c5": (8192, 1, 65536, 4, "c4 + node-global VU via RCCL all-reduce of the blocks' records"),
    "c3": (8192, 1, 65536, 6, "8192 mono streams x 65536 frames per GPU, int16 -> float + "
                              "3-band biquad EQ, planar float out"),
    # not a BASELINE config: the many-channel kernel (k_run_rows) under the same harness, for profiles/
    "x6": (2730, 6, 16384, 4, "2730 six-channel (5.1) streams x 16384 frames per GPU, gain 900/1000 -> VU, "
                              "PCM materialised"),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 0.1 s of warm-up and 0.35 s of timed steps; whatever --warmup says, warm-up
    # continues until MIN_WARMUP_S of wall time have passed (reported as warmup_ms_effective)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--frames", type=int, default=0, help="override frames per launch")
    ap.add_argument("--streams", type=int, default=0, help="override streams per GPU")
    ap.add_argument("--strong", action="store_true",
                    help="fixed total: the workload's streams are divided among the ranks (default: "
                         "weak scaling, the workload's streams per GPU)")
    ap.add_argument("--node-batch", type=int, default=8,
                    help="config 5: blocks whose node-global VU records travel in one all-reduce pair")
    ap.add_argument("--gain", default="general", choices=("general", "below", "off"),
                    help="arithmetic form: the workload's own gains (general), every gain below the scale "
                         "(one mulhi per sample in the read-only runs), or the gain disabled as the reference "
                         "creates a transform")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extras", action="store_true", help="skip ceilings / VU-only line / PCIe line")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch.launch_ranks(args.gpus))

    # stdout carries exactly one line, rank 0's JSON: whatever libraries print on the way (gloo
    # announces its connections on stdout) goes to stderr with everything else
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    # several processes on the GPUs of one host: the driver here supports dmabuf IPC only, and RCCL's
    # hipIpcGetMemHandle fails without this (set before anything initialises HIP; a value already there stays)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    # torch is plumbing for the multi-rank run only: rendezvous, the barrier and the clock over
    # the ranks, all over gloo on the CPU -- torch.cuda is never touched.  It brings its own HIP
    # runtime (ROCm 7.0 inside the wheel), so the engine (and for config 5 librccl, through
    # cmhip_node_new) is loaded FIRST and stays on the system runtime; a single rank never loads
    # torch.  The one real exchange, config 5's node-global VU, is the engine's own RCCL call.
    force_node = os.environ.get("COOLMIC_BENCH_FORCE_NODE") == "1"       # single-rank run of the reduce path
    # Rehearsal knob for a 1-GPU box (never set by the driver): all ranks share device 0; RCCL
    # refuses two ranks on one GPU, so every rank reduces in a one-rank communicator and the
    # records are merged over gloo on the host (cmhip_node_merge_host, the "replicas only" form).
    rehearsal = os.environ.get("COOLMIC_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0

    # Several ranks share the host: each sizes its dB-finish helper pool for its share of the container's
    # CPU-time quota (the engine alone would size it for the whole quota)
    if world > 1 and "CMHIP_POOL_THREADS" not in os.environ:
        from benchlib.cpu import cpu_quota
        quota = cpu_quota()
        if quota is not None and quota / world < 16:
            os.environ["CMHIP_POOL_THREADS"] = str(max(1, int(quota / world) - 3))

    import __graft_entry__ as ge
    cm = ge.load_package()
    from libcoolmic_dsp_amd import shard
    from benchlib import legs
    from benchlib.cpu import cpu_baseline

    if os.environ.get("COOLMIC_BENCH_DRYRUN") == "1":
        # launch plumbing only, for the CPU tests: rendezvous, node-id exchange, barrier, the
        # max-over-ranks clock, one JSON line from rank 0 -- no GPU work, no throughput
        return launch.dry_run(args, rank, world, json_fd, shard)
    if cm.device_count() < 1:
        sys.exit("bench.py: no HIP device; this path has no CPU fallback")

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
    node_on = node_vu and (world > 1 or force_node)
    NB = max(1, args.node_batch)

    read_only = args.workload == "c2ro"
    if eq:
        flags = cm.EQ | cm.OUT_F32
    elif read_only:
        flags = cm.VU
    else:
        flags = cm.OUT_PCM | cm.VU
    if os.environ.get("COOLMIC_BENCH_GAIN") == "0":       # (round 3's knob)
        args.gain = "off"
    gains, swap = legs.workload_gains(args.workload, Cn, args.gain)

    def make_batch(extra_flags=0):
        bb = cm.Batch(S, Cn, T, flags=flags | extra_flags, device=local_rank)
        if gains is not None:
            assert bb.set_gain(-1, len(gains), 1000, gains) == 0
        if swap:
            assert bb.set_chmap(-1, [1, 0]) == 0
        if eq:
            assert bb.set_eq(-1, cm.eq3(48000.0)) == 0
        # global stream id of local stream s is rank + s*world (round-robin sharding)
        bb.generate(cm.GEN_NOISE, 12345, T, first_global=first_global, global_step=global_step)
        bb.sync()
        return bb

    n_local, first_global, global_step = shard.shard(S * world, world, rank)
    assert n_local == S

    # The benchmarked batch is created the LIBRARY's default way: two plain allocations, no probe launch.  What
    # the engine's opt-in placement search would add on this box is a leg of its own at the end.
    setup = {"placement_search": "off for `value` and `roofline` (the library's default); see setup.place_search"}
    t_c = time.perf_counter()
    b = make_batch()
    setup["batch_create_ms"] = round((time.perf_counter() - t_c) * 1e3, 1)
    setup["placement"] = b.placement()

    # (the batch's PCM arrays are allocated before RCCL takes its buffers: the same order of
    # allocations as in the workloads without an exchange)
    node = None
    if node_on:
        if rehearsal:
            node = cm.Node(local_rank, 1, 0, cm.node_unique_id(), max_records=NB)
        else:
            uid = launch.exchange_node_id(rank, world, cm.node_unique_id)
            node = cm.Node(local_rank, world, rank, uid, max_records=NB)

    torch = dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29599")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    has_vu = bool(flags & cm.VU)
    results = (cm.VuResult * S)()
    rcs = (C.c_int * S)()
    # node-global VU (config 5): every block leaves one 34-word record per rank in a slot of the
    # node's current record set; the records of --node-batch blocks are reduced over the ranks by
    # ONE pair of RCCL all-reduces (the exchange is latency bound), on the node's own stream,
    # beside the next blocks' kernels, which fill the other set.  The host never waits in the loop.
    node_step = [0]
    node_last = [None]                     # (set, count) of the last exchange

    def node_exchange(k, count):
        node.allreduce(k, count, after=b)
        node_last[0] = (k, count)

    def device_sync():
        cm.device_synchronize(local_rank)       # hipDeviceSynchronize: every stream of this rank's GPU

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
                node.partial(b, k, slot, first_global=rank, global_step=world)
                if slot == NB - 1:
                    node_exchange(k, NB)
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
            node_exchange((node_step[0] // NB) & 1, node_step[0] % NB)
            node_step[0] += NB - node_step[0] % NB
        b.sync()

    # warm-up: the steps asked for, then on until MIN_WARMUP_S of wall time have passed, so that a
    # short run (the driver's --steps 20 --warmup 5) is timed at the clocks the chip then holds.  How
    # many chunks that takes is decided once for all ranks (collective_more): with a data-path
    # collective in the steps (config 5) every rank must issue the same number of them.
    t_w = time.perf_counter()
    # (with the benchmarked kernel itself: 0.15 s of another kernel -- a plain read or copy of the batch's arrays --
    # in front of a short --warmup left the timed launches 3 % slower than 0.15 s of the kernel's own launches,
    # round 4.  A rocprofv3 trace of this command therefore holds the ramp: tools/trace_summary.py averages the
    # timed region's launches beside the whole trace.)
    chunk = max(8, NB if node_on else 8)
    warm_steps = launch.warm_up(run_steps, args.warmup, chunk,
                                launch.collective_more(dist, rank, world, t_w, MIN_WARMUP_S))
    device_sync()
    warm_ms = (time.perf_counter() - t_w) * 1e3

    # Kernel time: the kernel's own dispatch stamps HIP events on its stream.  The events cost a launch about
    # 5 us of its stream's time (tools/step_overhead.py: 0.3334 -> 0.3380 ms per step of config 2), so every
    # eighth launch of the timed region carries them -- every one when the region is short (the driver's --steps 20:
    # SURVEY 8(d) wants the kernel time over at least 20 timed launches).
    time_every = 8 if args.steps >= 64 else 1
    b.timing(time_every)
    b.timing_read()
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    kern_ms, launches = b.timing_read()
    b.timing(False)

    if world > 1:
        dt = shard.max_over_ranks(dist, dt, device="cpu")

    samples_per_step_rank = S * Cn * T
    total_samples = samples_per_step_rank * world * args.steps
    value = total_samples / dt / 1e6

    kern_avg_ms = kern_ms / max(launches, 1)
    achieved = samples_per_step_rank * bps / (kern_avg_ms * 1e-3) / 1e9 if kern_avg_ms > 0 else 0.0
    traffic, traffic_source = legs.traffic_from_profiles(args.workload, S, T, args.gain)
    roofline = {
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
        "kernel": "k_eq_pipe" if eq else ("k_run_rows" if Cn > 2 else ("k_run_fast_ro" if read_only else "k_run_fast")),
        "kernel_avg_ms": round(kern_avg_ms, 4),
        "launches": launches, "launches_timed": "every %d%s of the %d timed steps" % (
            time_every, "th" if time_every > 1 else "", args.steps) if time_every > 1 else "all",
        "algorithmic_bytes_per_sample": bps,
        "algorithmic_bytes_per_launch": samples_per_step_rank * bps,
    }

    collective = "none"
    if node_vu:
        collective = ("RCCL: per %d blocks one ncclAllReduce(int64, sum) + one ncclAllReduce(uint64, max) over "
                      "their node records (17 + 17 words each), through cmhip_node_allreduce" % NB)
        if rehearsal and world > 1:
            collective += " [rehearsal: one-rank communicators + host merge over gloo]"
        elif not node_on:
            collective += " [one rank: nothing to exchange]"
    out = {
        "metric": "Msamples/s transform->vumeter", "value": round(value, 1), "unit": "Msamples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "warmup_steps_effective": warm_steps, "warmup_ms_effective": round(warm_ms, 1),
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "int16" if not eq else "f32",
        "arithmetic": "int16 PCM, exact integer gain (|x|*mi + mulhi(|x|, mf)), int64 VU accumulation, dB in f64 on the host"
        if not eq else "int16 in, exact integer gain, f32 biquads (fixed fmaf order), f32 out",
        "data": "synthetic (per-stream LCG noise generated on device, seed 12345 + stream id)",
        "config": {"workload": "%s: %s" % (args.workload, desc), "streams_per_gpu": S,
                   "channels": Cn, "frames_per_launch": T, "sharding": "stream s -> rank s %% %d" % world,
                   "gain_form": args.gain, "gains_over_1000": gains, "channel_swap": swap,
                   "collective": collective},
        "per_gpu_Msamples_s": round(value / world, 1),
        "roofline": roofline,
    }
    if rehearsal:
        out["rehearsal"] = "all %d ranks share GPU 0 (COOLMIC_BENCH_REHEARSAL=1): not a scaling number" % world
    out["setup"] = setup

    if node_on and node_last[0] is not None:
        # the last exchanged set, decoded (outside the timed region)
        k, count = node_last[0]
        words = node.fetch(k, count)[count - 1]
        if rehearsal and world > 1:
            mine = torch.from_numpy(words.copy())
            parts = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine)
            import numpy as np
            words = cm.node_merge_host(np.stack([p.numpy() for p in parts]))
        rc, r = cm.node_finish(words, Cn)
        if rc == 0:
            out["node_vu_last_block"] = {"frames": r.frames, "global_peak": r.global_peak,
                                         "global_power_db": r.global_power}
    if node is not None:
        node.close()

    # per-GPU efficiency against the committed N=1 line of the same workload (the driver computes
    # its own from its own runs; this is a convenience for a reader of the line)
    if rank == 0 and world > 1:
        ref = os.path.join(ROOT, "profiles", "n1_%s.json" % args.workload)
        try:
            one = json.load(open(ref))
            if one.get("config", {}).get("streams_per_gpu") == S and \
                    one.get("config", {}).get("frames_per_launch") == T and not args.strong:
                out["per_gpu_efficiency"] = round(value / world / one["value"], 4)
                out["per_gpu_efficiency_source"] = "profiles/n1_%s.json (N=1 line of the same workload)" % args.workload
        except Exception:
            pass

    # the parity gate that goes with every benchmark -- one more block of the same batch, outside the
    # timed region, sampled streams against the oracle (which is here as the checker, as in cpu_baseline)
    if rank == 0 and not args.no_cpu:
        out["parity_gate"] = legs.parity_gate(cm, b, args.workload, S, Cn, T, first_global, global_step, gains, swap,
                                              has_pcm=not read_only)

    if rank == 0 and not args.no_extras and not eq and not read_only:
        out["measured_ceilings"] = legs.measured_ceilings(b, achieved)
        out["small_blocks"] = legs.small_blocks(b, S, Cn, T, bps, results, rcs)
    if rank == 0 and not args.no_extras and read_only:
        try:
            ceiling = round(b.ceiling(0, 10), 1)
            out["measured_ceilings"] = {"hbm_read_ceiling_GBs": ceiling,
                                        "kernel_frac_of_read_ceiling": round(achieved / ceiling, 4) if ceiling > 0 else None}
        except Exception as e:
            out["measured_ceilings"] = {"ceiling_error": str(e)}
    b.close()

    if rank == 0 and not args.no_extras and not eq and not read_only:
        read_ceiling = out.get("measured_ceilings", {}).get("hbm_read_ceiling_GBs", 0)
        out.update(legs.vu_only_lines(cm, args.workload, S, Cn, T, local_rank, rank, world, read_ceiling))
    if rank == 0 and world == 1 and not args.no_extras and args.workload == "c2":
        out["other_kernels"] = legs.other_kernels(cm, local_rank)
        try:
            out["pcie_inclusive"] = legs.pcie_inclusive(cm, local_rank)
        except Exception as e:
            out["pcie_inclusive"] = {"error": str(e)}
        if os.environ.get("COOLMIC_BENCH_PLACE", "1") != "0":
            setup["place_search"] = legs.place_search_leg(cm, make_batch, T)

    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(args.workload, Cn, gains, swap)

    # Several ranks: configs 4 and 5 at this N, after everything else of the line is known, outside the timed
    # region and never part of `value` -- the driver's one command passes no --workload, and config 5's
    # exchange is the only collective of the path.  Guarded: whatever happens in there, rank 0's line goes out.
    printer = launch.LinePrinter(rank, json_fd, out)
    if world > 1:
        launch.guarded_legs(printer, dist,
                            lambda: legs.node_vu_legs(cm, shard, dist, rank, world, local_rank, rehearsal, NB))
    sys.stdout.flush()
    printer.emit()
    if out.get("parity_gate", {}).get("ok") is False:
        sys.exit("bench.py: the parity gate failed -- the numbers above are not results of the reference's arithmetic")


if __name__ == "__main__":
    main()
