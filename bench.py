#!/usr/bin/env python3
"""bench.py -- Msamples/s through transform -> vumeter on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload c2|c4|c3|c5]

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
  c4  8192 mono streams x 65536 frames, gain 900/1000, PCM materialised  [configs[3]]
  c5  c4 + node-global VU: RCCL all-reduce of the blocks' records         [configs[4]]
  c3  8192 mono streams, int16 -> float + 3-band EQ, float out           [configs[2]]
  x6  2730 six-channel streams x 16384 frames, PCM + VU (the many-channel kernel; for profiles/)

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel, timed with HIP
events on the stream it is launched on; `cpu_baseline` is the CPU oracle (the scalar
restatement of the reference loops) timed on this host, N=1 only.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MIN_WARMUP_S = 0.15            # the chip reaches the clocks it then holds after ~100 ms of load

WORKLOADS = {
    # name: (streams/GPU, channels, frames, bytes per sample, description)
    "c2": (4096, 2, 65536, 4, "4096 stereo int16 48 kHz streams x 65536 frames per GPU, "
                              "channel swap + gains {750,1250}/1000 -> VU, PCM materialised"),
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
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extras", action="store_true", help="skip ceilings / VU-only line / PCIe line")
    return ap.parse_args(argv)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def under_profiler():
    """rocprofv3 preloads its tool library (LD_PRELOAD = ...librocprofiler-sdk-tool.so:librocprofiler-sdk.so,
    ROCP_TOOL_LIBRARIES names it), which initialises the GPU before main() runs: starting rank processes
    from such a process is an exec from one that holds the GPU."""
    return "librocprofiler-sdk" in os.environ.get("LD_PRELOAD", "") or bool(os.environ.get("ROCP_TOOL_LIBRARIES"))


def launch_ranks(n):
    """The parent of a self-launched multi-GPU run: N fresh rank processes of this script, one
    per GPU, started before this process has loaded the engine or touched HIP (nothing is
    exec'ed from a process that initialised the GPU).  Relays rank 0's JSON line; returns the
    worst exit code.  The whole launch has a wall-clock deadline ($COOLMIC_BENCH_DEADLINE_S,
    default 900): ranks that are all alive but stuck -- a collective one of them never issued, a
    hung GPU -- are ended (exactly the processes started here) and named, instead of leaving the
    one command the driver runs without a line until gloo's half-hour timeout."""
    if under_profiler():
        sys.stderr.write("bench.py: --gpus %d under a profiler preload (rocprofv3): the profiler's library has "
                         "initialised the GPU in this process, so it must not start the rank processes.  Profile "
                         "one rank directly: RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 rocprofv3 ... -- python3 bench.py\n" % n)
        return 2
    env0 = dict(os.environ)
    env0.setdefault("MASTER_ADDR", "127.0.0.1")
    env0.setdefault("MASTER_PORT", str(_free_port()))
    env0["WORLD_SIZE"] = str(n)
    env0["LOCAL_WORLD_SIZE"] = str(n)
    procs = []
    for r in range(n):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    lines = []

    def drain():                                # rank 0 prints exactly one line on stdout
        for raw in procs[0].stdout:
            if raw.strip():
                lines.append(raw)

    import threading
    reader = threading.Thread(target=drain, daemon=True)
    reader.start()
    worst = 0
    t_start = time.time()
    overall = t_start + float(os.environ.get("COOLMIC_BENCH_DEADLINE_S", "900"))
    deadline = None
    pending = list(procs)
    while pending:
        for p in list(pending):
            rc = p.poll()
            if rc is None:
                continue
            pending.remove(p)
            if rc != 0:
                worst = worst or rc
                if deadline is None:            # a rank died: the others would wait at a barrier for ever
                    deadline = time.time() + 20.0
        now = time.time()
        if pending and now > overall:
            alive = [procs.index(p) for p in pending]
            sys.stderr.write("bench.py: deadline of %.0f s passed with rank(s) %s still running (stuck at a "
                             "collective or a hung GPU?); ending them\n" % (now - t_start, alive))
            worst = worst or 124
            deadline = now - 1.0
            overall = now + 1e9
        if deadline is not None and now > deadline:
            for p in pending:
                p.kill()                        # exactly the processes started above
            deadline = now + 1e9
        time.sleep(0.05)
    reader.join(timeout=10)
    line = lines[-1] if lines else b""
    if line:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    elif worst == 0:
        worst = 1
    return worst


def job_token():
    """what tells this job's ranks from another job's on the same host"""
    import hashlib
    key = ":".join(os.environ.get(k, "") for k in ("MASTER_ADDR", "MASTER_PORT", "WORLD_SIZE", "TORCHELASTIC_RUN_ID"))
    return hashlib.sha256(key.encode()).digest()[:16]


def exchange_node_id(rank, world, make_id):
    """Config 5: rank 0's 128-byte RCCL id reaches the other ranks over a plain TCP socket next
    to MASTER_PORT -- before torch is imported, so that the engine and librccl both sit on the
    system HIP runtime (torch, imported later for gloo only, brings a second one).  A client says
    who it is (magic, job token, rank); the server answers valid requests only and counts distinct
    ranks, so a stray connection or another job's rank takes nobody's place."""
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    base = int(os.environ.get("MASTER_PORT", "29599"))
    ports = [base + 101 + 37 * i for i in range(8)]
    magic = b"cmhip-node-id:"
    token = job_token()
    if world == 1:
        return make_id()
    if rank == 0:
        uid = make_id()
        srv = None
        for p in ports:
            try:
                srv = socket.create_server((addr, p), reuse_port=False)
                break
            except OSError:
                continue
        if srv is None:
            raise SystemExit("bench.py: no free port for the node id exchange near MASTER_PORT")
        t_end = time.time() + 120
        served = set()
        want = len(magic) + len(token) + 4
        while len(served) < world - 1:
            srv.settimeout(max(0.1, t_end - time.time()))
            try:
                conn, _ = srv.accept()
            except (socket.timeout, TimeoutError):
                raise SystemExit("bench.py: node id exchange: only rank(s) %s of %d asked for the id within 120 s"
                                 % (sorted(served), world))
            with conn:
                conn.settimeout(5)
                try:
                    req = b""
                    while len(req) < want:
                        chunk = conn.recv(want - len(req))
                        if not chunk:
                            break
                        req += chunk
                    peer = int.from_bytes(req[-4:], "little") if len(req) == want else -1
                    if req.startswith(magic + token) and 0 < peer < world:
                        conn.sendall(magic + token + uid)
                        served.add(peer)
                except OSError:
                    pass                          # whoever that was, it was not one of ours
        srv.close()
        return uid
    hello = magic + token + rank.to_bytes(4, "little")
    want = len(magic) + len(token) + 128
    t_end = time.time() + 120
    while time.time() < t_end:
        for p in ports:
            try:
                with socket.create_connection((addr, p), timeout=2) as c:
                    c.sendall(hello)
                    buf = b""
                    while len(buf) < want:
                        chunk = c.recv(want - len(buf))
                        if not chunk:
                            break
                        buf += chunk
                if len(buf) == want and buf.startswith(magic + token):
                    return buf[len(magic) + len(token):]
            except OSError:
                continue
        time.sleep(0.1)
    raise SystemExit("bench.py: rank %d never received the node id" % rank)


def warm_up(run_steps, steps, chunk, more):
    """the steps asked for, then on in chunks while more() says so.  With several ranks more() is ONE
    decision for all of them (rank 0's clock, broadcast): every rank runs the same number of steps, so
    ranks that issue collectives in their steps (config 5) issue the same number of them."""
    run_steps(steps)
    done = steps
    while more():
        run_steps(chunk)
        done += chunk
    return done


def collective_more(dist, rank, world, t_start, min_s):
    """-> more(): has MIN_WARMUP_S of wall time passed?  One rank: its own clock.  Several: rank 0's,
    broadcast over gloo, so that all ranks leave the warm-up loop after the same chunk."""
    if world == 1:
        return lambda: time.perf_counter() - t_start < min_s
    import torch

    def more():
        flag = torch.tensor([1 if (rank == 0 and time.perf_counter() - t_start < min_s) else 0], dtype=torch.int32)
        dist.broadcast(flag, src=0)
        return bool(flag.item())
    return more


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))

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
        quota = cpu_quota()
        if quota is not None and quota / world < 16:
            os.environ["CMHIP_POOL_THREADS"] = str(max(1, int(quota / world) - 3))

    import __graft_entry__ as ge
    cm = ge.load_package()
    from libcoolmic_dsp_amd import shard

    if os.environ.get("COOLMIC_BENCH_DRYRUN") == "1":
        # launch plumbing only, for the CPU tests: rendezvous, node-id exchange, barrier, the
        # max-over-ranks clock, one JSON line from rank 0 -- no GPU work, no throughput
        return dry_run(args, rank, world, json_fd, shard)
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

    if eq:
        flags = cm.EQ | cm.OUT_F32
    else:
        flags = cm.OUT_PCM | cm.VU

    def make_batch(extra_flags=0):
        bb = cm.Batch(S, Cn, T, flags=flags | extra_flags, device=local_rank)
        if args.workload == "c2":
            assert bb.set_gain(-1, 2, 1000, [750, 1250]) == 0
            assert bb.set_chmap(-1, [1, 0]) == 0
        elif os.environ.get("COOLMIC_BENCH_GAIN", "1") != "0":
            assert bb.set_gain(-1, 1, 1000, [900]) == 0
        if eq:
            assert bb.set_eq(-1, cm.eq3(48000.0)) == 0
        # global stream id of local stream s is rank + s*world (round-robin sharding)
        bb.generate(cm.GEN_NOISE, 12345, T, first_global=first_global, global_step=global_step)
        bb.sync()
        return bb

    n_local, first_global, global_step = shard.shard(S * world, world, rank)
    assert n_local == S

    # Placement of the two PCM arrays: the LIBRARY's default is two plain allocations; this
    # benchmark opts in to the engine's placement search (CMHIP_PLACE_SEARCH) and says so in the line --
    # together with what the same workload's kernel takes in this process WITHOUT it: a batch made the
    # default way first (the process's first two large allocations), timed, freed.
    place_search = not eq and os.environ.get("COOLMIC_BENCH_PLACE", "1") != "0"
    setup = {"placement_search": "on (bench.py passes CMHIP_PLACE_SEARCH; the library's default is off)"
             if place_search else "off"}
    if place_search:
        setup.update(place_off_leg(make_batch, T))
    t_c = time.perf_counter()
    b = make_batch(cm.PLACE_SEARCH if place_search else 0)
    setup["batch_create_ms"] = round((time.perf_counter() - t_c) * 1e3, 1)
    setup["placement"] = b.placement()

    # (the batch's PCM arrays are allocated before RCCL takes its buffers: the same order of
    # allocations as in the workloads without an exchange)
    node = None
    if node_on:
        if rehearsal:
            node = cm.Node(local_rank, 1, 0, cm.node_unique_id(), max_records=NB)
        else:
            uid = exchange_node_id(rank, world, cm.node_unique_id)
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
    chunk = max(8, NB if node_on else 8)
    warm_steps = warm_up(run_steps, args.warmup, chunk, collective_more(dist, rank, world, t_w, MIN_WARMUP_S))
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
    traffic, traffic_source = traffic_from_profiles(args.workload, S, T)
    roofline = {
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
        "kernel": "k_eq_pipe" if eq else ("k_run_rows" if Cn > 2 else "k_run_fast"), "kernel_avg_ms": round(kern_avg_ms, 4),
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
        "arithmetic": "int16 PCM, exact int32 products / division, int64 VU accumulation, dB in f64 on the host"
        if not eq else "int16 in, exact integer gain, f32 biquads (fixed fmaf order), f32 out",
        "data": "synthetic (per-stream LCG noise generated on device, seed 12345 + stream id)",
        "config": {"workload": "%s: %s" % (args.workload, desc), "streams_per_gpu": S,
                   "channels": Cn, "frames_per_launch": T, "sharding": "stream s -> rank s %% %d" % world,
                   "collective": collective},
        "per_gpu_Msamples_s": round(value / world, 1),
        "roofline": roofline,
    }
    if rehearsal:
        out["rehearsal"] = "all %d ranks share GPU 0 (COOLMIC_BENCH_REHEARSAL=1): not a scaling number" % world
    out["setup"] = setup
    if "kernel_avg_ms_place_off" in setup:
        out["kernel_avg_ms_place_off"] = setup["kernel_avg_ms_place_off"]
        out["config"]["workload"] += " [PCM arrays placed by the engine's search, CMHIP_PLACE_SEARCH: see setup]"

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
        out["parity_gate"] = parity_gate(cm, b, args.workload, S, Cn, T, first_global, global_step)

    if rank == 0 and not args.no_extras and not eq:
        out["measured_ceilings"] = measured_ceilings(b, achieved)
        out["small_blocks"] = small_blocks(b, S, Cn, T, bps, results, rcs)
    b.close()

    if rank == 0 and not args.no_extras and not eq:
        read_ceiling = out.get("measured_ceilings", {}).get("hbm_read_ceiling_GBs", 0)
        out.update(vu_only_lines(cm, args.workload, S, Cn, T, local_rank, rank, world, read_ceiling))
    if rank == 0 and world == 1 and not args.no_extras and args.workload == "c2":
        out["other_kernels"] = other_kernels(cm, local_rank)
        try:
            out["pcie_inclusive"] = pcie_inclusive(cm, local_rank)
        except Exception as e:
            out["pcie_inclusive"] = {"error": str(e)}

    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(args.workload, Cn)

    # Several ranks: configs 4 and 5 at this N, after everything else of the line is known, outside the timed
    # region and never part of `value` -- the driver's one command passes no --workload, and config 5's
    # exchange is the only collective of the path.  Guarded: whatever happens in there, rank 0's line goes out.
    printer = LinePrinter(rank, json_fd, out)
    if world > 1:
        guarded_legs(printer, dist, lambda: node_vu_legs(cm, shard, dist, rank, world, local_rank, rehearsal, NB))
    sys.stdout.flush()
    printer.emit()
    if out.get("parity_gate", {}).get("ok") is False:
        sys.exit("bench.py: the parity gate failed -- the numbers above are not results of the reference's arithmetic")


def kernel_only(batch, frames, warm=100, timed=100):
    """average kernel time (ms, HIP events) of `timed` launches after `warm` untimed ones; closes the batch"""
    for _ in range(warm):
        batch.run(frames)
    batch.sync()
    batch.timing(True)
    batch.timing_read()
    for _ in range(timed):
        batch.run(frames)
    ms, n = batch.timing_read()
    batch.close()
    return ms / n


def place_off_leg(make_batch, T):
    """what the workload's kernel takes in this process WITHOUT the placement search: a batch made the default
    way first (the process's first two large allocations), warmed up, timed over 64 launches, freed"""
    t_c = time.perf_counter()
    a0 = make_batch()
    res = {"batch_create_ms_place_off": round((time.perf_counter() - t_c) * 1e3, 1)}
    t_w0 = time.perf_counter()
    while time.perf_counter() - t_w0 < MIN_WARMUP_S:
        for _ in range(16):
            a0.run(T)
        a0.sync()
    res["kernel_avg_ms_place_off"] = round(kernel_only(a0, T, warm=0, timed=64), 4)
    return res


def traffic_from_profiles(workload, S, T):
    """HBM bytes per launch are NOT measured in a bench run -- PMC counters need rocprofv3 passes of their own
    (tools/hbm_pmc.sh: FETCH_SIZE and WRITE_SIZE separately, FETCH_SIZE doubled for gfx950): the figure is read
    from the committed summary of such a run on the same workload, and labelled"""
    pmc_path = os.path.join(ROOT, "profiles", "pmc_%s.json" % workload)
    try:
        pmc = json.load(open(pmc_path))
        if pmc.get("workload") == workload and pmc.get("frames") == T and pmc.get("streams") == S:
            return pmc.get("hbm_bytes_per_launch"), (
                "profiles/pmc_%s.json (%s; separate rocprofv3 --pmc passes, not this run)" % (
                    workload, pmc.get("round", "round 1")))
    except Exception:
        pass
    return None, None


def parity_gate(cm, b, workload, S, Cn, T, first_global, global_step):
    """One block of the benchmarked batch against the CPU oracle, bit for bit, on a sample of its streams: the
    int16 PCM and the VU window of that block (configs 2, 4, 5: integer gain, channel map, first-max peak, sum of
    squares, dB doubles), or the float planes of the equaliser from cleared filter state (config 3).  The oracle
    is test infrastructure; it checks here, it is never the thing measured."""
    import numpy as np
    from oracle import oracle_ffi
    orc = oracle_ffi.Oracle()
    pick = sorted({0, 1, S // 2, S - 1})
    gate = {"streams_checked": pick, "frames": T, "against": "oracle/ (scalar C restatement of the reference's loops)"}
    try:
        if workload == "c3":
            b.eq_reset(-1)
            b.run(T)
            b.sync()
            coef = cm.eq3(48000.0)
            q = (oracle_ffi.Biquad * 3)()
            for i in range(3):
                q[i].b0, q[i].b1, q[i].b2, q[i].a1, q[i].a2 = [float(v) for v in coef[5 * i:5 * i + 5]]
            _, g = orc.gain(1, 1, 1000, [900])
            if os.environ.get("COOLMIC_BENCH_GAIN", "1") == "0":
                g = None
            ok = True
            for s_ in pick:
                wf, _ = orc.eq_run_mono(g, q, 3, np.zeros(12, dtype=np.float32),
                                        orc.lcg(12345 + first_global + s_ * global_step, T))
                ok = ok and np.array_equal(b.download_f32(s_, 0, T).view(np.uint32), wf.view(np.uint32))
            gate["float_planes_bit_equal"] = bool(ok)
            gate["ok"] = bool(ok)
            return gate
        b.vu_reset(-1)
        b.run(T)
        res, rcs = b.vu_results()
        gains = [750, 1250] if workload == "c2" else [900]
        _, g = orc.gain(Cn, len(gains), 1000, gains)
        if workload != "c2" and os.environ.get("COOLMIC_BENCH_GAIN", "1") == "0":
            g = oracle_ffi.Gain()                # (the bench ran with the gain disabled)
        pcm_ok = vu_ok = True
        for s_ in pick:
            x = orc.lcg(12345 + first_global + s_ * global_step, T * Cn)
            if workload == "c2":
                x = orc.chmap([1, 0], x, Cn)
            want = orc.gain_apply(g, x, Cn)
            pcm_ok = pcm_ok and np.array_equal(b.download(s_, T), want)
            v = orc.vu_new(Cn)
            orc.vu_accumulate(v, want)
            _, r = orc.vu_result(v)
            vu_ok = vu_ok and rcs[s_] == 0 and res[s_].as_dict() == oracle_ffi.vu_result_dict(r)
        gate.update({"pcm_byte_equal": bool(pcm_ok), "vu_results_bit_equal": bool(vu_ok), "ok": bool(pcm_ok and vu_ok)})
    except Exception as e:
        gate.update({"ok": False, "error": "%s: %s" % (type(e).__name__, e)})
    return gate


def measured_ceilings(b, achieved):
    """the kernel against the ceilings measured on the same buffers as well as against the nominal
    peak (PCM materialised: the copy with the same access shape; read-only runs: the read ceiling)"""
    extras = {}
    try:
        extras["hbm_read_ceiling_GBs"] = round(b.ceiling(0, 10), 1)
        extras["hbm_copy_ceiling_GBs"] = round(b.ceiling(1, 10), 1)
        if extras["hbm_copy_ceiling_GBs"] > 0:
            extras["kernel_frac_of_copy_ceiling"] = round(achieved / extras["hbm_copy_ceiling_GBs"], 4)
    except Exception as e:           # measurement extras must not break the line
        extras["ceiling_error"] = str(e)
    return extras


def small_blocks(b, S, Cn, T, bps, results, rcs):
    """the small-block regime, same batch, fewer frames per launch.  Per block size the kernel alone,
    the whole STEP with a VU window per block (launch + packed snapshot + host dB finish of all windows), and the
    step when windows close every 20 blocks -- the reference's own granularity (a result every 20 reads,
    ref: src/simple.c:370)"""
    def loop(frames, every, nsteps):
        # the dB finish of window k-1 runs on the helper threads beside launch and snapshot of block k+1
        # (cmhip_batch_vu_collect_begin / _end); up to three snapshots are pending
        collecting, waiting = False, 0
        for i in range(nsteps):
            b.run(frames)
            if i % every != every - 1:
                continue
            b.vu_snapshot()
            waiting += 1
            if collecting:
                b.vu_collect_end()
                collecting = False
                waiting -= 1
            if waiting >= 2:
                b.vu_collect_begin(results, rcs)
                collecting = True
        if collecting:
            b.vu_collect_end()
            waiting -= 1
        while waiting:
            b.vu_collect(results, rcs)
            waiting -= 1
        b.sync()

    sweep = {}
    try:
        for frames in (512, 2880, 4096):
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
            entry = {"kernel_avg_ms": round(ms / n, 4),
                     "achieved_GBs": round(S * Cn * frames * bps / (ms / n * 1e-3) / 1e9, 1)}
            for every, key in ((1, "step_ms_window_per_block"), (20, "step_ms_window_per_20_blocks")):
                b.vu_reset(-1)
                b.sync()
                loop(frames, every, 200)
                t1 = time.perf_counter()
                loop(frames, every, 1000)
                entry[key] = round((time.perf_counter() - t1) / 1000 * 1e3, 4)
            sweep[str(frames)] = entry
    except Exception as e:
        sweep["error"] = str(e)
    return sweep


def vu_only_lines(cm, workload, S, Cn, T, device, rank, world, read_ceiling):
    """second line of SURVEY 8(d): VU only, 2 B/sample read -- never mixed with `value`: with the workload's
    gain, with the transform as the reference creates it (gain disabled, ref: src/transform.c:107-108), and with
    every gain below the scale (shorter arithmetic)"""
    samples = S * Cn * T
    res = {}
    for key, g in (("vu_only", [750, 1250] if workload == "c2" else [900]), ("vu_only_gain_disabled", None),
                   ("vu_only_gains_below_scale", [900, 800][:Cn])):
        v = cm.Batch(S, Cn, T, flags=cm.VU, device=device)
        if g is not None:
            v.set_gain(-1, len(g), 1000, g)
        if workload == "c2":
            v.set_chmap(-1, [1, 0])
        v.generate(cm.GEN_NOISE, 12345, T, first_global=rank, global_step=world)
        ms1 = kernel_only(v, T)
        gbs = samples * 2 / (ms1 * 1e-3) / 1e9
        res[key] = {"kernel_avg_ms": round(ms1, 4), "achieved_GBs": round(gbs, 1),
                    "frac_of_peak": round(gbs / HBM_PEAK_GBS, 4)}
    res["vu_only"].update({"frac_of_read_ceiling": round(res["vu_only"]["achieved_GBs"] / read_ceiling, 4)
                           if read_ceiling > 0 else None,
                           "Msamples_per_s_kernel": round(samples / (res["vu_only"]["kernel_avg_ms"] * 1e-3) / 1e6, 1),
                           "algorithmic_bytes_per_sample": 2})
    return res


def other_kernels(cm, device):
    """the other kernels of the path, kernel time only: never part of `value`"""
    other = {}
    try:
        for name, (s_, c_, t_, fl, bps_, eqz) in {
                "c3_eq_float_planes": (8192, 1, 65536, cm.EQ | cm.OUT_F32, 6, True),
                "eq_stereo_int16_vu": (4096, 2, 65536, cm.EQ | cm.OUT_PCM | cm.VU, 4, True),
                "six_channels_pcm_vu": (2730, 6, 16384, cm.OUT_PCM | cm.VU, 4, False),
                "six_channels_vu_only": (2730, 6, 16384, cm.VU, 2, False)}.items():
            o = cm.Batch(s_, c_, t_, flags=fl, device=device)
            o.set_gain(-1, 1, 1000, [900])
            if eqz:
                o.set_eq(-1, cm.eq3(48000.0))
            o.generate(cm.GEN_NOISE, 12345, t_)
            ms1 = kernel_only(o, t_)             # ~0.1 s of warm-up: the clocks the chip then holds
            gbs = s_ * c_ * t_ * bps_ / (ms1 * 1e-3) / 1e9
            other[name] = {"streams": s_, "channels": c_, "frames": t_, "kernel_avg_ms": round(ms1, 4),
                           "algorithmic_bytes_per_sample": bps_, "achieved_GBs": round(gbs, 1),
                           "frac_of_peak": round(gbs / HBM_PEAK_GBS, 4)}
    except Exception as e:
        other["error"] = str(e)
    return other


class LinePrinter:
    """rank 0's one JSON line, written exactly once -- by the main thread at the end of the run or by the
    watchdog of guarded_legs() -- and never by another rank"""

    def __init__(self, rank, json_fd, out):
        import threading
        self.rank, self.fd, self.out = rank, json_fd, out
        self.lock = threading.Lock()
        self.done = False

    def emit(self):
        with self.lock:
            if self.done:
                return
            self.done = True
            if self.rank == 0:
                os.write(self.fd, (json.dumps(self.out) + "\n").encode())
            os.close(self.fd)


def guarded_legs(printer, dist, legs):
    """The config-4 / config-5 legs hold the run's only data-path collective.  If a rank fails in there while its
    peers wait inside a collective, nothing in the process would ever end the wait (gloo gives up after half an
    hour, RCCL never): so every rank arms a watchdog around the legs and the closing barrier
    ($COOLMIC_BENCH_LEGS_TIMEOUT_S, default 300).  When it fires, rank 0 writes its line -- complete but for
    `node_vu`, which says what happened -- and every rank leaves the process."""
    import threading
    out = printer.out
    limit = float(os.environ.get("COOLMIC_BENCH_LEGS_TIMEOUT_S", "300"))
    dist.barrier()                             # (rank 0 has been busy with the extras of its line until here)

    def fire():
        nv = out.get("node_vu")
        if nv and "rccl_ranks" in nv:          # this rank's legs were through: a peer never reached the closing barrier
            nv.setdefault("note", "a rank did not reach the closing barrier within %.0f s" % limit)
        elif not nv or "error" not in nv:
            out["node_vu"] = {"error": "the config-4 / config-5 legs did not finish within %.0f s on rank %d "
                                       "(a rank failed or a collective never completed); value and roofline above "
                                       "are unaffected" % (limit, printer.rank)}
        printer.emit()
        os._exit(0)

    dog = threading.Timer(limit, fire)
    dog.daemon = True
    dog.start()
    try:
        out["node_vu"] = legs()
    except SystemExit:
        raise
    except Exception as e:
        out["node_vu"] = {"error": "%s: %s" % (type(e).__name__, e)}
    dist.barrier()
    dog.cancel()
    dist.destroy_process_group()


NODE_LEG_SHAPE = (8192, 1, 65536)      # configs 4 / 5 per GPU: 65 536 mono streams round-robin over 8 GPUs
NODE_VU_KEYS = ("rccl_ranks", "steps", "blocks_per_exchange", "shape_per_gpu", "ms_per_step_c4", "ms_per_step_c5",
                "Msamples_s_c4", "Msamples_s_c5", "matches_host_merge", "check")


def node_vu_legs(cm, shard, dist, rank, world, local_rank, rehearsal, NB, steps=None, warm=None):
    """Configs 4 and 5 on all ranks of this run, one batch of the config-4 shape per rank:
    leg c4 -- launch, window snapshot, host dB finish per step, no exchange; leg c5 -- the same plus the
    block's node record and, per NB blocks, ONE pair of RCCL all-reduces over the records
    (cmhip_node_allreduce).  Every rank runs the same fixed number of steps, so every rank issues the same
    number of collectives.  Then the parity check of the RCCL path: the combined record of the last block
    against cmhip_node_merge_host() of the ranks' un-reduced records of that block, gathered over gloo
    (the "replicas only" form of SURVEY 8e)."""
    import numpy as np
    import torch
    S, Cn, T = NODE_LEG_SHAPE
    if os.environ.get("COOLMIC_BENCH_NODE_SHAPE"):           # (tests on small boxes)
        S, Cn, T = (int(v) for v in os.environ["COOLMIC_BENCH_NODE_SHAPE"].split(","))
    steps = steps or max(64, int(os.environ.get("COOLMIC_BENCH_NODE_STEPS", "128")))
    steps -= steps % NB                                      # whole sets: the last block's set is full
    warm = warm if warm is not None else 4 * NB
    n_local, first_global, global_step = shard.shard(S * world, world, rank)
    b = cm.Batch(S, Cn, T, flags=cm.OUT_PCM | cm.VU, device=local_rank)
    assert b.set_gain(-1, 1, 1000, [900]) == 0
    b.generate(cm.GEN_NOISE, 12345, T, first_global=first_global, global_step=global_step)
    b.sync()
    if rehearsal:
        node = cm.Node(local_rank, 1, 0, cm.node_unique_id(), max_records=NB)
    else:
        # (the id travels over gloo here: torch is loaded by now, and librccl was resolved -- next to the
        # HIP runtime the engine runs on -- before that, when the engine made the id)
        uid = torch.zeros(cm.NODE_ID_BYTES, dtype=torch.uint8)
        if rank == 0:
            uid = torch.tensor(list(cm.node_unique_id()), dtype=torch.uint8)
        dist.broadcast(uid, src=0)
        node = cm.Node(local_rank, world, rank, bytes(uid.tolist()), max_records=NB)
    results = (cm.VuResult * S)()
    rcs = (C.c_int * S)()
    own = [None]                                 # this rank's un-reduced record of the last block

    def sync_all():
        cm.device_synchronize(local_rank)
        dist.barrier()
        cm.device_synchronize(local_rank)

    def run(n, with_node, keep_last=False):
        pending = False
        for i in range(n):
            b.run(T)
            if with_node:
                k, slot = (i // NB) & 1, i % NB
                if keep_last and i == n - 1:
                    own[0] = b.node_record(first_global=first_global, global_step=global_step)
                node.partial(b, k, slot, first_global=first_global, global_step=global_step)
                if slot == NB - 1:
                    node.allreduce(k, NB, after=b)
            b.vu_snapshot()
            if pending:
                b.vu_collect(results, rcs)
            pending = True
        if pending:
            b.vu_collect(results, rcs)
        b.sync()

    def timed(with_node, keep_last=False):
        run(warm, with_node)
        sync_all()
        t0 = time.perf_counter()
        run(steps, with_node, keep_last)
        sync_all()
        return shard.max_over_ranks(dist, time.perf_counter() - t0, device="cpu") / steps * 1e3

    ms_c4 = timed(False)
    ms_c5 = timed(True, keep_last=True)
    k_last = ((steps - 1) // NB) & 1
    combined = node.fetch(k_last, NB)[NB - 1]
    mine = torch.from_numpy(own[0].copy())
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    records = np.stack([p.numpy() for p in parts])
    merged = cm.node_merge_host(records)
    # what the communicator's ranks should have produced together (rehearsal: one-rank communicators)
    expect = cm.node_merge_host(records[rank:rank + 1]) if rehearsal else merged
    ok = torch.tensor([1 if np.array_equal(expect, combined) else 0], dtype=torch.int32)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    rc, r = cm.node_finish(merged, Cn)
    out = {"rccl_ranks": node.ranks(), "steps": steps, "blocks_per_exchange": NB,
           "shape_per_gpu": "%d x %d ch x %d frames, gain 900/1000, PCM + VU" % (S, Cn, T),
           "ms_per_step_c4": round(ms_c4, 4), "ms_per_step_c5": round(ms_c5, 4),
           "Msamples_s_c4": round(S * Cn * T * world / (ms_c4 * 1e-3) / 1e6, 1),
           "Msamples_s_c5": round(S * Cn * T * world / (ms_c5 * 1e-3) / 1e6, 1),
           "matches_host_merge": bool(ok.item()),
           "check": "RCCL-combined record of the last block == cmhip_node_merge_host of the %s un-reduced "
                    "records gathered over gloo, on every rank" % ("rank's own" if rehearsal else "ranks'")}
    assert set(NODE_VU_KEYS) <= set(out)
    if rehearsal:
        out["rehearsal"] = "one-rank communicators (RCCL refuses two ranks on one GPU): rccl_ranks is 1"
    if rc == 0:
        out["last_block"] = {"frames": r.frames, "global_peak": r.global_peak, "global_power_db": r.global_power}
    node.close()
    b.close()
    return out


def dry_run(args, rank, world, json_fd, shard):
    if os.environ.get("COOLMIC_BENCH_DRYRUN_FAIL_RANK") == str(rank):      # (test hook: a rank that dies early)
        sys.exit(3)
    if os.environ.get("COOLMIC_BENCH_DRYRUN_HANG_RANK") == str(rank):      # (test hook: a rank that never gets there)
        time.sleep(3600)
    uid = exchange_node_id(rank, world, lambda: os.urandom(128)) if args.workload == "c5" else b""
    out = {"metric": "Msamples/s transform->vumeter", "value": 0.0, "unit": "Msamples/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "dry_run": True,
           "config": {"workload": args.workload}}
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
    # the warm-up loop of the real run with steps that take rank-dependent time: left to its own clock every
    # rank would stop after a different chunk; the decision is rank 0's, so the counts are equal
    t_w = time.perf_counter()
    warm_steps = warm_up(lambda n: time.sleep(n * 0.002 * (1 + 2 * rank)), args.warmup, 8,
                         collective_more(dist, rank, world, t_w, MIN_WARMUP_S))
    out["warmup_steps_effective"] = warm_steps
    if world > 1:
        out["clock_max_over_ranks"] = shard.max_over_ranks(dist, 1.0 + rank, device="cpu")
        ids = [torch.zeros(128, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(ids, torch.tensor(list(uid.ljust(128, b"\0")), dtype=torch.uint8))
        out["node_id_same_on_all_ranks"] = all(bool((i == ids[0]).all()) for i in ids)
        ranks = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(ranks, torch.tensor([rank, warm_steps]))
        out["ranks_seen"] = [int(r[0].item()) for r in ranks]
        out["warmup_steps_all_ranks"] = [int(r[1].item()) for r in ranks]

        # the keys the real run's config-4 / config-5 legs report (node_vu_legs), no GPU work behind them here;
        # through the same guard (test hook: a rank that never comes back from its legs)
        def legs():
            if os.environ.get("COOLMIC_BENCH_DRYRUN_LEGS_HANG_RANK") == str(rank):
                time.sleep(3600)
            flag = torch.tensor([rank], dtype=torch.int32)
            dist.all_reduce(flag)              # (peers of a hanging rank wait here, as in a real collective)
            d = dict.fromkeys(NODE_VU_KEYS)
            d["rccl_ranks"] = 0
            return d
        printer = LinePrinter(rank, json_fd, out)
        guarded_legs(printer, dist, legs)
        printer.emit()
        return
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)


def pcie_inclusive(cm, device):
    """Throughput with the PCM starting and ending in HOST memory (reported
    separately, never as `value`): (a) pinned host buffers, whole-batch upload -> fused kernel ->
    download, two batches in flight so copies and kernels overlap; (b) the slots themselves in
    pinned device-mapped host memory (CMHIP_HOSTPCM): the kernel reads and writes over PCIe."""
    import numpy as np
    S, Cn, T = 4096, 2, 16384                   # 256 MiB in + 256 MiB out per block
    res = {"workload": "config 2 shape, %d x %d x %d per block" % (S, Cn, T), "unit": "Msamples/s"}
    bs, hin, hout = [], [], []
    for _ in range(2):
        bb = cm.Batch(S, Cn, T, flags=cm.OUT_PCM | cm.VU, device=device)
        bb.set_gain(-1, 2, 1000, [750, 1250])
        bb.set_chmap(-1, [1, 0])
        bs.append(bb)
        hin.append(cm.PinnedPcm(bb))
        hout.append(cm.PinnedPcm(bb))
    rng = np.random.default_rng(1)
    blk = rng.integers(-32768, 32768, size=hin[0].shape[1], dtype=np.int64).astype(np.int16)
    for h in hin:
        h.array[:] = blk

    def step(i):
        bb = bs[i & 1]
        bb.sync()
        bb.upload_all(hin[i & 1].ptr, T)
        bb.run(T)
        bb.download_all(hout[i & 1].ptr, T)

    for i in range(4):
        step(i)
    for bb in bs:
        bb.sync()
    steps = 10
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    for bb in bs:
        bb.sync()
    dt = time.perf_counter() - t0
    n = S * Cn * T * steps
    res["copy_engines_two_batches_in_flight"] = {"value": round(n / dt / 1e6, 1),
                                                 "GBs_each_direction": round(n * 2 / dt / 1e9, 1),
                                                 "ms_per_block": round(dt / steps * 1e3, 3)}
    for h in hin + hout:
        h.free()
    for bb in bs:
        bb.close()
    z = cm.Batch(S, Cn, T, flags=cm.OUT_PCM | cm.VU | cm.HOSTPCM, device=device)
    z.set_gain(-1, 2, 1000, [750, 1250])
    z.set_chmap(-1, [1, 0])
    for s in range(0, S, 256):
        z.upload(s, blk[:T * Cn])
    for _ in range(2):
        z.run(T)
    z.sync()
    steps = 6
    t0 = time.perf_counter()
    for _ in range(steps):
        z.run(T)
    z.sync()
    dt = time.perf_counter() - t0
    z.close()
    n = S * Cn * T * steps
    res["zero_copy_slots_in_host_memory"] = {"value": round(n / dt / 1e6, 1),
                                             "GBs_each_direction": round(n * 2 / dt / 1e9, 1),
                                             "ms_per_block": round(dt / steps * 1e3, 3)}
    return res


def cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max), or None when unlimited / unknown"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else round(int(q) / int(per), 2)
    except Exception:
        return None


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
    quota = cpu_quota()                  # a container's CPU-time quota may be far below the threads it can see
    return {
        "value": round(n_all / secs / 1e6, 1), "unit": "Msamples/s", "cores": cores, "cpu_quota_cpus": quota,
        "kind": "port",
        "sample": "%d of the workload's streams (%d per thread) x %d frames x %d ch, same "
                  "generator and parameters, block-at-once" % (streams, per_thread, frames, channels),
        "one_thread_Msamples_s": round(n_one / secs1 / 1e6, 1),
        "pull_chain_1024B_one_thread_Msamples_s": round(chain_frames / secs_chain / 1e6, 1),
        "cpu_model": model,
    }


if __name__ == "__main__":
    main()
