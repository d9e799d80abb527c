"""bench.py's process plumbing: starting the rank processes, the node-id exchange, one warm-up decision for all
ranks, rank 0's single JSON line and the watchdog around the legs that hold a collective, the GPU-less dry run the
CPU tests drive.  Nothing here touches HIP."""
import json
import os
import socket
import subprocess
import sys
import time

MIN_WARMUP_S = 0.15            # the chip reaches the clocks it then holds after ~100 ms of load
BENCH_PY = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
# the keys of `node_vu` (legs.node_vu_legs; the dry run reports the same keys without GPU work behind them)
NODE_VU_KEYS = ("rccl_ranks", "steps", "blocks_per_exchange", "shape_per_gpu", "ms_per_step_c4", "ms_per_step_c5",
                "Msamples_s_c4", "Msamples_s_c5", "matches_host_merge", "check")


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
        procs.append(subprocess.Popen([sys.executable, BENCH_PY] + sys.argv[1:], env=env,
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
