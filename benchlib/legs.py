"""The measurements bench.py reports beside `value` -- never part of it: the parity gate against the oracle, the
ceilings measured on the benchmarked buffers, the small-block sweep, the VU-only forms, the other kernels of the
path, the PCIe-inclusive rates, the placement-search leg, and configs 4 / 5 on all ranks of a multi-GPU run."""
import ctypes as C
import json
import os
import sys
import time

from .launch import NODE_VU_KEYS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MIN_WARMUP_S = 0.15


def kernel_only(batch, frames, warm=100, timed=100):
    """average kernel time (ms, HIP events) of `timed` launches after at least `warm` untimed ones AND at least
    MIN_WARMUP_S of load (a batch that has just been created finds the chip at idle clocks: round 3 timed the VU-only
    forms after 17 ms of warm-up and read them 6 % slow); closes the batch"""
    t0 = time.perf_counter()
    done = 0
    while done < warm or time.perf_counter() - t0 < MIN_WARMUP_S:
        for _ in range(16):
            batch.run(frames)
        batch.sync()
        done += 16
    batch.timing(True)
    batch.timing_read()
    for _ in range(timed):
        batch.run(frames)
    ms, n = batch.timing_read()
    batch.close()
    return ms / n


def place_search_leg(cm, make_batch, T):
    """What the engine's OPT-IN placement search (CMHIP_PLACE_SEARCH) would buy on this box: a second batch
    of the workload created with the flag, after the benchmarked one is gone, its kernel timed the same way.  Never
    `value`, never `roofline`: those come from a batch created the library's default way."""
    res = {}
    try:
        t_c = time.perf_counter()
        b = make_batch(cm.PLACE_SEARCH)
        res["batch_create_ms_place_search"] = round((time.perf_counter() - t_c) * 1e3, 1)
        res["placement"] = b.placement()
        res["kernel_avg_ms_place_search"] = round(kernel_only(b, T, warm=64, timed=64), 4)
    except Exception as e:           # measurement extras must not break the line
        res["error"] = "%s: %s" % (type(e).__name__, e)
    return res


def traffic_from_profiles(workload, S, T, gain_form="general"):
    """HBM bytes per launch are NOT measured in a bench run -- PMC counters need rocprofv3 passes of their own
    (tools/hbm_pmc.sh: FETCH_SIZE and WRITE_SIZE separately, FETCH_SIZE doubled for gfx950): the figure is read
    from the committed summary of such a run on the same workload, and labelled"""
    pmc_path = os.path.join(ROOT, "profiles", "pmc_%s.json" % workload)
    try:
        pmc = json.load(open(pmc_path))
        if pmc.get("workload") == workload and pmc.get("frames") == T and pmc.get("streams") == S and \
                pmc.get("gain_form", "general") == gain_form:
            return pmc.get("hbm_bytes_per_launch"), (
                "profiles/pmc_%s.json (%s; separate rocprofv3 --pmc passes, not this run)" % (
                    workload, pmc.get("round", "round 1")))
    except Exception:
        pass
    return None, None


def parity_gate(cm, b, workload, S, Cn, T, first_global, global_step, gains, swap, has_pcm=True):
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
            g = orc.gain(1, len(gains), 1000, gains)[1] if gains else None
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
        # (a disabled gain is the reference's scale 0: the oracle's zeroed Gain)
        g = orc.gain(Cn, len(gains), 1000, gains)[1] if gains else oracle_ffi.Gain()
        pcm_ok = vu_ok = True
        for s_ in pick:
            x = orc.lcg(12345 + first_global + s_ * global_step, T * Cn)
            if swap:
                x = orc.chmap([1, 0], x, Cn)
            want = orc.gain_apply(g, x, Cn)
            if has_pcm:
                pcm_ok = pcm_ok and np.array_equal(b.download(s_, T), want)
            v = orc.vu_new(Cn)
            orc.vu_accumulate(v, want)
            _, r = orc.vu_result(v)
            vu_ok = vu_ok and rcs[s_] == 0 and res[s_].as_dict() == oracle_ffi.vu_result_dict(r)
        gate.update({"pcm_byte_equal": bool(pcm_ok) if has_pcm else None, "vu_results_bit_equal": bool(vu_ok),
                     "ok": bool(pcm_ok and vu_ok)})
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


GAIN_FORMS = ("general", "below", "off")


def workload_gains(workload, channels, form="general"):
    """-> (gains or None, channel swap): the workload's own parameters for "general"; "below": every gain
    below the scale (the one-mulhi form of the read-only runs); "off": the transform as the reference creates it
    (gain disabled, ref: src/transform.c:107-108)"""
    swap = workload in ("c2", "c2ro")
    if form == "off":
        return None, swap
    if form == "below":
        return [900, 800][:channels], swap
    return ([750, 1250] if channels == 2 else [900]), swap


def vu_only_lines(cm, workload, S, Cn, T, device, rank, world, read_ceiling):
    """second line of SURVEY 8(d): VU only, 2 B/sample read -- never mixed with `value`: with the workload's
    gain, with the transform as the reference creates it (gain disabled), and with every gain below the scale
    (shorter arithmetic).  Each batch is warmed up for MIN_WARMUP_S before its kernel is timed.  `traffic` of the
    first comes from the committed PMC summary of `--workload c2ro` (same kernel, same shape)."""
    samples = S * Cn * T
    res = {}
    for key, form in (("vu_only", "general"), ("vu_only_gain_disabled", "off"), ("vu_only_gains_below_scale", "below")):
        g, swap = workload_gains(workload, Cn, form)
        v = cm.Batch(S, Cn, T, flags=cm.VU, device=device)
        if g is not None:
            v.set_gain(-1, len(g), 1000, g)
        if swap:
            v.set_chmap(-1, [1, 0])
        v.generate(cm.GEN_NOISE, 12345, T, first_global=rank, global_step=world)
        ms1 = kernel_only(v, T)
        gbs = samples * 2 / (ms1 * 1e-3) / 1e9
        res[key] = {"kernel_avg_ms": round(ms1, 4), "achieved_GBs": round(gbs, 1),
                    "frac_of_peak": round(gbs / HBM_PEAK_GBS, 4)}
    res["vu_only"].update({"frac_of_read_ceiling": round(res["vu_only"]["achieved_GBs"] / read_ceiling, 4)
                           if read_ceiling > 0 else None,
                           "Msamples_per_s_kernel": round(samples / (res["vu_only"]["kernel_avg_ms"] * 1e-3) / 1e6, 1),
                           "algorithmic_bytes_per_sample": 2, "algorithmic_bytes_per_launch": samples * 2})
    if workload == "c2":
        traffic, source = traffic_from_profiles("c2ro", S, T)
        res["vu_only"].update({"traffic": traffic, "traffic_source": source})
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


NODE_LEG_SHAPE = (8192, 1, 65536)      # configs 4 / 5 per GPU: 65 536 mono streams round-robin over 8 GPUs


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
