"""CPU, world_size 2 over gloo: the N>1 path of bench.py -- round-robin sharding, the
max-over-ranks clock and the node-global VU combine (SUM / MAX all-reduce + host finish).
Per-rank node records are built here from the oracle's windows (the device kernel that
builds them on a GPU is covered by tests/test_gpu_parity.py::test_node_partial_matches_host_merge)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C, T, TOTAL = 2, 3000, 7          # 7 streams over 2 ranks: 4 + 3


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _stream_pcm(orc, gs):
    _, g = orc.gain(C, 2, 1000, [750, 1250])
    return orc.gain_apply(g, orc.lcg(777 + gs, T * C), C)


def _node_record(orc, streams):
    """what cmhip_batch_vu_node_partial writes for these global streams (see the key layout
    in include/coolmic_hip.h and csrc/k_misc.hip:k_node_partial)"""
    w = np.zeros(34, dtype=np.int64)
    for gs in streams:
        x = _stream_pcm(orc, gs).astype(np.int64).reshape(-1, C)
        w[16] += T
        for c in range(C):
            w[c] += int((x[:, c] ** 2).sum())
            mag = int(np.abs(x[:, c]).max())
            if mag == 0:
                continue
            fr = int(np.argmax(np.abs(x[:, c]) == mag))
            neg = 1 if x[fr, c] < 0 else 0
            key = (mag << 46) | ((0x1FFFFFFF - min(fr, 0x1FFFFFFF)) << 17) | ((65535 - gs % 65536) << 1) | neg
            w[17 + c] = max(w[17 + c], key)
            w[33] = max(w[33], key)
    return w


def _worker(rank, world, port, out):
    import sys
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    from oracle import oracle_ffi
    cm = ge.load_package()
    from libcoolmic_dsp_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = oracle_ffi.Oracle()
        n, first, step = shard.shard(TOTAL, world, rank)
        mine = [shard.global_id(s, world, rank) for s in range(n)]
        assert mine == list(range(first, TOTAL, step))
        # every stream has exactly one owner
        owned = torch.zeros(TOTAL, dtype=torch.int64)
        for gs in mine:
            owned[gs] = 1
            assert shard.owner(gs, world) == (rank, mine.index(gs))
        dist.all_reduce(owned)
        assert owned.tolist() == [1] * TOTAL
        # the clock: max over ranks
        assert shard.max_over_ranks(dist, 1.0 + rank) == float(world)
        # node-global VU
        words = torch.from_numpy(_node_record(orc, mine))
        shard.combine_node_records(dist, words)
        rc, r = cm.node_finish(words.numpy(), C)
        assert rc == 0
        # the batched form: several blocks' records in one all-gather, same result per block
        mine_rec = torch.from_numpy(_node_record(orc, mine))
        batch = torch.stack([mine_rec, torch.zeros_like(mine_rec), mine_rec])
        comb = shard.gather_node_records(dist, batch)
        assert torch.equal(comb[0], words) and torch.equal(comb[2], words)
        assert int(comb[1].abs().sum()) == 0
        if rank == 0:
            out.put(r.as_dict())
    finally:
        dist.destroy_process_group()


def test_sharding_and_node_vu_world2(oracle):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = q.get()
    # expected straight from the oracle over all streams
    pw = np.zeros(C, dtype=np.int64)
    best = [(0, 0, 0, 0)] * C
    for gs in range(TOTAL):
        x = _stream_pcm(oracle, gs).astype(np.int64).reshape(-1, C)
        for c in range(C):
            pw[c] += int((x[:, c] ** 2).sum())
            mag = int(np.abs(x[:, c]).max())
            fr = int(np.argmax(np.abs(x[:, c]) == mag))
            cand = (mag, -fr, -gs, int(x[fr, c]))
            if cand[:3] > best[c][:3]:
                best[c] = cand
    assert got["frames"] == TOTAL * T and got["channels"] == C
    for c in range(C):
        assert got["channel_power"][c] == oracle.lib.oracle_power_db(int(pw[c]), TOTAL * T)
        assert got["channel_peak"][c] == best[c][3]
    assert got["global_power"] == oracle.lib.oracle_power_db(int(pw.sum()), TOTAL * T * C)
    assert got["global_peak"] == max(best, key=lambda b: b[:3])[3]


def test_shard_edges():
    import __graft_entry__ as ge
    ge.load_package()
    from libcoolmic_dsp_amd import shard
    assert shard.shard(65536, 8, 3) == (8192, 3, 8)
    assert [shard.shard(10, 4, r)[0] for r in range(4)] == [3, 3, 2, 2]
    assert shard.shard(1, 8, 5)[0] == 0
    with pytest.raises(ValueError):
        shard.shard(8, 2, 2)
