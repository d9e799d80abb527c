"""T = 2880 frames: the packet length of the reference's Opus feeder (ref: src/enc_opus.c:197-251,
2880 frames per opus_encode at 48 kHz) is the natural block of an encoder-driven host
(SURVEY 8f-2).  2880 frames are 45 of the EQ kernel's 64-frame blocks, but a packet is never a
whole number of the block kernels' tiles (5760 / 11520 / 34560 bytes against 4, 8 and 16 KiB tiles
and 63- or 60-vector rows): the last tile of every stream is ragged.  Batch, group and EQ
paths, several packets with carried window / filter state."""
import numpy as np
import pytest

from oracle import oracle_ffi as of

pytestmark = pytest.mark.gpu
T = 2880


@pytest.mark.parametrize("C", [1, 2, 6])
def test_batch_blocks_of_one_opus_packet(gpu, oracle, C):
    cm = gpu
    S, blocks = 37, 3
    rng = np.random.default_rng(2880 + C)
    b = cm.Batch(S, C, T, flags=cm.OUT_PCM | cm.VU | cm.OUT_F32)
    gains = [[int(v) for v in rng.integers(200, 2500, C)] for _ in range(S)]
    cmaps = [[int(v) for v in rng.permutation(C)] if s % 3 == 0 else None for s in range(S)]
    for s in range(S):
        assert b.set_gain(s, C, 1000, gains[s]) == 0
        assert b.set_chmap(s, cmaps[s]) == 0
    raws = [oracle.lcg(555 + s, T * C * blocks) for s in range(S)]
    wants = [[] for _ in range(S)]
    for k in range(blocks):
        b.generate(cm.GEN_NOISE, 555, T, frame_offset=k * T)
        b.run(T)
        for s in range(S):
            x = raws[s][k * T * C:(k + 1) * T * C]
            _, g = oracle.gain(C, C, 1000, gains[s])
            want = oracle.gain_apply(g, oracle.chmap(cmaps[s], x, C) if cmaps[s] else x, C)
            wants[s].append(want)
            assert np.array_equal(b.download(s, T), want), (C, s, k)
            if s % 5 == 0:
                for c in range(C):
                    f = b.download_f32(s, c, T)
                    assert np.array_equal(f.view(np.uint32), oracle.to_f32_planar(want, C)[c].view(np.uint32)), (C, s, k, c)
    res, rcs = b.vu_results()                     # one window over the three packets
    for s in range(S):
        v = oracle.vu_new(C)
        for w in wants[s]:
            oracle.vu_accumulate(v, w)
        _, r = oracle.vu_result(v)
        assert rcs[s] == 0 and res[s].as_dict() == of.vu_result_dict(r), (C, s)
    b.close()


@pytest.mark.parametrize("C", [1, 2])
def test_eq_blocks_of_one_opus_packet(gpu, oracle, C):
    """EQ batch, int16 result + VU + float planes, three packets with carried filter state"""
    cm = gpu
    S, blocks = 70, 3                             # three workgroups on mono, five on stereo
    coef = cm.eq3(48000.0)
    b = cm.Batch(S, C, T, flags=cm.EQ | cm.OUT_PCM | cm.OUT_F32 | cm.VU)
    assert b.set_eq(-1, coef) == 0
    assert b.set_gain(-1, 1, 1000, [900]) == 0
    q = (of.Biquad * 3)()
    for i in range(3):
        q[i].b0, q[i].b1, q[i].b2, q[i].a1, q[i].a2 = [float(v) for v in coef[5 * i:5 * i + 5]]
    _, g = oracle.gain(1, 1, 1000, [900])
    states = [[np.zeros(12, dtype=np.float32) for _ in range(C)] for _ in range(S)]
    vus = [oracle.vu_new(C) for _ in range(S)]
    for k in range(blocks):
        b.generate(cm.GEN_NOISE, 4242, T, frame_offset=k * T)
        b.run(T)
        for s in range(0, S, 3):
            x = oracle.lcg(4242 + s, T * C * (k + 1))[k * T * C:]
            out = np.empty((T, C), dtype=np.int16)
            for c in range(C):
                wf, wi = oracle.eq_run_mono(g, q, 3, states[s][c], x.reshape(-1, C)[:, c].copy())
                out[:, c] = wi
                f = b.download_f32(s, c, T)
                assert np.array_equal(f.view(np.uint32), wf.view(np.uint32)), (C, s, k, c)
            assert np.array_equal(b.download(s, T), out.reshape(-1)), (C, s, k)
            oracle.vu_accumulate(vus[s], out.reshape(-1))
    res, rcs = b.vu_results()
    for s in range(0, S, 3):
        _, r = oracle.vu_result(vus[s])
        assert rcs[s] == 0 and res[s].as_dict() == of.vu_result_dict(r), (C, s)
    b.close()


def test_group_blocks_of_one_opus_packet(gpu, oracle):
    """coolmic_group_t with block_frames = 2880: readers pull exactly one packet's bytes at a
    time, as opus's feeder does (ref: src/enc_opus.c:215)"""
    cm = gpu
    C, N, packets = 2, 5, 7
    grp = cm.Group(C, 8, T, queue_blocks=2)
    xs, handles = [], []
    for i in range(N):
        frames = packets * T - (i * 100)          # the last packet of most streams is short
        x = oracle.lcg(9000 + i, frames * C)
        src = cm.IoHandle.from_bytes(x.tobytes(), chunk=[0, 1024, 7, 4096, 333][i])
        slot = grp.add_stream(src)
        src.unref()
        assert grp.set_master_gain(slot, C, 1000, [750, 1250]) == 0
        assert grp.set_channel_map(slot, [1, 0]) == 0
        xs.append(x)
        handles.append(grp.get_iohandle(slot))
    got = [b"" for _ in range(N)]
    active = set(range(N))
    guard = 0
    while active and guard < 10000:
        guard += 1
        for i in list(active):
            n, data = handles[i].read(T * C * 2)
            assert n >= 0 and n % (2 * C) == 0
            got[i] += data
            if n == 0 and handles[i].eof() == 1:
                active.discard(i)
    assert not active
    _, g = oracle.gain(C, C, 1000, [750, 1250])
    for i in range(N):
        want = oracle.gain_apply(g, oracle.chmap([1, 0], xs[i], C), C)
        assert np.array_equal(np.frombuffer(got[i], np.int16), want), i
        rc, r = grp.vumeter_result(i)
        v = oracle.vu_new(C)
        oracle.vu_accumulate(v, want)
        _, ro = oracle.vu_result(v)
        assert rc == 0 and r.as_dict() == of.vu_result_dict(ro), i
    for h in handles:
        h.unref()
    grp.unref()
