"""GPU parity: the HIP path (through the C ABI of include/coolmic_hip.h) against the CPU
oracle on the same seeded inputs, and against the SURVEY 8(c) golden vectors.

Bar: bit-exact for int16 PCM, VU accumulators and peaks; dB values bit-equal doubles
(they are finished on the host with the reference's own formula).
"""
import math
import os

import numpy as np
import pytest

from oracle import oracle_ffi as of

pytestmark = pytest.mark.gpu


def _pow(x):
    return -math.inf if x == "-inf" else float(x)


def _oracle_block(orc, pcm, C, gain_args, cmap):
    """reference result for one block: (pcm_out, vu_dict or None)"""
    x = np.asarray(pcm, dtype=np.int16)
    if cmap is not None:
        x = orc.chmap(cmap, x, C)
    rc, g = orc.gain(C, *gain_args) if gain_args is not None else (0, of.Gain())
    assert rc == 0
    y = orc.gain_apply(g, x, C)
    return y


def _oracle_vu(orc, blocks, C):
    v = orc.vu_new(C)
    for blk in blocks:
        orc.vu_accumulate(v, blk)
    rc, r = orc.vu_result(v)
    return rc, r


def _rand_pcm(rng, n, kind):
    if kind == "full":
        return rng.integers(-32768, 32768, size=n, dtype=np.int64).astype(np.int16)
    if kind == "edges":
        return rng.choice(np.array([-32768, -32767, -1, 0, 1, 32766, 32767], dtype=np.int16), size=n)
    if kind == "small":
        return rng.integers(-5, 6, size=n, dtype=np.int64).astype(np.int16)
    raise ValueError(kind)


# ---------------------------------------------------------------------------


def test_golden_known_answers_through_batch(gpu, golden):
    cm = gpu
    for name in ("K1", "K2", "K3", "K4", "K5", "K9"):
        case = golden["cases"][name]
        x = np.array(golden[case["input"]], dtype=np.int16)
        C = case["channels"]
        b = cm.Batch(1, C, 64, flags=cm.OUT_PCM | cm.VU)
        g = case["gain"]
        assert b.set_gain(0, g["channels"], g["scale"], g["gain"]) == 0
        b.upload(0, x)
        b.run(x.size // C)
        assert b.download(0, x.size // C).tolist() == case["pcm"], name
        rc, r = b.vu_result(0)
        assert rc == 0
        exp = case["vu"]
        if "frames" in exp:
            assert r.frames == exp["frames"]
        if "global_peak" in exp:
            assert r.global_peak == exp["global_peak"], name
        assert r.global_power == _pow(exp["global_power"]), name
        for i, p in enumerate(exp.get("channel_peak", [])):
            assert r.channel_peak[i] == p
        for i, p in enumerate(exp.get("channel_power", [])):
            assert r.channel_power[i] == _pow(p)
        # a second result without new frames is INVAL (ref: src/vumeter.c:198-199)
        rc2, _ = b.vu_result(0)
        assert rc2 == cm.ERROR_INVAL
        b.close()


def test_golden_K6_invalid_gain_shape(gpu, golden):
    cm = gpu
    case = golden["cases"]["K6"]
    b = cm.Batch(1, 2, 64, flags=cm.OUT_PCM)
    assert b.set_gain(0, 3, 1000, [1, 2, 3]) == case["set_gain_rc"]
    x = np.array(golden[case["input"]], dtype=np.int16)
    b.upload(0, x)
    b.run(4)
    assert b.download(0, 4).tolist() == case["pcm"]
    b.close()


def test_golden_G4_and_sine_G1_G2_G3(gpu, golden, oracle):
    cm = gpu
    # G4: stereo LCG, gains {750,1250}/1000, 48128 frames
    case = golden["cases"]["G4"]
    b = cm.Batch(1, 2, case["frames"], flags=cm.VU)
    assert b.set_gain(0, 2, 1000, [750, 1250]) == 0
    b.generate(cm.GEN_NOISE, case["seed"], case["frames"])
    b.run(case["frames"])
    rc, r = b.vu_result(0)
    exp = case["vu"]
    assert rc == 0 and r.frames == exp["frames"] and r.global_peak == exp["global_peak"]
    assert r.global_power == exp["global_power"]
    assert [r.channel_peak[i] for i in range(2)] == exp["channel_peak"]
    assert [r.channel_power[i] for i in range(2)] == exp["channel_power"]
    b.close()

    # G1..G3: the sine source's stream, cut where the reference chain cut it
    g1, g2, g3 = (golden["cases"][k] for k in ("G1", "G2", "G3"))
    n1 = g1["vu"]["frames"]
    b = cm.Batch(1, 1, n1, flags=cm.VU)
    b.generate(cm.GEN_SINE, 0, n1)
    assert b.set_gain(0, 1, 1000, [1000]) == 0
    b.run(n1)
    rc, r = b.vu_result(0)
    assert rc == 0 and r.frames == n1 and r.global_peak == g1["vu"]["global_peak"]
    assert r.global_power == g1["vu"]["global_power"] == r.channel_power[0]
    pos = n1
    for case, gain in ((g2, 500), (g3, 2000)):
        b.generate(cm.GEN_SINE, 0, 512, frame_offset=pos)
        assert b.set_gain(0, 1, 1000, [gain]) == 0
        b.run(512)
        rc, r = b.vu_result(0)
        assert rc == 0 and r.frames == 512
        assert r.global_peak == case["vu"]["global_peak"]
        assert r.global_power == case["vu"]["global_power"]
        pos += 512
    b.close()


def test_null_source_is_minus_inf(gpu, golden):
    cm = gpu
    b = cm.Batch(3, 2, 256, flags=cm.VU | cm.OUT_PCM)
    b.generate(cm.GEN_NULL, 0, 256)
    b.run(256)
    res, rcs = b.vu_results()
    for s in range(3):
        assert rcs[s] == 0 and res[s].frames == 256 and res[s].global_peak == 0
        assert res[s].global_power == -math.inf
        assert res[s].channel_power[0] == -math.inf and res[s].channel_power[1] == -math.inf
    b.close()


def test_generators_match_host_generators(gpu, oracle):
    cm = gpu
    S, T = 5, 1000
    for C in (1, 2, 3):
        b = cm.Batch(S, C, T, flags=cm.VU)
        b.generate(cm.GEN_NOISE, 12345, T, first_global=3, global_step=8, frame_offset=77)
        for s in range(S):
            gs = 3 + 8 * s
            full = oracle.lcg(12345 + gs, (77 + T) * C)
            assert np.array_equal(b.download_input(s, T), full[77 * C:]), (C, s)
        b.generate(cm.GEN_SINE, 0, T, first_global=1, global_step=2, frame_offset=5)
        rc, table = oracle.sine_table(48000)
        for s in range(S):
            gs = 1 + 2 * s
            want = np.repeat(table[(np.arange(T) + 5 + 7 * gs) % 48], C)
            assert np.array_equal(b.download_input(s, T), want), (C, s)
        b.close()


@pytest.mark.parametrize("C", [1, 2, 3, 4, 5, 6, 8, 11, 16])
def test_random_blocks_bit_exact(gpu, oracle, C):
    """ragged stream lengths, random gains/scales/maps, PCM + VU compared per stream"""
    cm = gpu
    rng = np.random.default_rng(1000 + C)
    lens = [0, 1, 2, 3, 7, 8, 9, 63, 64, 65, 255, 511, 1000, 4095, 4096, 4097, 5000]
    S = len(lens)
    T = max(lens)
    for inplace in (False, True):
        flags = cm.OUT_PCM | cm.VU | (cm.INPLACE if inplace else 0)
        b = cm.Batch(S, C, T, flags=flags)
        params = []
        for s in range(S):
            kind = ["full", "edges", "small"][s % 3]
            x = _rand_pcm(rng, lens[s] * C, kind)
            mode = s % 5
            if mode == 0:
                ga = None                                          # never set: disabled
            elif mode == 1:
                ga = (C, int(rng.integers(1, 65536)), [int(v) for v in rng.integers(0, 65536, C)])
            elif mode == 2:
                ga = (1, int(rng.choice([1, 2, 1000, 32768, 65535])), [int(rng.integers(0, 65536))])
            elif mode == 3:
                ga = (C, 1, [65535] * C)
            else:
                ga = (C, 65535, [int(v) for v in rng.integers(0, 65536, C)])
            cmap = None if s % 2 == 0 else [int(v) for v in rng.integers(0, C, C)]
            if ga is not None:
                assert b.set_gain(s, *ga) == 0
            if cmap is not None:
                assert b.set_chmap(s, cmap) == 0
            if lens[s]:
                b.upload(s, x)
            params.append((x, ga, cmap))
        b.run(T, frames_per_stream=lens)
        for s in range(S):
            x, ga, cmap = params[s]
            want = _oracle_block(oracle, x, C, ga, cmap)
            got = b.download(s, lens[s]) if lens[s] else np.zeros(0, np.int16)
            assert np.array_equal(got, want), (C, s, inplace)
            rc_o, r_o = _oracle_vu(oracle, [want], C)
            rc_g, r_g = b.vu_result(s)
            assert rc_g == rc_o, (C, s)
            if rc_o == 0:
                assert r_g.as_dict() == of.vu_result_dict(r_o), (C, s, inplace)
        b.close()


@pytest.mark.parametrize("C", [1, 2, 5])
def test_vu_only_and_float_outputs(gpu, oracle, C):
    cm = gpu
    rng = np.random.default_rng(50 + C)
    S, T = 6, 3001
    gains = [int(v) for v in rng.integers(100, 3000, C)]
    xs = [_rand_pcm(rng, T * C, "full") for _ in range(S)]
    wants = [_oracle_block(oracle, x, C, (C, 1000, gains), None) for x in xs]
    # VU only (no PCM is written anywhere)
    b = cm.Batch(S, C, T, flags=cm.VU)
    assert b.set_gain(-1, C, 1000, gains) == 0
    for s in range(S):
        b.upload(s, xs[s])
    b.run(T)
    for s in range(S):
        rc, r = b.vu_result(s)
        _, ro = _oracle_vu(oracle, [wants[s]], C)
        assert rc == 0 and r.as_dict() == of.vu_result_dict(ro)
    b.close()
    # planar float = transformed PCM / 32768.f (ref: src/enc_vorbis.c:108-115), exact
    for flags in (cm.OUT_F32, cm.OUT_F32 | cm.VU, cm.OUT_F32 | cm.OUT_PCM | cm.VU):
        b = cm.Batch(S, C, T, flags=flags)
        assert b.set_gain(-1, C, 1000, gains) == 0
        for s in range(S):
            b.upload(s, xs[s])
        b.run(T)
        for s in range(S):
            planar = oracle.to_f32_planar(wants[s], C)
            for c in range(C):
                got = b.download_f32(s, c, T)
                assert np.array_equal(got.view(np.uint32), planar[c].view(np.uint32)), (C, s, c)
        b.close()


@pytest.mark.parametrize("C", [1, 2])
def test_short_gain_forms_of_the_read_only_runs(gpu, oracle, C):
    """Runs with a VU window and no PCM result pick a shorter form of the gain arithmetic per stream
    (StreamParam::mode): no gain / unity -> the samples' own magnitudes; every gain below the scale ->
    one mulhi by ceil(gain * 2^32 / scale).  One batch mixes every case, on whole tiles and a ragged
    end, with the extreme samples and the extreme constants; windows and float planes against the oracle."""
    cm = gpu
    rng = np.random.default_rng(600 + C)
    T = 20011                                    # whole 16 KiB tiles and a ragged last one
    cases = [None,                                                  # gain disabled
             (C, 1000, [1000] * C),                                 # unity
             (C, 1000, [900, 999][:C]),                             # below the scale
             (C, 65535, [65534, 1][:C]),                            # extreme constants, below
             (C, 3, [2, 0][:C]),                                    # a zero gain
             (C, 32768, [32767, 16384][:C]),
             (C, 1000, [750, 1250][:C] if C == 2 else [1001]),      # one gain above: general form
             (1, 7, [6]),                                           # one value for every channel
             (C, 1000, [1000, 999][:C])]                            # unity on one channel only
    S = len(cases)
    xs = []
    for s in range(S):
        x = _rand_pcm(rng, T * C, "full" if s % 2 == 0 else "edges")
        x[:4] = [-32768, 32767, -32768, -1]
        xs.append(x)
    cmaps = [None, [1, 0]] if C == 2 else [None]
    for cmap in cmaps:
        wants = [_oracle_block(oracle, xs[s], C, cases[s], cmap) for s in range(S)]
        for flags in (cm.VU, cm.OUT_F32 | cm.VU):
            b = cm.Batch(S, C, T, flags=flags)
            if cmap is not None:
                assert b.set_chmap(-1, cmap) == 0
            for s in range(S):
                if cases[s] is not None:
                    assert b.set_gain(s, *cases[s]) == 0
                b.upload(s, xs[s])
            for lo, hi in ((0, T // 2 + 3), (T // 2 + 3, T)):       # two launches, one window
                if lo:
                    for s in range(S):
                        b.upload(s, xs[s][lo * C:])
                b.run(hi - lo)
                if flags & cm.OUT_F32:
                    for s in range(S):
                        planar = oracle.to_f32_planar(wants[s][lo * C:hi * C], C)
                        for c in range(C):
                            got = b.download_f32(s, c, hi - lo)
                            assert np.array_equal(got.view(np.uint32), planar[c].view(np.uint32)), (s, c)
            for s in range(S):
                rc, r = b.vu_result(s)
                _, ro = _oracle_vu(oracle, [wants[s]], C)
                assert rc == 0 and r.as_dict() == of.vu_result_dict(ro), (C, cmap, flags, s)
            b.close()


@pytest.mark.parametrize("nw", ["1", "4", "8"])
@pytest.mark.parametrize("C", [1, 2])
def test_workgroups_of_several_waves_merge_their_windows(gpu, oracle, C, nw, monkeypatch):
    """Runs that write PCM and keep a window take workgroups of four waves (a tile each) that add their
    window sums up in LDS; the last wave to finish adds the workgroup's to the stream's window
    (run_fast in k_block.hip; $CMHIP_FAST_NW picks 1, 4 or 8 waves).  Streams of every length around a
    workgroup's 4 x 4 KiB -- whole workgroups, ragged last ones with idle waves, a single tile, one
    frame, none -- with the peak in every wave's tile in turn, ties between waves (the first wins),
    two launches per window; PCM and windows against the oracle."""
    cm = gpu
    monkeypatch.setenv("CMHIP_FAST_NW", nw)
    rng = np.random.default_rng(800 + C)
    per_tile = 2048 // C                         # frames of a 4 KiB tile
    lens = [0, 1, per_tile - 1, per_tile, per_tile + 1, 4 * per_tile, 4 * per_tile + 3, 7 * per_tile - 5,
            8 * per_tile, 9 * per_tile + 1, 13 * per_tile + 77]
    T = max(lens)
    S = len(lens)
    gain = (C, 1000, [750, 1250][:C])
    cmap = [1, 0] if C == 2 else None
    xs = []
    for s, n in enumerate(lens):
        x = _rand_pcm(rng, T * C, "small")
        # the same extreme magnitude in several tiles, +/-: the earliest one is the window's peak
        for k, at in enumerate(range(s % 5, max(n, 1), max(per_tile * 2 // 3, 1))):
            x[(at * C + k % C) % (T * C)] = 32767 if k % 2 else -32767
        xs.append(x)
    b = cm.Batch(S, C, T, flags=cm.OUT_PCM | cm.VU)
    assert b.set_gain(-1, *gain) == 0
    if cmap:
        assert b.set_chmap(-1, cmap) == 0
    wants = [[], []]
    for half in range(2):                        # two launches, one window; the second with other lengths
        ls = lens if half == 0 else lens[::-1]
        for s in range(S):
            b.upload(s, xs[s] if half == 0 else xs[s][::-1].copy())
        b.run(T, frames_per_stream=ls)
        for s in range(S):
            src = (xs[s] if half == 0 else xs[s][::-1])[:ls[s] * C]
            want = _oracle_block(oracle, src, C, gain, cmap)
            wants[half].append(want)
            got = b.download(s, ls[s]) if ls[s] else np.zeros(0, np.int16)
            assert np.array_equal(got, want), (C, nw, half, s)
    for s in range(S):
        rc, r = b.vu_result(s)
        blocks = [w for w in (wants[0][s], wants[1][s]) if len(w)]
        _, ro = _oracle_vu(oracle, blocks, C)
        assert rc == 0 and r.as_dict() == of.vu_result_dict(ro), (C, nw, s)
    b.close()


@pytest.mark.parametrize("C", [3, 4, 6, 8, 13, 16])
def test_many_channel_read_only_runs_without_gain(gpu, oracle, C):
    """A batch in which no stream has a gain (the transform as the reference creates it, or unity
    everywhere) and only the VU window is asked for runs the many-channel kernels without the gain
    arithmetic; with and without channel maps, two launches per window, ragged lengths."""
    cm = gpu
    rng = np.random.default_rng(700 + C)
    S, T = 5, 9001
    xs = []
    for s in range(S):
        x = _rand_pcm(rng, T * C, "full" if s % 2 == 0 else "edges")
        x[:3] = [-32768, 32767, -32768]
        xs.append(x)
    lens = [T, T - 1, 4097, 17, T]
    for mapped in (False, True):
        cmaps = [[int(v) for v in rng.integers(0, C, C)] if mapped and s != 1 else None for s in range(S)]
        b = cm.Batch(S, C, T, flags=cm.VU)
        for s in range(S):
            if s == 2:
                assert b.set_gain(s, C, 777, [777] * C) == 0      # unity: still no gain
            if cmaps[s] is not None:
                assert b.set_chmap(s, cmaps[s]) == 0
        for lo_f, hi_f in ((0.0, 0.4), (0.4, 1.0)):
            nfr = []
            for s in range(S):
                lo, hi = int(lens[s] * lo_f), int(lens[s] * hi_f)
                b.upload(s, xs[s][lo * C:hi * C])
                nfr.append(hi - lo)
            b.run(max(nfr), frames_per_stream=nfr)
        for s in range(S):
            want = _oracle_block(oracle, xs[s][:lens[s] * C], C, None, cmaps[s])
            rc, r = b.vu_result(s)
            _, ro = _oracle_vu(oracle, [want], C)
            assert rc == 0 and r.as_dict() == of.vu_result_dict(ro), (C, mapped, s)
        # one stream gets a real gain: the whole batch is back on the general kernels
        assert b.set_gain(0, 1, 1000, [1500]) == 0
        for s in range(S):
            b.upload(s, xs[s][:lens[s] * C])
        b.run(T, frames_per_stream=lens)
        for s in range(S):
            want = _oracle_block(oracle, xs[s][:lens[s] * C], C, (1, 1000, [1500]) if s == 0 else None, cmaps[s])
            rc, r = b.vu_result(s)
            _, ro = _oracle_vu(oracle, [want], C)
            assert rc == 0 and r.as_dict() == of.vu_result_dict(ro), (C, mapped, s, "general")
        b.close()


def test_peak_tie_breaks_across_lanes_waves_chunks_and_launches(gpu, oracle):
    """first max-|x| in interleaved order wins, also when the candidates sit in different lanes, rows of a
    step, steps, tiles or launches (ref: src/vumeter.c:163-168); read-only and writing PCM in place (the
    row kernel looks the winning step up again in what it wrote)"""
    cm = gpu
    T = 70000                      # several tiles per stream
    for C, flags in ((1, cm.VU), (2, cm.VU), (3, cm.VU), (3, cm.OUT_PCM | cm.VU | cm.INPLACE), (6, cm.VU),
                     (6, cm.OUT_PCM | cm.VU), (12, cm.VU), (12, cm.OUT_PCM | cm.VU | cm.INPLACE)):
        spots = [(5, 3000), (9, -3000), (1023, 3000), (1024, -3000), (8 * 64 * 4 + 1, 3000),
                 (40000, -3000), (69999, 3000)]
        cases = []
        for first in range(len(spots)):
            x = np.zeros(T * C, dtype=np.int16)
            for (f, v) in spots[first:]:
                x[f * C + (C - 1)] = v
            cases.append(x)
        # equal magnitudes on different channels: global peak is the earliest sample
        x = np.zeros(T * C, dtype=np.int16)
        x[100 * C + 0] = -777
        if C > 1:
            x[50 * C + 1] = 777
        cases.append(x)
        # the same magnitude in every sample from some frame on, signs alternating: every lane, row and tile ties
        x = np.zeros(T * C, dtype=np.int16)
        x[(31 * C + 1):] = 4321
        x[(31 * C + 1)::2] = -4321
        cases.append(x)
        S = len(cases)
        b = cm.Batch(S, C, T, flags=flags)
        for s in range(S):
            b.upload(s, cases[s])
        b.run(T)
        for s in range(S):
            rc, r = b.vu_result(s)
            _, ro = _oracle_vu(oracle, [cases[s]], C)
            assert rc == 0 and r.as_dict() == of.vu_result_dict(ro), (C, flags, s)
        # same value in two launches of one window: the first launch keeps the peak
        blk1 = np.zeros(T * C, dtype=np.int16)
        blk2 = np.zeros(T * C, dtype=np.int16)
        blk1[(T - 1) * C] = -1234
        blk2[0] = 1234
        for s in range(S):
            b.upload(s, blk1)
        b.run(T)
        for s in range(S):
            b.upload(s, blk2)
        b.run(T)
        rc, r = b.vu_result(0)
        _, ro = _oracle_vu(oracle, [blk1, blk2], C)
        assert rc == 0 and r.as_dict() == of.vu_result_dict(ro)
        assert r.global_peak == -1234 and r.frames == 2 * T
        b.close()


def test_window_is_chunk_size_invariant(gpu, oracle):
    """a VU window does not depend on how the stream is cut into launches"""
    cm = gpu
    C, T = 2, 8192
    x = oracle.lcg(4242, T * C)
    one = cm.Batch(1, C, T, flags=cm.VU)
    one.upload(0, x)
    one.run(T)
    _, r1 = one.vu_result(0)
    many = cm.Batch(1, C, 1000, flags=cm.VU)
    pos = 0
    for n in (1, 7, 1000, 999, 512, 3, 1000, 1000, 1000, 1000, 1000, 670):
        many.upload(0, x[pos * C:(pos + n) * C])
        many.run(n)
        pos += n
    assert pos == T
    _, r2 = many.vu_result(0)
    assert r1.as_dict() == r2.as_dict()
    _, ro = _oracle_vu(oracle, [x], C)
    assert r1.as_dict() == of.vu_result_dict(ro)
    one.close()
    many.close()


def test_every_scale_divides_exactly_on_device(gpu, oracle):
    """the mul-hi division against C's truncating division for awkward scales"""
    cm = gpu
    xs = np.array([-32768, -32767, -12345, -3, -2, -1, 0, 1, 2, 3, 12345, 32766, 32767], np.int16)
    scales = [1, 2, 3, 5, 7, 10, 255, 256, 257, 641, 1000, 4095, 4096, 4097, 21845, 32767, 32768,
              32769, 43691, 65521, 65534, 65535]
    gains = [0, 1, 2, 3, 999, 1000, 1001, 32767, 32768, 65534, 65535]
    S = len(scales) * len(gains)
    b = cm.Batch(S, 1, 16, flags=cm.OUT_PCM)
    combos = [(sc, g) for sc in scales for g in gains]
    for s, (sc, g) in enumerate(combos):
        assert b.set_gain(s, 1, sc, [g]) == 0
        b.upload(s, xs)
    b.run(xs.size)
    for s, (sc, g) in enumerate(combos):
        want = _oracle_block(oracle, xs, 1, (1, sc, [g]), None)
        assert np.array_equal(b.download(s, xs.size), want), (sc, g)
    b.close()


def test_results_for_all_streams_and_snapshot_overlap(gpu, oracle):
    cm = gpu
    S, C, T = 37, 2, 2048
    b = cm.Batch(S, C, T, flags=cm.VU | cm.OUT_PCM)
    assert b.set_gain(-1, 2, 1000, [750, 1250]) == 0
    assert b.set_chmap(-1, [1, 0]) == 0
    b.generate(cm.GEN_NOISE, 999, T)
    b.run(T)
    b.vu_snapshot()                       # window 1 travels to the host ...
    b.generate(cm.GEN_NOISE, 555, T)
    b.run(T)                              # ... while window 2 accumulates
    res1, rc1 = b.vu_collect()
    res1 = [r.as_dict() for r in res1]
    res2, rc2 = b.vu_results()
    for s in range(S):
        for seed, res, rcs in ((999, res1, rc1), (555, res2, rc2)):
            want = _oracle_block(oracle, oracle.lcg(seed + s, T * C), C, (2, 1000, [750, 1250]),
                                 [1, 0])
            _, ro = _oracle_vu(oracle, [want], C)
            got = res[s] if isinstance(res[s], dict) else res[s].as_dict()
            assert rcs[s] == 0 and got == of.vu_result_dict(ro), (s, seed)
    b.close()


@pytest.mark.parametrize("S,C", [(700, 2), (1500, 1), (600, 5)])
def test_window_per_block_with_the_collect_in_two_halves(gpu, oracle, S, C):
    """The small-block loop of bench.py: a VU window per block, the snapshot packed by one kernel into host
    memory (1 + 2C words per window), the dB finish begun on the helper threads and ended a launch later
    (cmhip_batch_vu_collect_begin / _end; S >= 512: the pool works).  Every block's windows of sampled
    streams against the oracle; the ring of two pending snapshots refuses a third; begin twice is refused."""
    import ctypes as Ct
    cm = gpu
    T, blocks = 480, 9
    b = cm.Batch(S, C, T, flags=cm.VU | cm.OUT_PCM)
    gains = [750, 1250, 900, 1000, 300][:C]
    assert b.set_gain(-1, C, 1000, gains) == 0
    outs = [((cm.VuResult * S)(), (Ct.c_int * S)()) for _ in range(blocks)]
    collecting = None
    for k in range(blocks):
        b.generate(cm.GEN_NOISE, 4000 + k, T)
        b.run(T)
        b.vu_snapshot()                      # pending now: k-1 being finished, (k) new -- or k-2, k-1, k
        if k >= 2:
            assert cm.lib.cmhip_batch_vu_snapshot(b.h) == cm.ERROR_BUSY          # three are pending
        if collecting is not None:
            b.vu_collect_end()
            collecting = None
        if k >= 1:
            b.vu_collect_begin(*outs[k - 1])
            assert cm.lib.cmhip_batch_vu_collect_begin(b.h, outs[k][0], outs[k][1]) == cm.ERROR_BUSY
            collecting = k - 1
    b.vu_collect_end()
    b.vu_collect(*outs[blocks - 1])
    assert cm.lib.cmhip_batch_vu_collect_end(b.h) == cm.ERROR_INVAL
    for k in range(blocks):
        res, rcs = outs[k]
        for s in (0, 1, 63, 64, S // 2, S - 1):
            want = _oracle_block(oracle, oracle.lcg(4000 + k + s, T * C), C, (C, 1000, gains), None)
            _, ro = _oracle_vu(oracle, [want], C)
            assert rcs[s] == 0 and res[s].as_dict() == of.vu_result_dict(ro), (k, s)
    # a window nobody filled: INVAL for every stream, as coolmic_vumeter_result (ref: src/vumeter.c:198-199)
    res, rcs = b.vu_results()
    assert all(r == cm.ERROR_INVAL for r in rcs)
    b.close()


def test_node_partial_matches_host_merge(gpu, oracle):
    """config 5's per-GPU record; two 'ranks' emulated as two batches on this GPU and
    combined on the host the way the all-reduce combines them (SUM / MAX)"""
    cm = gpu
    C, T, S = 2, 4096, 16
    N = 2
    words = []
    allblocks = []
    for rank in range(N):
        b = cm.Batch(S // N, C, T, flags=cm.VU)
        assert b.set_gain(-1, 2, 1000, [750, 1250]) == 0
        b.generate(cm.GEN_NOISE, 31337, T, first_global=rank, global_step=N)
        b.run(T)
        dst = cm.DeviceWords(cm.NODE_WORDS)
        b.node_partial(dst.dev, first_global=rank, global_step=N)
        b.sync()
        words.append(dst.read())
        dst.free()
        # the host form of the same record (for a host that brings its own collective)
        assert np.array_equal(b.node_record(first_global=rank, global_step=N), words[-1])
        for s in range(S // N):
            gs = rank + N * s
            allblocks.append(_oracle_block(oracle, oracle.lcg(31337 + gs, T * C), C,
                                           (2, 1000, [750, 1250]), None))
        b.close()
    merged = np.concatenate([words[0][:17] + words[1][:17],
                             np.maximum(words[0][17:], words[1][17:])])
    rc, r = cm.node_finish(merged, C)
    assert rc == 0 and r.frames == S * T
    # expected: sums over all streams; peak = largest magnitude over all streams
    pw = np.zeros(C, dtype=np.int64)
    peaks = []
    for blk in allblocks:
        v = oracle.vu_new(C)
        oracle.vu_accumulate(v, blk)
        pw += np.array([v.power[c] for c in range(C)], dtype=np.int64)
        peaks.append([v.result.channel_peak[c] for c in range(C)])
    for c in range(C):
        assert r.channel_power[c] == oracle.lib.oracle_power_db(int(pw[c]), S * T)
        assert abs(int(r.channel_peak[c])) == max(abs(int(p[c])) for p in peaks)
    assert r.global_power == oracle.lib.oracle_power_db(int(pw.sum()), S * T * C)
    assert abs(int(r.global_peak)) == max(abs(int(v)) for p in peaks for v in p)
    # the host-side merge of the C ABI ("replicas only" form) is the same combine
    assert np.array_equal(cm.node_merge_host(np.stack(words)), merged)


def test_node_exchange_through_rccl_one_rank(gpu, oracle):
    """cmhip_node_*: the RCCL path of config 5 with a one-rank communicator (all a 1-GPU box can
    hold): records of three blocks in slots of both sets, all-reduced by ncclAllReduce(int64, sum) +
    ncclAllReduce(uint64, max), fetched, and equal to the records cmhip_batch_vu_node_partial writes
    in one piece -- which test_node_partial_matches_host_merge pins against the oracle"""
    cm = gpu
    C, T, S = 2, 4096, 24
    node = cm.Node(0, 1, 0, cm.node_unique_id(), max_records=3)
    assert cm.lib.cmhip_node_ranks(node.h) == 1
    # librccl comes from next to the HIP runtime the engine is bound to -- the system's (conftest loads the
    # engine before anything brings torch's copies): the runtime pair bench.py runs on
    hip_path, rccl_path = (kv.split("=", 1)[1] for kv in cm.node_runtime().split(" "))
    assert os.path.dirname(os.path.realpath(hip_path)) == os.path.dirname(os.path.realpath(rccl_path)), cm.node_runtime()
    assert "/torch/" not in hip_path, cm.node_runtime()
    b = cm.Batch(S, C, T, flags=cm.VU)
    assert b.set_gain(-1, 2, 1000, [750, 1250]) == 0
    direct = []
    for k in range(4):
        b.generate(cm.GEN_NOISE, 99, T, first_global=3, global_step=5, frame_offset=k * T)
        b.run(T)
        dst = cm.DeviceWords(cm.NODE_WORDS)
        b.node_partial(dst.dev, first_global=3, global_step=5)
        set_, slot = (0, k) if k < 3 else (1, 0)
        node.partial(b, set_, slot, first_global=3, global_step=5)
        b.sync()
        direct.append(dst.read())
        dst.free()
        b.vu_reset(-1)
        if k == 2:
            node.allreduce(0, 3, after=b)
    node.allreduce(1, 1, after=b)
    got0 = node.fetch(0, 3)
    got1 = node.fetch(1, 1)
    for k in range(3):
        assert np.array_equal(got0[k], direct[k]), k
    assert np.array_equal(got1[0], direct[3])
    rc, r = cm.node_finish(got0[1], C)
    assert rc == 0 and r.frames == S * T
    # a set can be refilled after its exchange (the batch's stream waits for it on the device)
    b.generate(cm.GEN_NOISE, 99, T, first_global=3, global_step=5)
    b.run(T)
    node.partial(b, 0, 0, first_global=3, global_step=5)
    node.allreduce(0, 1, after=b)
    assert np.array_equal(node.fetch(0, 1)[0], direct[0])
    # argument checks
    assert cm.lib.cmhip_node_partial(node.h, b.h, 2, 0, 0, 1) == cm.ERROR_INVAL
    assert cm.lib.cmhip_node_partial(node.h, b.h, 0, 3, 0, 1) == cm.ERROR_INVAL
    assert cm.lib.cmhip_node_allreduce(node.h, 0, 4, None) == cm.ERROR_INVAL
    assert cm.lib.cmhip_node_new(0, 2, 2, (cm.C.c_ubyte * 128)(), 4) is None
    b.close()
    node.close()


def test_config5_per_gpu_shape_through_the_rccl_path(gpu, oracle):
    """BASELINE config 5 at the shape one GPU of eight carries -- 8192 mono streams x 65536 frames, gain 900/1000,
    PCM materialised -- through the node-global VU exactly as bench.py's c5 leg drives it: per block
    cmhip_node_partial, per NB = 4 blocks ONE cmhip_node_allreduce (ncclAllReduce(int64, sum) +
    ncclAllReduce(uint64, max)) on a one-rank communicator (all a one-GPU box can hold; the N > 1 exchange is
    unmeasured on hardware), then cmhip_node_fetch.  Size-independent properties, per block:
      * the fetched record == cmhip_node_merge_host of the un-reduced record == the host form of it;
      * its 16 + 1 sums == the sums of cmhip_batch_vu_raw over ALL 8192 streams (a checksum of checksums), its
        frames == 8192 x 65536;
      * its global peak key names the right sample: largest magnitude over all streams, among equals the
        earliest frame, among those the lowest global stream id -- recomputed on the host from the generator for
        every stream of block 0 (LCG jump-ahead in numpy), and checked against the oracle's window of that stream;
      * sampled streams: PCM and the raw window against the oracle."""
    cm = gpu
    S, C, T, NB = 8192, 1, 65536, 4
    first_global, step = 3, 8                                       # rank 3 of 8: global stream 3 + 8 s
    node = cm.Node(0, 1, 0, cm.node_unique_id(), max_records=NB)
    b = cm.Batch(S, C, T, flags=cm.OUT_PCM | cm.VU)
    assert b.set_gain(-1, 1, 1000, [900]) == 0
    _, g = oracle.gain(1, 1, 1000, [900])
    own, sums, mags = [], [], []
    for k in range(NB):
        b.generate(cm.GEN_NOISE, 12345, T, first_global=first_global, global_step=step, frame_offset=k * T)
        b.run(T)
        own.append(b.node_record(first_global=first_global, global_step=step))
        node.partial(b, 0, k, first_global=first_global, global_step=step)
        tot, top = 0, 0
        for s_ in range(S):
            power, peak, frames = b.vu_raw(s_)
            assert frames == T
            tot += int(power[0])
            top = max(top, abs(int(peak[0])))
        sums.append(tot)
        mags.append(top)
        if k == 0:
            for s_ in (0, 1, 4095, 8191):
                want = oracle.gain_apply(g, oracle.lcg(12345 + first_global + s_ * step, T), 1)
                assert np.array_equal(b.download(s_, T), want), s_
                v = oracle.vu_new(1)
                oracle.vu_accumulate(v, want)
                power, peak, _ = b.vu_raw(s_)
                assert int(power[0]) == int(v.power[0]) and int(peak[0]) == int(v.result.channel_peak[0]), s_
        b.vu_reset(-1)
    node.allreduce(0, NB, after=b)
    got = node.fetch(0, NB)
    for k in range(NB):
        assert np.array_equal(got[k], own[k]), k
        assert np.array_equal(cm.node_merge_host(own[k][None, :]), own[k]), k
        w = got[k].astype(np.uint64)
        assert int(w[0]) == sums[k] and int(w[16]) == S * T, k
        assert not w[1:16].any() and not w[18:33].any()              # mono: the other channels' slots stay empty
        assert int(w[17]) == int(w[33])                               # one channel: its key is the global one
        assert int(w[33]) >> 46 == mags[k], k
        rc, r = cm.node_finish(got[k], C)
        assert rc == 0 and r.frames == S * T and abs(int(r.global_peak)) == mags[k]
        assert r.global_power == oracle.lib.oracle_power_db(sums[k], S * T)
    # block 0: which sample the global key names.  |x| * 900 // 1000 is largest for x = -32768 (29491; 32767 gives
    # 29490), so the winner is the first -32768 over all streams: earliest frame, then lowest global stream.
    a_t = np.empty(T, dtype=np.uint32)                                # x_t = a_t * seed + c_t  (t = 1 .. T draws)
    c_t = np.empty(T, dtype=np.uint32)
    a, c = np.uint32(1), np.uint32(0)
    with np.errstate(over="ignore"):
        for t in range(T):
            a = np.uint32(a * np.uint32(1664525))
            c = np.uint32(c * np.uint32(1664525) + np.uint32(1013904223))
            a_t[t], c_t[t] = a, c
        best = None
        for s_ in range(S):
            gs = first_global + s_ * step
            hit = np.flatnonzero(((a_t * np.uint32(12345 + gs) + c_t) >> np.uint32(16)) == 0x8000)
            if hit.size and (best is None or (int(hit[0]), gs) < best):
                best = (int(hit[0]), gs)
    assert best is not None and mags[0] == 29491
    key = int(got[0].astype(np.uint64)[33])
    frame = (1 << 29) - 1 - ((key >> 17) & ((1 << 29) - 1))
    stream = 65535 - ((key >> 1) & 0xffff)
    assert (frame, stream, key & 1) == (best[0], best[1], 1), (frame, stream, best)
    b.close()
    node.close()


def test_full_size_config2_properties(gpu, oracle):
    """BASELINE config 2 at full size (4096 x 2ch x 65536 frames): sampled streams against
    the oracle, VU-only run equal to the PCM run, two half-blocks equal to one block."""
    cm = gpu
    S, C, T = 4096, 2, 65536
    gains, cmap = [750, 1250], [1, 0]
    b = cm.Batch(S, C, T, flags=cm.OUT_PCM | cm.VU)
    assert b.set_gain(-1, 2, 1000, gains) == 0 and b.set_chmap(-1, cmap) == 0
    b.generate(cm.GEN_NOISE, 12345, T)
    b.run(T)
    res, rcs = b.vu_results()
    full = [r.as_dict() for r in res]
    assert all(rc == 0 for rc in rcs)
    for s in (0, 1, 7, 2047, 4095):
        want = _oracle_block(oracle, oracle.lcg(12345 + s, T * C), C, (2, 1000, gains), cmap)
        assert np.array_equal(b.download(s, T), want), s
        _, ro = _oracle_vu(oracle, [want], C)
        assert full[s] == of.vu_result_dict(ro), s
    # checksum of checksums: total power over all streams equals the sum of per-stream raws
    b.close()
    v = cm.Batch(S, C, T // 2, flags=cm.VU)
    assert v.set_gain(-1, 2, 1000, gains) == 0 and v.set_chmap(-1, cmap) == 0
    for half in range(2):
        v.generate(cm.GEN_NOISE, 12345, T // 2, frame_offset=half * (T // 2))
        v.run(T // 2)
    res2, rcs2 = v.vu_results()
    assert all(rc == 0 for rc in rcs2)
    bad = [s for s in range(S) if res2[s].as_dict() != full[s]]
    assert not bad, bad[:8]
    v.close()


@pytest.mark.parametrize("place", ["off", "flag", "env2"])
def test_arrays_placed_apart_hold_the_same_results(gpu, oracle, place, monkeypatch):
    """A batch created with CMHIP_PLACE_SEARCH (PCM arrays of 256 MiB and more) may move both of them at
    the end of its creation, to where its own run is fastest (`place_arrays_apart`, DESIGN 4.1;
    $CMHIP_PLACE=2 makes every batch search, 0 none).  Whatever it chose, the probes leave nothing behind:
    uploads, the input read back, PCM and the windows of the first launch against the oracle; and the
    search keeps inside its stated budget, half of the memory the card reported free."""
    cm = gpu
    monkeypatch.delenv("CMHIP_PLACE", raising=False)
    if place == "env2":
        monkeypatch.setenv("CMHIP_PLACE", "2")
    S, C, T = 1024, 2, 65536                      # 256 MiB per array
    b = cm.Batch(S, C, T, flags=cm.OUT_PCM | cm.VU | (cm.PLACE_SEARCH if place == "flag" else 0))
    rec = b.placement()
    assert rec["searched"] == (place != "off"), rec
    if rec["searched"]:
        assert 2 <= rec["candidates"] <= 7 and rec["probe_launches"] > 0 and rec["first_pair_ms"] > 0
        assert rec["GiB_requested"] <= 0.5 * rec["GiB_free_before"] + 0.01, rec
        assert rec["chosen"][0] != rec["chosen"][1] and max(rec["chosen"]) < rec["candidates"]
    else:
        assert rec["probe_launches"] == 0 and rec["GiB_requested"] == 0 and rec["chosen"] == [0, 1]
    for s in (0, 511, 1023):                      # nothing in the windows, zeros in the arrays
        assert b.vu_raw(s)[2] == 0
        assert not b.download(s, 64).any()
    gains, cmap = [750, 1250], [1, 0]
    assert b.set_gain(-1, 2, 1000, gains) == 0 and b.set_chmap(-1, cmap) == 0
    rng = np.random.default_rng(31)
    xs = {s: _rand_pcm(rng, T * C, "full") for s in (0, 3, 1023)}
    b.generate(cm.GEN_NOISE, 12345, T)
    for s, x in xs.items():
        b.upload(s, x)
        assert np.array_equal(b.download_input(s, T), x), s
    b.run(T)
    res, rcs = b.vu_results()
    for s in (0, 3, 7, 512, 1023):
        src = xs[s] if s in xs else oracle.lcg(12345 + s, T * C)
        want = _oracle_block(oracle, src, C, (2, 1000, gains), cmap)
        assert np.array_equal(b.download(s, T), want), (place, s)
        _, ro = _oracle_vu(oracle, [want], C)
        assert rcs[s] == 0 and res[s].as_dict() == of.vu_result_dict(ro), (place, s)
    b.close()


def test_no_placement_search_unless_asked_for(gpu, monkeypatch):
    """The library's default: creating a large batch is two allocations and no probe launch -- fast, and
    the card's free memory afterwards is down by the batch's own arrays and tables, nothing else."""
    import time
    cm = gpu
    monkeypatch.delenv("CMHIP_PLACE", raising=False)
    warm = cm.Batch(1, 2, 64, flags=cm.OUT_PCM | cm.VU)       # the device's first batch pays the runtime's start
    warm.close()
    cm.device_synchronize(0)
    free0, _ = cm.device_mem_info(0)
    S, C, T = 1024, 2, 65536                      # 256 MiB per array: large enough for a search
    t0 = time.perf_counter()
    b = cm.Batch(S, C, T, flags=cm.OUT_PCM | cm.VU)
    dt = time.perf_counter() - t0
    free1, _ = cm.device_mem_info(0)
    rec = b.placement()
    own = 2 * S * b.stride * 2
    assert not rec["searched"] and rec["probe_launches"] == 0 and rec["GiB_requested"] == 0, rec
    assert dt < 0.5, dt
    assert own <= free0 - free1 <= own + (64 << 20), (free0 - free1, own)
    b.close()


def test_full_size_config4_total_on_one_gpu(gpu, oracle):
    """BASELINE configs 4/5 hold 65 536 mono streams; all of them on ONE GPU at 65 536 frames are
    8 GiB of PCM per array, so slot offsets pass 2^32 bytes (stream 32 768 starts at exactly 4 GiB).
    Sampled streams on both sides of that line against the oracle; the node-global record of the
    batch against the host merge of all 65 536 per-stream windows (a checksum of checksums)."""
    cm = gpu
    S, C, T = 65536, 1, 65536
    b = cm.Batch(S, C, T, flags=cm.OUT_PCM | cm.VU)
    assert b.set_gain(-1, 1, 1000, [900]) == 0
    b.generate(cm.GEN_NOISE, 12345, T)
    b.run(T)
    rc, node = cm.node_finish(b.node_record(), C)
    assert rc == 0 and node.frames == S * T
    total, top = 0, 0
    for s in range(S):
        power, peak, frames = b.vu_raw(s)
        assert frames == T, s
        total += int(power[0])
        top = max(top, abs(int(peak[0])))
    assert node.channel_power[0] == oracle.lib.oracle_power_db(total, S * T)
    assert abs(int(node.global_peak)) == top
    res, rcs = b.vu_results()
    assert all(r == 0 for r in rcs)
    for s in (0, 1, 32767, 32768, 32769, 65535):
        want = _oracle_block(oracle, oracle.lcg(12345 + s, T), C, (1, 1000, [900]), None)
        assert np.array_equal(b.download(s, T), want), s
        _, ro = _oracle_vu(oracle, [want], C)
        assert res[s].as_dict() == of.vu_result_dict(ro), s
    b.close()


@pytest.mark.parametrize("C", [3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16])
def test_wide_channel_kernels(gpu, oracle, C):
    """more than two channels with identity maps take the vector kernels (k_run_wide for 4/8/16,
    k_run_rows for every other count): ragged lengths, per-stream gains, in place and not,
    VU only, float planes, windows over two launches"""
    cm = gpu
    rng = np.random.default_rng(500 + C)
    lens = [0, 1, 2, 3, 63, 64, 65, 127, 128, 129, 511, 1000, 1023, 1024, 1025, 3000, 9973]
    S, T = len(lens), max(lens)
    xs = [_rand_pcm(rng, lens[s] * C, ["full", "edges", "small"][s % 3]) for s in range(S)]
    gas = []
    for s in range(S):
        if s % 4 == 0:
            gas.append(None)
        elif s % 4 == 1:
            gas.append((C, int(rng.integers(1, 65536)), [int(v) for v in rng.integers(0, 65536, C)]))
        elif s % 4 == 2:
            gas.append((1, 1000, [int(rng.integers(0, 5000))]))
        else:
            gas.append((C, 1, [65535] * C))
    wants = [_oracle_block(oracle, xs[s], C, gas[s], None) for s in range(S)]
    for flags in (cm.OUT_PCM | cm.VU, cm.OUT_PCM | cm.VU | cm.INPLACE, cm.VU, cm.OUT_PCM,
                  cm.OUT_F32 | cm.OUT_PCM | cm.VU, cm.OUT_F32):
        b = cm.Batch(S, C, T, flags=flags)
        for s in range(S):
            if gas[s] is not None:
                assert b.set_gain(s, *gas[s]) == 0
            if lens[s]:
                b.upload(s, xs[s])
        b.run(T, frames_per_stream=lens)
        for s in range(S):
            if flags & cm.OUT_PCM:
                got = b.download(s, lens[s]) if lens[s] else np.zeros(0, np.int16)
                assert np.array_equal(got, wants[s]), (C, s, flags)
            if flags & cm.OUT_F32 and lens[s]:
                planar = oracle.to_f32_planar(wants[s], C)
                for c in range(C):
                    gotf = b.download_f32(s, c, lens[s])
                    assert np.array_equal(gotf.view(np.uint32), planar[c].view(np.uint32)), (C, s, c)
        if flags & cm.VU:
            # second launch into the same window: the reversed block
            for s in range(S):
                if lens[s]:
                    b.upload(s, xs[s][::-1].copy())
            b.run(T, frames_per_stream=lens)
            for s in range(S):
                w2 = _oracle_block(oracle, xs[s][::-1].copy(), C, gas[s], None)
                rc_o, r_o = _oracle_vu(oracle, [wants[s], w2], C)
                rc_g, r_g = b.vu_result(s)
                assert rc_g == rc_o, (C, s, flags)
                if rc_o == 0:
                    assert r_g.as_dict() == of.vu_result_dict(r_o), (C, s, flags)
        b.close()


@pytest.mark.parametrize("C", [3, 4, 6, 8, 13, 16])
def test_channel_maps_on_many_channels(gpu, oracle, C):
    """channel maps (permutations and maps that repeat a channel) on more than two channels:
    k_run_rows gathers through LDS; every output set, in place, ragged lengths"""
    cm = gpu
    rng = np.random.default_rng(900 + C)
    lens = [0, 1, 2, 5, 63, 64, 65, 500, 1021, 1024, 2049, 7777]
    S, T = len(lens), max(lens)
    xs = [_rand_pcm(rng, lens[s] * C, ["full", "edges", "small"][s % 3]) for s in range(S)]
    gas, maps = [], []
    for s in range(S):
        gas.append(None if s % 3 == 0 else
                   (C, int(rng.integers(1, 65536)), [int(v) for v in rng.integers(0, 65536, C)]))
        if s % 4 == 3:
            maps.append(None)                                   # a stream without a map among mapped ones
        elif s % 2 == 0:
            maps.append([int(v) for v in rng.permutation(C)])
        else:
            maps.append([int(v) for v in rng.integers(0, C, C)])
    wants = [_oracle_block(oracle, xs[s], C, gas[s], maps[s]) for s in range(S)]
    for flags in (cm.OUT_PCM | cm.VU, cm.OUT_PCM | cm.VU | cm.INPLACE, cm.VU,
                  cm.OUT_F32 | cm.OUT_PCM | cm.VU, cm.OUT_F32):
        b = cm.Batch(S, C, T, flags=flags)
        for s in range(S):
            if gas[s] is not None:
                assert b.set_gain(s, *gas[s]) == 0
            if maps[s] is not None:
                assert b.set_chmap(s, maps[s]) == 0
            if lens[s]:
                b.upload(s, xs[s])
        b.run(T, frames_per_stream=lens)
        for s in range(S):
            if flags & cm.OUT_PCM:
                got = b.download(s, lens[s]) if lens[s] else np.zeros(0, np.int16)
                assert np.array_equal(got, wants[s]), (C, s, flags)
            if flags & cm.OUT_F32 and lens[s]:
                planar = oracle.to_f32_planar(wants[s], C)
                for c in range(C):
                    gotf = b.download_f32(s, c, lens[s])
                    assert np.array_equal(gotf.view(np.uint32), planar[c].view(np.uint32)), (C, s, c)
            if flags & cm.VU:
                rc_o, r_o = _oracle_vu(oracle, [wants[s]], C)
                rc_g, r_g = b.vu_result(s)
                assert rc_g == rc_o, (C, s, flags)
                if rc_o == 0:
                    assert r_g.as_dict() == of.vu_result_dict(r_o), (C, s, flags)
        b.close()


@pytest.mark.parametrize("C", [1, 2, 6])
def test_host_resident_slots(gpu, oracle, C):
    """CMHIP_HOSTPCM: the PCM slots live in pinned host memory that the kernels read and write
    directly (what the per-stream stages use for their 1 KiB blocks): same results, block after
    block into one window, also in place and with the equaliser"""
    cm = gpu
    rng = np.random.default_rng(77 + C)
    S, T = 5, 700
    for extra in (0, cm.INPLACE, cm.EQ):
        b = cm.Batch(S, C, T, flags=cm.OUT_PCM | cm.VU | cm.HOSTPCM | extra)
        gas = []
        for s in range(S):
            ga = None if s == 0 else (C, 1000, [int(v) for v in rng.integers(100, 3000, C)])
            if ga:
                assert b.set_gain(s, *ga) == 0
            gas.append(ga)
        wants = [[] for _ in range(S)]
        for k in range(4):
            lens = [int(v) for v in rng.integers(0, T + 1, S)]
            xs = [_rand_pcm(rng, lens[s] * C, "full") for s in range(S)]
            for s in range(S):
                if lens[s]:
                    b.upload(s, xs[s])
            b.run(T, frames_per_stream=lens)
            for s in range(S):
                want = _oracle_block(oracle, xs[s], C, gas[s], None)
                got = b.download(s, lens[s]) if lens[s] else np.zeros(0, np.int16)
                assert np.array_equal(got, want), (C, extra, k, s)
                wants[s].append(want)
        for s in range(S):
            rc_o, r_o = _oracle_vu(oracle, wants[s], C)
            rc_g, r_g = b.vu_result(s)
            assert rc_g == rc_o, (C, s)
            if rc_o == 0:
                assert r_g.as_dict() == of.vu_result_dict(r_o), (C, extra, s)
        b.close()


def test_batch_api_error_paths(gpu):
    """argument checking of the C ABI (include/coolmic_hip.h): errors are numbers, never faults"""
    import ctypes as C
    cm = gpu
    with pytest.raises(cm.CoolmicError):
        cm.Batch(0, 2, 64)
    with pytest.raises(cm.CoolmicError):
        cm.Batch(1, 17, 64)
    with pytest.raises(cm.CoolmicError):
        cm.Batch(1, 2, 64, flags=0)
    with pytest.raises(cm.CoolmicError):
        cm.Batch(1, 1, 64, device=99)
    b = cm.Batch(2, 2, 64, flags=cm.OUT_PCM | cm.VU)
    assert b.set_gain(2, 2, 1000, [1, 1]) == cm.ERROR_INVAL       # stream out of range
    assert b.set_gain(0, 3, 1000, [1, 1, 1]) == cm.ERROR_INVAL    # shape rule of the reference
    assert b.set_gain(0, 2, 0, [1, 1]) == 0                       # disables
    assert b.set_chmap(0, [0, 2]) == cm.ERROR_INVAL
    assert b.set_eq(0, cm.eq3()) == cm.ERROR_INVAL                # batch without EQ
    assert cm.lib.cmhip_batch_run(b.h, 65, None) == cm.ERROR_INVAL
    assert cm.lib.cmhip_batch_run(b.h, 0, None) == 0
    bad = (C.c_uint32 * 2)(10, 100)
    assert cm.lib.cmhip_batch_run(b.h, 64, bad) == cm.ERROR_INVAL
    x = np.zeros(65 * 2, dtype=np.int16)
    assert cm.lib.cmhip_batch_upload(b.h, 0, x.ctypes.data, 65) == cm.ERROR_INVAL
    assert cm.lib.cmhip_batch_upload(b.h, 2, x.ctypes.data, 1) == cm.ERROR_INVAL
    assert cm.lib.cmhip_batch_upload(b.h, 0, None, 1) == cm.ERROR_FAULT
    assert cm.lib.cmhip_batch_download_f32(b.h, 0, 0, x.ctypes.data, 1) == cm.ERROR_INVAL
    assert b.vu_result(0)[0] == cm.ERROR_INVAL                   # nothing accounted yet
    assert cm.lib.cmhip_batch_vu_collect(b.h, None, None) == cm.ERROR_FAULT
    r = (cm.VuResult * 2)()
    assert cm.lib.cmhip_batch_vu_collect(b.h, r, None) == cm.ERROR_INVAL    # no snapshot pending
    b.vu_snapshot()
    b.vu_snapshot()
    b.vu_snapshot()                                               # three may be pending
    assert cm.lib.cmhip_batch_vu_snapshot(b.h) == cm.ERROR_BUSY
    b.vu_collect()
    b.vu_collect()
    b.vu_collect()
    assert cm.lib.cmhip_batch_run(None, 1, None) == cm.ERROR_FAULT
    assert b"" != cm.lib.cmhip_last_error()
    vonly = cm.Batch(1, 1, 64, flags=cm.VU)
    assert cm.lib.cmhip_batch_download(vonly.h, 0, x.ctypes.data, 1) == cm.ERROR_INVAL
    assert vonly.ceiling(1, 1) < 0                                # copy needs a PCM output
    vonly.close()
    b.close()


@pytest.mark.parametrize("C", [1, 2, 4, 3])
def test_random_sessions_match_per_stream_oracle(gpu, oracle, C):
    """long random sessions: ragged blocks, gain / map changes between launches, results of
    random streams at random times, whole-batch snapshots in between -- every result and
    every PCM block must equal what one oracle transform+vumeter per stream produces"""
    cm = gpu
    rng = np.random.default_rng(9000 + C)
    S, T = 12, 2500
    b = cm.Batch(S, C, T, flags=cm.OUT_PCM | cm.VU)
    gains = [None] * S                                 # (channels, scale, gains) or None
    maps = [None] * S
    windows = [oracle.vu_new(C) for _ in range(S)]

    def check_result(s, rc, r):
        rc_o, r_o = oracle.vu_result(windows[s])        # resets the oracle window on success
        assert rc == rc_o, (C, s)
        if rc_o == 0:
            assert r.as_dict() == of.vu_result_dict(r_o), (C, s)

    for launch in range(40):
        for s in range(S):
            roll = rng.integers(0, 10)
            if roll == 0:
                gains[s] = (C, int(rng.integers(1, 4000)), [int(v) for v in rng.integers(0, 5000, C)])
                assert b.set_gain(s, *gains[s]) == 0
            elif roll == 1:
                gains[s] = None
                assert b.set_gain(s, 0, 0, None) == 0
            elif roll == 2 and C <= 2:
                maps[s] = [int(v) for v in rng.integers(0, C, C)]
                assert b.set_chmap(s, maps[s]) == 0
            elif roll == 3:
                maps[s] = None
                assert b.set_chmap(s, None) == 0
        lens = [int(v) for v in rng.integers(0, T + 1, S)]
        if launch % 7 == 0:
            lens = [T] * S
        xs = [_rand_pcm(rng, lens[s] * C, ["full", "edges", "small"][int(rng.integers(0, 3))])
              for s in range(S)]
        for s in range(S):
            if lens[s]:
                b.upload(s, xs[s])
        b.run(T, frames_per_stream=lens)
        for s in range(S):
            want = _oracle_block(oracle, xs[s], C, gains[s], maps[s])
            if lens[s]:
                assert np.array_equal(b.download(s, lens[s]), want), (C, launch, s)
            oracle.vu_accumulate(windows[s], want)
        what = rng.integers(0, 4)
        if what == 0:                                   # some single results
            for s in rng.choice(S, 3, replace=False):
                rc, r = b.vu_result(int(s))
                check_result(int(s), rc, r)
        elif what == 1:                                 # all windows at once
            res, rcs = b.vu_results()
            for s in range(S):
                if rcs[s] == 0:
                    check_result(s, 0, res[s])
                else:
                    assert windows[s].result.frames == 0
                    oracle.lib.oracle_vumeter_reset(windows[s])
    b.close()


def test_run_on_slot_arrays_named_per_run(gpu, oracle):
    """cmhip_batch_run_slots: a batch without PCM arrays of its own (CMHIP_EXTSLOTS) runs on pinned,
    device-mapped host arrays named per run -- two input sets and two output sets in rotation, the VU
    window and the parameters staying with the batch -- and the entries that need own slots say so"""
    cm = gpu
    C, T, S = 2, 3000, 19
    b = cm.Batch(S, C, T, flags=cm.OUT_PCM | cm.VU | cm.EXTSLOTS)
    assert b.set_gain(-1, 2, 1000, [750, 1250]) == 0
    assert b.set_chmap(3, [1, 0]) == 0
    ins = [cm.MappedPcm(b), cm.MappedPcm(b)]
    outs = [cm.MappedPcm(b), cm.MappedPcm(b)]
    rng = np.random.default_rng(5)
    wants = [[] for _ in range(S)]
    for k in range(4):
        lens = [int(v) for v in rng.integers(0, T + 1, S)]
        lens[0] = T
        i, o = ins[k & 1], outs[k & 1]
        blocks = []
        for s in range(S):
            x = _rand_pcm(rng, lens[s] * C, "full")
            i.array[s, :x.size] = x
            blocks.append(x)
        b.run_slots(T, i.dev, o.dev, frames_per_stream=lens)
        b.sync()
        for s in range(S):
            want = _oracle_block(oracle, blocks[s], C, (2, 1000, [750, 1250]), [1, 0] if s == 3 else None)
            assert np.array_equal(o.array[s, :want.size], want), (k, s)
            wants[s].append(want)
    res, rcs = b.vu_results()
    for s in range(S):
        rc_o, ro = _oracle_vu(oracle, wants[s], C)
        assert rcs[s] == rc_o and (rc_o != 0 or res[s].as_dict() == of.vu_result_dict(ro)), s
    # no slots of its own
    assert cm.lib.cmhip_batch_run(b.h, 10, None) == cm.ERROR_INVAL
    x = np.zeros(10 * C, np.int16)
    assert cm.lib.cmhip_batch_upload(b.h, 0, x.ctypes.data, 10) == cm.ERROR_INVAL
    assert cm.lib.cmhip_batch_generate(b.h, 0, 0, 10, 0, 1, 0) == cm.ERROR_INVAL
    assert cm.lib.cmhip_batch_run_slots(b.h, 10, None, ins[0].dev, None) == cm.ERROR_INVAL   # writes PCM: needs an output
    for m in ins + outs:
        m.free()
    b.close()


def test_kernel_timing_on_every_launch_and_on_a_sample(gpu):
    """`cmhip_batch_timing(b, 1)` stamps events on every run, `(b, n)` on every n-th (what `bench.py` uses:
    the events cost a run about 5 us of its stream's time); `timing_read` returns their sum and count and
    starts over."""
    cm = gpu
    b = cm.Batch(8, 2, 4096, flags=cm.OUT_PCM | cm.VU)
    b.generate(cm.GEN_NOISE, 1, 4096)
    b.timing(True)
    for _ in range(6):
        b.run(4096)
    ms, n = b.timing_read()
    assert n == 6 and 0 < ms < 100
    b.timing(4)
    for _ in range(10):                          # runs 0, 4, 8 of these carry the events
        b.run(4096)
    ms, n = b.timing_read()
    assert n == 3 and 0 < ms < 100
    ms, n = b.timing_read()
    assert n == 0 and ms == 0
    b.timing(False)
    b.run(4096)
    assert b.timing_read()[1] == 0
    b.close()


def test_a_batch_freed_inside_a_collect_leaves_the_callers_arrays_alone(gpu):
    """cmhip_batch_free() between cmhip_batch_vu_collect_begin() and _end(): the helpers are waited for, nothing
    is finished into the caller's arrays at free time (they may be gone by then) -- both forms, the helper pool
    (512 streams and more) and the deferred one below that."""
    cm = gpu
    for S in (8, 1024):
        b = cm.Batch(S, 1, 256, flags=cm.VU)
        b.generate(cm.GEN_NOISE, 5, 256)
        b.run(256)
        b.vu_snapshot()
        results = (cm.VuResult * S)()
        rcs = (cm.C.c_int * S)(*([77] * S))
        b.vu_collect_begin(results, rcs)
        b.close()
        if S < 512:                                  # the deferred form never ran: the arrays are as the caller left them
            assert list(rcs) == [77] * S
