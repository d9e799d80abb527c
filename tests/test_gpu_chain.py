"""The per-stream operator API (coolmic_* names of the reference) running on the GPU:
snddev -> transform -> vumeter wired through coolmic_iohandle_t exactly as
src/simple.c:198-229 wires them, checked against SURVEY 8(c) and the oracle."""
import math

import numpy as np
import pytest

from oracle import oracle_ffi as of

pytestmark = pytest.mark.gpu


def _pow(x):
    return -math.inf if x == "-inf" else float(x)


def _chain(cm, src_handle, channels, rate=48000):
    tr = cm.Transform(rate, channels)
    assert tr.attach(src_handle) == 0
    src_handle.unref()                       # attach-then-unref idiom (ref: src/simple.c:212-229)
    h = tr.get_iohandle()
    vu = cm.Vumeter(rate, channels)
    assert vu.attach(h) == 0
    return tr, h, vu


def test_config1_sine_chain_G1_G2_G3(gpu, golden):
    cm = gpu
    dev = cm.Snddev("sine", 48000, 1)
    tr, h, vu = _chain(cm, dev.get_iohandle(), 1)
    dev.unref()
    for name in ("G1", "G2", "G3"):
        case = golden["cases"][name]
        g = case["gain"]
        assert tr.set_master_gain(g["channels"], g["scale"], g["gain"]) == 0
        for _ in range(case["reads"]):
            assert vu.read(-1) == 1024
        rc, r = vu.result()
        assert rc == 0
        exp = case["vu"]
        assert r.frames == exp["frames"] and r.global_peak == exp["global_peak"]
        assert r.global_power == exp["global_power"]
        assert r.rate == 48000 and r.channels == 1
    rc, _ = vu.result()
    assert rc == cm.ERROR_INVAL
    h.unref(); vu.unref(); tr.unref()


def test_null_chain_G5(gpu, golden):
    cm = gpu
    dev = cm.Snddev("null", 48000, 2)
    vu = cm.Vumeter(48000, 2)
    h = dev.get_iohandle()
    assert vu.attach(h) == 0
    h.unref(); dev.unref()
    assert vu.read(-1) == 1024
    rc, r = vu.result()
    assert rc == 0 and r.frames == 256 and r.global_peak == 0
    assert r.global_power == -math.inf and r.channel_power[1] == -math.inf
    vu.unref()


@pytest.mark.parametrize("name", ["K1", "K2", "K3", "K4", "K5", "K9"])
def test_known_answers_through_handles(gpu, golden, name):
    cm = gpu
    case = golden["cases"][name]
    x = np.array(golden[case["input"]], dtype=np.int16)
    tr, h, vu = _chain(cm, cm.IoHandle.from_bytes(x.tobytes()), case["channels"])
    g = case["gain"]
    assert tr.set_master_gain(g["channels"], g["scale"], g["gain"]) == 0
    n, data = h.read(x.nbytes)
    assert n == x.nbytes and np.frombuffer(data, np.int16).tolist() == case["pcm"]
    # same data again through the meter
    tr2, h2, vu2 = _chain(cm, cm.IoHandle.from_bytes(x.tobytes()), case["channels"])
    tr2.set_master_gain(g["channels"], g["scale"], g["gain"])
    assert vu2.read(-1) == x.nbytes
    rc, r = vu2.result()
    exp = case["vu"]
    assert rc == 0 and r.global_power == _pow(exp["global_power"])
    if "global_peak" in exp:
        assert r.global_peak == exp["global_peak"]
    for i, p in enumerate(exp.get("channel_peak", [])):
        assert r.channel_peak[i] == p
    for i, p in enumerate(exp.get("channel_power", [])):
        assert r.channel_power[i] == _pow(p)
    for o in (h, vu, tr, h2, vu2, tr2):
        o.unref()


def test_K6_K7_K8_framing(gpu, golden):
    cm = gpu
    x = np.array(golden["k_input_stereo"], dtype=np.int16)
    tr, h, vu = _chain(cm, cm.IoHandle.from_bytes(x.tobytes()), 2)
    assert tr.set_master_gain(3, 1000, [1, 2, 3]) == -10           # K6
    assert tr.set_master_gain(2, 1000, [1000, 1000]) == 0
    n, data = h.read(7)                                            # K7
    assert n == 4 and np.frombuffer(data, np.int16).tolist() == [5, -7] and h.eof() == 0
    for o in (h, vu, tr):
        o.unref()
    tr, h, vu = _chain(cm, cm.IoHandle.from_bytes(x.tobytes(), chunk=3), 2)   # K8
    assert tr.set_master_gain(2, 1000, [1000, 1000]) == 0
    n, data = h.read(16)
    assert n == 16 and np.frombuffer(data, np.int16).tolist() == x.tolist()
    for o in (h, vu, tr):
        o.unref()


def test_chain_matches_oracle_on_noise_with_map(gpu, oracle):
    cm = gpu
    C, frames = 2, 20000
    x = oracle.lcg(2024, frames * C)
    tr, h, vu = _chain(cm, cm.IoHandle.from_bytes(x.tobytes(), chunk=1000), C)
    assert tr.set_master_gain(2, 1000, [750, 1250]) == 0
    assert tr.set_channel_map([1, 0]) == 0
    total = 0
    while True:
        n = vu.read(-1)
        if n <= 0:
            break
        total += n
    assert total == x.nbytes
    rc, r = vu.result()
    _, g = oracle.gain(C, 2, 1000, [750, 1250])
    want = oracle.gain_apply(g, oracle.chmap([1, 0], x, C), C)
    v = oracle.vu_new(C)
    oracle.vu_accumulate(v, want)
    _, ro = oracle.vu_result(v)
    assert rc == 0 and r.as_dict() == of.vu_result_dict(ro)
    for o in (h, vu, tr):
        o.unref()


def _tee_chain(cm, x, C, chunk=4096, gain=(2, 1000, [750, 1250])):
    """source -> transform -> tee -> {reader 0 (the encoder branch), vumeter on reader 1}, in the order
    ref: src/simple.c:212-229 attaches them"""
    src = cm.IoHandle.from_bytes(x.tobytes(), chunk=chunk)
    tr = cm.Transform(48000, C)
    assert tr.attach(src) == 0
    src.unref()
    if gain:
        assert tr.set_master_gain(*gain) == 0
    tee = cm.Tee(2)
    h = tr.get_iohandle()
    assert tee.attach(h) == 0
    h.unref()
    enc_in = tee.get_iohandle(0)
    vu = cm.Vumeter(48000, C)
    h = tee.get_iohandle(1)
    assert vu.attach(h) == 0
    h.unref()
    return tr, tee, enc_in, vu


def test_product_wiring_with_tee(gpu, oracle):
    """snddev -> transform -> tee -> {encoder branch, vumeter}, as ref: src/simple.c:183-236
    wires it; the encoder branch is played by a reader pulling 1024 bytes at a time
    (ref: src/enc_vorbis.c:91).  The meter behind the tee shares the transform's launches (window
    records, vumeter.c): ONE launch per 1024-byte pull for both branches, results as the oracle's, a
    result every 20 reads as the product takes them (ref: src/simple.c:370,486-491)."""
    cm = gpu
    C, frames = 2, 30000
    x = oracle.lcg(77, frames * C)
    tr, tee, enc_in, vu = _tee_chain(cm, x, C)
    assert vu.mode() == 2                                  # through the tee, by records
    _, g = oracle.gain(C, 2, 1000, [750, 1250])
    want = oracle.gain_apply(g, x, C)
    pcm = b""
    runs0 = cm.lib.cmhip_debug_run_count()
    pulls = reads = seen = 0
    v = oracle.vu_new(C)
    for _ in range(100000):
        n, d = enc_in.read(1024)
        pcm += d
        pulls += 1 if n > 0 else 0
        m = vu.read(-1)
        if m > 0:
            oracle.vu_accumulate(v, want[seen // 2: (seen + m) // 2])
            seen += m
            reads += 1
            if reads % 20 == 0:
                rc, r = vu.result()
                _, ro = oracle.vu_result(v)
                assert rc == 0 and r.as_dict() == of.vu_result_dict(ro), reads
                v = oracle.vu_new(C)
        if n == 0 and m <= 0 and enc_in.eof() == 1:
            break
    assert np.array_equal(np.frombuffer(pcm, np.int16), want)
    assert seen == x.nbytes and vu.mode() == 2
    # both branches served by the transform's launches alone: one per pull of the tee, none of the meter's
    assert cm.lib.cmhip_debug_run_count() - runs0 == pulls
    rc, r = vu.result()
    _, ro = oracle.vu_result(v)
    assert rc == 0 and r.as_dict() == of.vu_result_dict(ro)
    assert vu.result()[0] == cm.ERROR_INVAL
    for o in (enc_in, vu, tee, tr):
        o.unref()


def test_a_long_window_behind_a_tee_outlives_the_record_ring(gpu, oracle):
    """700 pulls of 256 bytes without a result(): more launches than the transform's ring of 256 device windows
    holds (the transform fetches them to the host before a slot comes round again) and more whole records than
    the meter keeps unmerged (512: it folds them into its window on the way).  One window over all of it."""
    cm = gpu
    C, pulls = 1, 700
    x = oracle.lcg(4242, pulls * 128)
    tr, tee, enc_in, vu = _tee_chain(cm, x, C, chunk=1000, gain=(1, 1000, [1100]))
    assert vu.mode() == 2
    _, g = oracle.gain(C, 1, 1000, [1100])
    want = oracle.gain_apply(g, x, C)
    runs0 = cm.lib.cmhip_debug_run_count()
    for i in range(pulls):
        n, d = enc_in.read(256)
        assert n == 256
        assert vu.read(256) == 256
    assert cm.lib.cmhip_debug_run_count() - runs0 == pulls and vu.mode() == 2
    rc, r = vu.result()
    v = oracle.vu_new(C)
    oracle.vu_accumulate(v, want)
    _, ro = oracle.vu_result(v)
    assert rc == 0 and r.as_dict() == of.vu_result_dict(ro)
    for o in (enc_in, vu, tee, tr):
        o.unref()


@pytest.mark.parametrize("who_leads", ["encoder", "meter", "mixed"])
def test_meter_behind_a_tee_lags_and_cuts_blocks(gpu, oracle, who_leads):
    """The meter's reads do not line up with the transform's launches: the encoder branch pulls odd sizes
    and runs up to the tee's 8 KiB ahead (or the meter leads and its 1024-byte reads drive the launches),
    result() and reset() fall inside launches' blocks, reads of a few bytes leave partial frames in the
    meter's buffer.  Every window against the oracle's, whatever mixture of whole records and cut blocks
    it was put together from."""
    cm = gpu
    C, frames = 2, 40000
    x = oracle.lcg(123, frames * C)
    tr, tee, enc_in, vu = _tee_chain(cm, x, C, chunk=3000, gain=(2, 1000, [1250, 600]))
    assert vu.mode() == 2
    _, g = oracle.gain(C, 2, 1000, [1250, 600])
    want = oracle.gain_apply(g, x, C)
    rng = np.random.default_rng({"encoder": 1, "meter": 2, "mixed": 3}[who_leads])
    enc_sizes = [1024, 4096, 300, 8192, 20, 2048]
    vu_sizes = [-1, -1, 7, 500, -1, 3, 1000, -1]
    enc_got = vu_got = 0            # bytes each branch has taken
    acc = 0                         # meter bytes accounted (whole frames) in v
    v = oracle.vu_new(C)
    pcm = b""
    step = 0
    windows = cuts = 0
    while True:
        step += 1
        lead = who_leads if who_leads != "mixed" else ("encoder" if (step // 7) % 2 else "meter")
        n = m = 0
        if lead == "encoder" or step % 3 == 0:
            n, d = enc_in.read(enc_sizes[step % len(enc_sizes)])
            pcm += d
            enc_got += max(n, 0)
        k = 1 if lead == "encoder" else 3
        for i in range(k):
            m = vu.read(vu_sizes[(step + i) % len(vu_sizes)])
            assert m >= 0
            vu_got += m
        whole = vu_got - vu_got % (2 * C)          # the meter accounts whole frames, keeps the rest
        if whole > acc:
            oracle.vu_accumulate(v, want[acc // 2: whole // 2])
            acc = whole
        r_ = rng.random()
        if r_ < 0.12:
            rc, r = vu.result()
            rc_o, ro = oracle.vu_result(v)
            assert rc == rc_o, (step, rc, rc_o)
            if rc == 0:
                assert r.as_dict() == of.vu_result_dict(ro), (step, who_leads)
                windows += 1
            v = oracle.vu_new(C)
        elif r_ < 0.20:
            assert vu.reset() == 0
            v = oracle.vu_new(C)
            cuts += 1
        if enc_got >= x.nbytes and vu_got >= x.nbytes:
            break
        if step > 20000:
            raise AssertionError("no progress")
        if lead == "meter" and enc_got + 8192 < vu_got:   # the tee holds 8 KiB for the slower reader
            n, d = enc_in.read(4096)
            pcm += d
            enc_got += max(n, 0)
    assert vu.mode() == 2 and windows > 8 and cuts > 1
    while enc_got < x.nbytes:
        n, d = enc_in.read(4096)
        assert n > 0
        pcm += d
        enc_got += n
    assert np.array_equal(np.frombuffer(pcm, np.int16), want)
    rc, r = vu.result()
    rc_o, ro = oracle.vu_result(v)
    assert rc == rc_o and (rc != 0 or r.as_dict() == of.vu_result_dict(ro))
    for o in (enc_in, vu, tee, tr):
        o.unref()


def test_meter_joins_a_running_tee_and_survives_a_second_reader_of_the_transform(gpu, oracle):
    """(a) The meter is attached when the encoder branch has already pulled for a while and the tee holds
    bytes no record covers: they go through a launch of the meter's own, the rest by records.  (b) Then
    somebody reads the transform's handle past the tee: the tee's bytes no longer continue the transform's
    output, the meter notices and goes back to a batch of its own without losing a frame of its window.
    (c) A tee whose transform is attached after the meter got the tee's handle is looked at again at the
    first read."""
    cm = gpu
    C, frames = 1, 30000
    x = oracle.lcg(9, frames * C)
    src = cm.IoHandle.from_bytes(x.tobytes(), chunk=2000)
    tr = cm.Transform(48000, C)
    assert tr.attach(src) == 0
    src.unref()
    assert tr.set_master_gain(1, 1000, [800]) == 0
    _, g = oracle.gain(C, 1, 1000, [800])
    want = oracle.gain_apply(g, x, C)
    tee = cm.Tee(2)
    th = tr.get_iohandle()
    assert tee.attach(th) == 0
    enc_in = tee.get_iohandle(0)
    # 3 KiB in the tee before any meter exists (one read: the tee sizes its buffer by the request and holds
    # everything for the reader that has not started, ref: src/tee.c:83-135)
    n, d = enc_in.read(3072)
    assert n == 3072 and np.array_equal(np.frombuffer(d, np.int16), want[: n // 2])
    pos_e = n
    vu = cm.Vumeter(48000, C)
    h = tee.get_iohandle(1)
    assert vu.attach(h) == 0
    h.unref()
    assert vu.mode() == 2
    v = oracle.vu_new(C)
    pos_v = 0
    for i in range(12):
        if i % 2:
            n, d = enc_in.read(1024)
            pos_e += n
        m = vu.read(-1)
        assert m > 0
        oracle.vu_accumulate(v, want[pos_v // 2: (pos_v + m) // 2])
        pos_v += m
    rc, r = vu.result()
    _, ro = oracle.vu_result(v)
    assert rc == 0 and r.as_dict() == of.vu_result_dict(ro)
    # (b) a second consumer of the transform's own handle: the bytes it takes never reach the tee
    v = oracle.vu_new(C)
    stream = max(pos_e, pos_v)                           # what the transform has produced = what the tee pulled
    for i in range(4):
        pos_e += enc_in.read(1024)[0]                    # (the tee stalls a reader that runs its buffer ahead)
        m = vu.read(-1)
        oracle.vu_accumulate(v, want[pos_v // 2: (pos_v + m) // 2])
        pos_v += m
    stream = max(stream, pos_v, pos_e)
    n, d = th.read(600)
    assert n == 600 and np.array_equal(np.frombuffer(d, np.int16), want[stream // 2: stream // 2 + 300])
    # from here the tee's stream is the transform's output without those 600 bytes
    rest = np.concatenate([want[: stream // 2], want[stream // 2 + 300:]])
    for i in range(10):
        pos_e += enc_in.read(1024)[0]
        m = vu.read(-1)
        assert m > 0
        oracle.vu_accumulate(v, rest[pos_v // 2: (pos_v + m) // 2])
        pos_v += m
    assert vu.mode() == 0                                # noticed, and back on a batch of its own
    rc, r = vu.result()
    _, ro = oracle.vu_result(v)
    assert rc == 0 and r.as_dict() == of.vu_result_dict(ro)
    for o in (th, enc_in, vu, tee, tr):
        o.unref()
    # (c) meter first, transform under the tee later
    y = oracle.lcg(10, 4000)
    tee = cm.Tee(1)
    vu = cm.Vumeter(48000, 1)
    h = tee.get_iohandle(0)
    assert vu.attach(h) == 0 and vu.mode() == 0
    h.unref()
    src = cm.IoHandle.from_bytes(y.tobytes())
    tr = cm.Transform(48000, 1)
    assert tr.attach(src) == 0 and tr.set_master_gain(1, 2, [1]) == 0
    src.unref()
    th = tr.get_iohandle()
    assert tee.attach(th) == 0
    th.unref()
    v = oracle.vu_new(1)
    while True:
        m = vu.read(-1)
        if m <= 0:
            break
        assert vu.mode() == 2
    _, g = oracle.gain(1, 1, 2, [1])
    oracle.vu_accumulate(v, oracle.gain_apply(g, y, 1))
    rc, r = vu.result()
    _, ro = oracle.vu_result(v)
    assert rc == 0 and r.as_dict() == of.vu_result_dict(ro)
    for o in (vu, tee, tr):
        o.unref()


def _eq_expect(oracle, cm, x, C, coef, scale, gains, cmap):
    """per channel: map -> gain -> the mono oracle filter; interleaved int16 back"""
    q = (of.Biquad * 3)()
    for i in range(3):
        q[i].b0, q[i].b1, q[i].b2, q[i].a1, q[i].a2 = [float(v) for v in coef[5 * i:5 * i + 5]]
    frames = x.size // C
    out = np.empty((frames, C), dtype=np.int16)
    for c in range(C):
        src = c if cmap is None else cmap[c]
        rc, g = oracle.gain(1, 1, scale, [gains[c]])
        assert rc == 0
        st = np.zeros(12, dtype=np.float32)
        _, oi = oracle.eq_run_mono(g, q, 3, st, x.reshape(-1, C)[:, src].copy())
        out[:, c] = oi
    return out.reshape(-1)


def test_transform_equaliser_behind_the_handle(gpu, oracle):
    """coolmic_transform_set_eq(): the IIR of the transform stage behind the pull API -- a
    stereo stream read in odd pieces (every read is a launch of its own, the filter state
    carries over), then the filter switched off in mid-stream"""
    cm = gpu
    C, frames = 2, 30000
    x = oracle.lcg(77, frames * C)
    coef = cm.eq3(48000.0)
    tr = cm.Transform(48000, C)
    src = cm.IoHandle.from_bytes(x.tobytes(), chunk=1500)
    assert tr.attach(src) == 0
    src.unref()
    h = tr.get_iohandle()
    assert tr.set_master_gain(2, 1000, [800, 1100]) == 0
    assert tr.set_channel_map([1, 0]) == 0
    assert tr.set_eq(coef) == 0
    assert tr.set_eq(np.zeros(25, np.float32)) == cm.ERROR_INVAL      # five sections: too many
    got = b""
    first = 20000 * 2 * C                       # bytes with the filter on
    sizes = [4096, 10, 2 * C * 333, 65536, 2 * C]
    i = 0
    while len(got) < first:
        n, data = h.read(min(sizes[i % len(sizes)], first - len(got)))
        i += 1
        assert n > 0
        got += data
    assert tr.set_eq(None) == 0
    while True:
        n, data = h.read(8192)
        if n <= 0:
            break
        got += data
    assert len(got) == x.nbytes
    res = np.frombuffer(got, np.int16)
    want_eq = _eq_expect(oracle, cm, x[: 20000 * C], C, coef, 1000, [800, 1100], [1, 0])
    assert np.array_equal(res[: 20000 * C], want_eq)
    _, g = oracle.gain(C, 2, 1000, [800, 1100])
    want_plain = oracle.gain_apply(g, oracle.chmap([1, 0], x[20000 * C:], C), C)
    assert np.array_equal(res[20000 * C:], want_plain)
    h.unref(); tr.unref()


def test_equaliser_off_and_on_again_starts_from_silence(gpu, oracle):
    """set_eq(A), read, set_eq(off), read, set_eq(B), read -- with no gain and no map, so that the
    middle read takes the reference's early-out (ref: src/transform.c:107-108) and never touches
    the device: "0 sections switches the filter off and clears its state"
    (include/coolmic-dsp/transform.h), so filter B must start from zero state, not from what A
    left behind"""
    cm = gpu
    C, n = 1, 6000
    x = oracle.lcg(4711, 3 * n)
    coef_a = cm.eq3(48000.0)
    coef_b = np.concatenate([cm.design_biquad(1, 48000.0, 2500.0, 5.0, 1.5), coef_a[5:10], coef_a[:5]])
    tr = cm.Transform(48000, C)
    src = cm.IoHandle.from_bytes(x.tobytes(), chunk=1024)
    assert tr.attach(src) == 0
    src.unref()
    h = tr.get_iohandle()

    def pull(nbytes):
        got = b""
        while len(got) < nbytes:
            k, data = h.read(min(4096, nbytes - len(got)))
            assert k > 0
            got += data
        return np.frombuffer(got, np.int16)

    assert tr.set_eq(coef_a) == 0
    a = pull(2 * n)
    assert tr.set_eq(None) == 0
    mid = pull(2 * n)
    assert tr.set_eq(coef_b) == 0
    b = pull(2 * n)
    assert np.array_equal(a, _eq_expect(oracle, cm, x[:n], C, coef_a, 1, [1], None))
    assert np.array_equal(mid, x[n:2 * n])                      # untouched
    assert np.array_equal(b, _eq_expect(oracle, cm, x[2 * n:], C, coef_b, 1, [1], None))   # zero state
    h.unref(); tr.unref()


def test_meter_attached_to_a_transform_that_has_been_running(gpu, oracle):
    """frames have flowed through the transform (and its device batch exists, with a partial frame carried)
    before a meter is attached directly to its handle: the meter's window starts empty at the attachment,
    shares the transform's launches from there (odd-sized upstream pieces: the carry path), and a re-attachment
    in the middle of a window keeps the frames counted so far, as the reference's attach does
    (ref: src/vumeter.c:101-110 touches nothing but the handle)"""
    cm = gpu
    C, frames = 2, 6000
    x = oracle.lcg(77, frames * C)
    src = cm.IoHandle.from_bytes(x.tobytes(), chunk=333)          # never a whole number of frames
    tr = cm.Transform(48000, C)
    assert tr.attach(src) == 0
    src.unref()
    assert tr.set_master_gain(2, 1000, [1500, 500]) == 0
    _, g = oracle.gain(C, 2, 1000, [1500, 500])
    want = oracle.gain_apply(g, x, C)
    h = tr.get_iohandle()
    n, d = h.read(1000)                                           # somebody reads before any meter exists
    assert n == 1000 and np.array_equal(np.frombuffer(d, np.int16), want[:500])
    pos = 1000
    vu = cm.Vumeter(48000, C)
    assert vu.attach(h) == 0 and vu.mode() == 1
    v = oracle.vu_new(C)
    for size in (-1, 100, 7, -1, 1024):
        m = vu.read(size)
        assert m >= 0 and m % (2 * C) == 0
        oracle.vu_accumulate(v, want[pos // 2: (pos + m) // 2])
        pos += m
    # re-attach in the middle of the window: to a tee's reader handle over the same transform
    tee = cm.Tee(1)
    assert tee.attach(h) == 0
    th = tee.get_iohandle(0)
    assert vu.attach(th) == 0
    th.unref()
    for size in (-1, -1, 300):
        m = vu.read(size)
        assert m > 0
        oracle.vu_accumulate(v, want[pos // 2: (pos + m) // 2])
        pos += m
    rc, r = vu.result()
    _, ro = oracle.vu_result(v)
    assert rc == 0 and r.as_dict() == of.vu_result_dict(ro)
    for o in (h, vu, tee, tr):
        o.unref()


def test_meter_directly_on_a_transform_shares_its_launch(gpu, oracle):
    """A VU meter attached straight to a transform's handle lets the transform's launch accumulate
    the window (vumeter.c / transform.c, coolmic_transform_fuse_vu).  What the meter reports must
    not change: windows cut by result() calls between reads of odd sizes, reset(), gain changes
    between windows, gain off (the reference's early-out: the fused launch still has to run for
    the window), then the meter moved to a plain source (its own batch again)."""
    cm = gpu
    C, frames = 2, 9000
    x = oracle.lcg(31, frames * C)
    tr, h, vu = _chain(cm, cm.IoHandle.from_bytes(x.tobytes(), chunk=700), C)
    pos = 0

    def window(nreads, sizes, gain):
        nonlocal pos
        v = oracle.vu_new(C)
        rcg, g = oracle.gain(C, *gain) if gain else (0, of.Gain())
        for i in range(nreads):
            n = vu.read(sizes[i % len(sizes)])
            assert n >= 0 and n % (2 * C) == 0
            blk = x[pos // 2: (pos + n) // 2]
            oracle.vu_accumulate(v, oracle.gain_apply(g, blk, C))
            pos += n
        return v

    for gain, sizes, nreads in (((2, 1000, [750, 1250]), [-1, 7, 1000, 3, 4], 9),
                                (None, [-1, 512], 5),                 # gain off: early-out in the reference
                                ((1, 3, [2]), [100, -1], 6)):
        assert tr.set_master_gain(*(gain if gain else (0, 0, None))) == 0
        v = window(nreads, sizes, gain)
        # someone else reads the transform's handle in between: transformed PCM for them, and none of
        # it in the meter's window (the meter arms the window only around its own reads)
        n, data = h.read(120)
        rcg, g = oracle.gain(C, *gain) if gain else (0, of.Gain())
        assert n == 120 and np.array_equal(np.frombuffer(data, np.int16), oracle.gain_apply(g, x[pos // 2: pos // 2 + 60], C))
        pos += n
        rc, r = vu.result()
        rc_o, r_o = oracle.vu_result(v)
        assert rc == rc_o == 0 and r.as_dict() == of.vu_result_dict(r_o), gain
    rc, _ = vu.result()
    assert rc == cm.ERROR_INVAL                      # nothing since the last result
    window(2, [-1], (1, 3, [2]))
    assert vu.reset() == 0                           # drops the frames of those two reads
    v = window(3, [-1, 64], (1, 3, [2]))
    rc, r = vu.result()
    _, r_o = oracle.vu_result(v)
    assert rc == 0 and r.as_dict() == of.vu_result_dict(r_o)
    # the meter moves to a plain handle: a batch of its own from here on
    y = oracle.lcg(32, 3000 * C)
    src = cm.IoHandle.from_bytes(y.tobytes())
    assert vu.attach(src) == 0
    src.unref()
    while vu.read(-1) > 0:
        pass
    v = oracle.vu_new(C)
    oracle.vu_accumulate(v, y)
    rc, r = vu.result()
    _, r_o = oracle.vu_result(v)
    assert rc == 0 and r.as_dict() == of.vu_result_dict(r_o)
    # ... and the transform's handle still delivers transformed PCM to whoever reads it
    _, g = oracle.gain(C, 1, 3, [2])
    n, data = h.read(400)
    assert n == 400 and np.array_equal(np.frombuffer(data, np.int16), oracle.gain_apply(g, x[pos // 2: pos // 2 + 200], C))
    for o in (h, vu, tr):
        o.unref()
