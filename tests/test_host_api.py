"""CPU: host logic of the drop-in boundary -- handles, refcounts, sources, framing,
parameter rules, division constants -- everything that needs no arithmetic on PCM.
The arithmetic itself only exists on the GPU (tests/test_gpu_*.py)."""
import ctypes as C

import numpy as np
import pytest


def test_iohandle_contract(cm):
    # ref: src/iohandle.c:54-113
    assert not cm.lib.coolmic_iohandle_new(None, None, None, cm.FREE_FN(), cm.READ_FN(), cm.EOF_FN())
    buf = (C.c_ubyte * 8)()
    assert cm.lib.coolmic_iohandle_read(None, buf, 8) == cm.ERROR_FAULT
    assert cm.lib.coolmic_iohandle_eof(None) == cm.ERROR_FAULT

    script = [b"ab", b"cde", 0, b"zz"]
    h = cm.IoHandle.from_callbacks(lambda n: script.pop(0))
    assert cm.lib.coolmic_iohandle_read(h.ptr, None, 8) == cm.ERROR_FAULT
    assert cm.lib.coolmic_iohandle_read(h.ptr, buf, 0) == 0
    n, data = h.read(8)                  # loops until a 0 comes back
    assert (n, data) == (5, b"abcde")
    assert h.eof() == 0                  # no eof callback: endless
    h.unref()

    script = [b"xy", -1]
    h = cm.IoHandle.from_callbacks(lambda n: script.pop(0))
    assert h.read(8) == (2, b"xy")       # error after data: the data
    script[:] = [-1]
    assert h.read(8)[0] == -1            # error with nothing: the error
    script[:] = [-9]
    assert h.read(8)[0] == -9
    h.unref()

    h = cm.IoHandle.from_bytes(b"0123456789", chunk=3)
    assert h.read(4) == (4, b"0123") and h.eof() == 0
    assert h.read(100) == (6, b"456789") and h.eof() == 1
    h.unref()


def test_refcounting_and_ownership(cm):
    # ref: src/transform.c:83-99,181-193 and the attach-then-unref idiom of src/simple.c:212-229
    freed = []
    src = cm.IoHandle.from_callbacks(lambda n: 0, free=lambda: freed.append("src"))
    assert src.refcount() == 1
    tr = cm.Transform(48000, 2)
    assert tr.attach(src) == 0 and src.refcount() == 2
    src_ptr = src.ptr
    src.unref()                                   # the transform now holds the only reference
    assert cm.lib.coolmic_ro_refcount(src_ptr) == 1 and not freed
    h = tr.get_iohandle()
    assert tr.refcount() == 2                     # the handle keeps its transform alive
    tr_ptr = tr.ptr
    tr.unref()
    assert cm.lib.coolmic_ro_refcount(tr_ptr) == 1
    h.unref()                                     # last handle -> transform -> source
    assert freed == ["src"]
    # NULL is a reported, harmless error
    assert cm.lib.coolmic_ro_ref(None) == cm.ERROR_FAULT
    assert cm.lib.coolmic_ro_unref(None) == cm.ERROR_FAULT
    # re-attaching replaces and releases the previous handle; NULL detaches
    a = cm.IoHandle.from_callbacks(lambda n: 0, free=lambda: freed.append("a"))
    b = cm.IoHandle.from_callbacks(lambda n: 0, free=lambda: freed.append("b"))
    vu = cm.Vumeter(48000, 1)
    assert vu.attach(a) == 0 and vu.attach(b) == 0
    a.unref()
    assert freed == ["src", "a"]
    assert vu.attach(None) == 0
    b.unref()
    assert freed == ["src", "a", "b"]
    vu.unref()
    assert cm.lib.coolmic_transform_attach_iohandle(None, None) == cm.ERROR_FAULT
    assert cm.lib.coolmic_vumeter_attach_iohandle(None, None) == cm.ERROR_FAULT


def test_constructors_reject_bad_formats(cm):
    for rate, ch in ((0, 1), (48000, 0), (48000, 17)):
        assert not cm.lib.coolmic_transform_new(None, None, rate, ch)
        assert not cm.lib.coolmic_vumeter_new(None, None, rate, ch)
    # ref: src/snddev.c:104-129, src/snddev_sine.c:172-177
    new = cm.lib.coolmic_snddev_new
    assert not new(None, None, b"sine", None, 48000, 2, 1, -1)      # sine is mono only
    assert not new(None, None, b"sine", None, 22050, 1, 1, -1)      # rate without a table
    assert not new(None, None, b"alsa", None, 48000, 1, 1, -1)      # unknown driver
    assert not new(None, None, b"null", None, 48000, 1, 0, -1)      # no flags
    d = new(None, None, b"SINE", None, 48000, 1, 1, -1)             # case-insensitive
    assert d
    cm.lib.coolmic_ro_unref(d)
    d = new(None, None, None, None, 48000, 2, 1, -1)                # AUTO
    assert d
    cm.lib.coolmic_ro_unref(d)


def test_sources_match_oracle_sources(cm, oracle):
    from oracle import oracle_ffi as of
    for rate in (8000, 16000, 24000, 32000, 44000, 44100, 48000, 96000):
        rc, t = cm.sine_period(rate)
        rco, to = oracle.sine_table(rate)
        assert rc == rco == 0 and np.array_equal(t, to)
    assert cm.sine_period(11025)[0] == cm.ERROR_NOSYS
    # byte-granular phase across odd-sized reads (ref: src/snddev_sine.c:118-150)
    dev = cm.Snddev("sine", 48000, 1)
    h = dev.get_iohandle()
    sine = of.Sine()
    oracle.lib.oracle_sine_init(C.byref(sine), 48000)
    for n in (1, 3, 95, 96, 97, 1024, 5, 200):
        got = h.read(n)
        buf = (C.c_ubyte * n)()
        assert oracle.lib.oracle_sine_read(C.byref(sine), buf, n) == n
        assert got == (n, bytes(buf))
    assert h.eof() == 0
    h.unref(); dev.unref()
    dev = cm.Snddev("null", 48000, 2)
    h = dev.get_iohandle()
    assert h.read(33) == (33, bytes(33))
    h.unref(); dev.unref()


def test_transform_framing_without_arithmetic(cm, golden):
    """gain disabled is the reference's own early-out (ref: src/transform.c:107-108): the
    framing logic can be exercised on the CPU box.  K7, K8, K9 shapes."""
    x = np.array(golden["k_input_stereo"], dtype=np.int16).tobytes()
    tr = cm.Transform(48000, 2)
    tr.attach(cm.IoHandle.from_bytes(x))
    h = tr.get_iohandle()
    assert h.read(3) == (0, b"")                    # less than a frame asked: nothing
    n, data = h.read(7)                             # K7: cut to one frame
    assert (n, data) == (4, x[:4]) and h.eof() == 0
    assert h.read(1000) == (12, x[4:]) and h.eof() == 1
    h.unref()
    # K8: upstream in 3-byte pieces, partial frames carried between reads
    tr = cm.Transform(48000, 2)
    tr.attach(cm.IoHandle.from_bytes(x, chunk=3))
    h = tr.get_iohandle()
    assert h.read(16) == (16, x)
    h.unref()
    # a short upstream read that ends inside a frame: the tail waits in the carry buffer
    script = [x[:6], 0, 0, x[6:16], 0]   # the consumer-side loop asks again after a short read
    tr = cm.Transform(48000, 2)
    tr.attach(cm.IoHandle.from_callbacks(lambda n: script.pop(0) if script else 0))
    h = tr.get_iohandle()
    assert h.read(16) == (4, x[:4])
    assert h.read(16) == (12, x[4:16])
    h.unref()
    # no upstream at all: FAULT from the handle is swallowed into an empty read, eof is 1
    tr = cm.Transform(48000, 2)
    h = tr.get_iohandle()
    assert h.read(16) == (0, b"") and h.eof() == 1
    h.unref(); tr.unref()
    # K9: mono, scale 0 -> PCM unchanged
    m = np.array(golden["k_input_mono"], dtype=np.int16).tobytes()
    tr = cm.Transform(48000, 1)
    tr.attach(cm.IoHandle.from_bytes(m))
    assert tr.set_master_gain(1, 0, [123]) == 0
    h = tr.get_iohandle()
    assert h.read(len(m)) == (len(m), m)
    h.unref()


def test_set_master_gain_rules(cm):
    # ref: src/transform.c:195-222
    assert cm.lib.coolmic_transform_set_master_gain(None, 1, 1, None) == cm.ERROR_FAULT
    st = cm.Transform(48000, 2)
    assert st.set_master_gain(2, 1000, [750, 1250]) == 0
    assert st.set_master_gain(1, 1000, [900]) == 0           # broadcast
    assert st.set_master_gain(3, 1000, [1, 2, 3]) == cm.ERROR_INVAL
    assert st.set_master_gain(0, 1000, [1]) == 0             # disable
    assert st.set_master_gain(2, 0, [1, 1]) == 0             # disable
    assert st.set_master_gain(2, 1000, None) == 0            # disable
    mono = cm.Transform(48000, 1)
    assert mono.set_master_gain(2, 1, [3, 4]) == 0           # stereo pair on mono: mean
    assert mono.set_master_gain(3, 1, [3, 4, 5]) == cm.ERROR_INVAL
    assert st.set_channel_map([1, 0]) == 0
    assert st.set_channel_map([0, 2]) == cm.ERROR_INVAL
    assert st.set_channel_map(None) == 0
    # the equaliser setter (an addition): argument rules only, no device needed
    assert cm.lib.coolmic_transform_set_eq(None, 0, None) == cm.ERROR_FAULT
    assert st.set_eq(np.zeros(15, np.float32)) == 0
    assert st.set_eq(np.zeros(20, np.float32)) == 0
    assert st.set_eq(np.zeros(25, np.float32)) == cm.ERROR_INVAL       # more than four sections
    assert cm.lib.coolmic_transform_set_eq(st.ptr, 2, None) == cm.ERROR_INVAL
    assert st.set_eq(None) == 0
    st.unref(); mono.unref()


def test_vumeter_host_rules(cm):
    # ref: src/vumeter.c:148-151,195-199
    assert cm.lib.coolmic_vumeter_read(None, -1) == -1
    assert cm.lib.coolmic_vumeter_reset(None) == cm.ERROR_FAULT
    vu = cm.Vumeter(48000, 2)
    r = cm.VuResult()
    assert cm.lib.coolmic_vumeter_result(None, C.byref(r)) == cm.ERROR_FAULT
    assert cm.lib.coolmic_vumeter_result(vu.ptr, None) == cm.ERROR_FAULT
    assert vu.result()[0] == cm.ERROR_INVAL                  # no frames yet
    assert vu.reset() == 0
    # upstream error with nothing buffered surfaces as -1; nothing to read gives 0
    script = [-1, 0]
    h = cm.IoHandle.from_callbacks(lambda n: script.pop(0))
    vu.attach(h)
    assert vu.read(-1) == -1
    assert vu.read(-1) == 0
    # a partial frame is only buffered (no arithmetic yet): 3 bytes of a 4-byte frame
    script[:] = [b"abc", 0, -1]
    assert vu.read(-1) == 3
    assert vu.read(-1) == 0                                  # error hidden while bytes are buffered
    assert vu.result()[0] == cm.ERROR_INVAL
    h.unref(); vu.unref()


def test_product_fails_loudly_without_gpu(cm):
    if cm.device_count() > 0:
        pytest.skip("a GPU is present")
    logs = []
    cm.set_log_callback(lambda lvl, msg: logs.append((lvl, msg)))
    try:
        with pytest.raises(cm.CoolmicError):
            cm.Batch(1, 1, 64)
        assert "no CPU path" in cm.last_error()
        # a read that needs arithmetic cannot be served: -1 and an ERROR line, no CPU result
        tr = cm.Transform(48000, 1)
        tr.attach(cm.IoHandle.from_bytes(bytes(64)))
        tr.set_master_gain(1, 2, [1])
        h = tr.get_iohandle()
        assert h.read(64)[0] == -1
        vu = cm.Vumeter(48000, 1)
        vu.attach(cm.IoHandle.from_bytes(bytes(64)))
        assert vu.read(-1) == -1
        assert any(lvl == 1 and "no CPU path" in msg for lvl, msg in logs)
        assert all(msg.startswith("libcoolmic-dsp/") for _, msg in logs)
        # device placement behind the operator API: no device is a device the process sees
        assert tr.set_device(0) == cm.ERROR_INVAL and vu.set_device(0) == cm.ERROR_INVAL
        assert cm.lib.coolmic_transform_set_device(None, 0) == cm.ERROR_FAULT
        assert not cm.lib.coolmic_group_new_on(0, None, None, 48000, 2, 4, 64, 2)
        assert any("no HIP device 0 for the group" in msg for _, msg in logs)
        h.unref(); tr.unref(); vu.unref()
    finally:
        cm.set_log_callback(None)


def test_division_constants_are_exact(cm):
    """floor(|x| * gain / scale) == |x| * mi + ((|x| * mf) >> 32) for every magnitude 0..32768 -- the two
    instructions the kernels spend per sample (v_mul_hi_u32, v_mad_u32_u24; StreamParam in cmhip_internal.h),
    against the reference's 64-bit multiply and truncating divide (ref: src/transform.c:111-119).  Every class
    of (gain, scale): gain below / equal to / a multiple of / just beside a multiple of the scale, scale 1,
    powers of two and their neighbours, primes, the extremes 65535/1 and 1/65535, and random pairs.  The GPU
    test test_every_scale_divides_exactly_on_device repeats a subset on the device."""
    rng = np.random.default_rng(3)
    x = np.arange(0, 32769, dtype=np.uint64)
    scales = list(range(1, 260)) + [511, 512, 513, 999, 1000, 1001, 4095, 4096, 4097, 21845, 32767, 32768, 32769,
                                    43691, 65521, 65534, 65535] + [int(v) for v in rng.integers(1, 65536, 120)]
    checked = 0
    for scale in scales:
        gains = {1, 2, scale, 65535, 65534, 32768, 32767, max(1, scale - 1), min(65535, scale + 1),
                 max(1, scale // 2), max(1, scale // 3)}
        for k in (2, 3, 7, 64, 65535 // scale):
            for d in (-1, 0, 1):
                g = k * scale + d
                if 1 <= g <= 65535:
                    gains.add(g)
        gains.update(int(v) for v in rng.integers(1, 65536, 6))
        for gain in sorted(gains):
            mi, mf = cm.gain_consts(gain, scale)
            assert mi == gain // scale and mf < 2 ** 32
            assert (mf == 0) == (gain % scale == 0)
            q = x * np.uint64(mi) + ((x * np.uint64(mf)) >> np.uint64(32))
            assert np.array_equal(q, x * np.uint64(gain) // np.uint64(scale)), (gain, scale)
            checked += 1
    assert checked > 5000
    # the short form of the read-only runs is the mi == 0 case: nothing but the mulhi
    for gain, scale in ((900, 1000), (1, 65535), (65534, 65535), (1, 2), (32767, 32768)):
        mi, mf = cm.gain_consts(gain, scale)
        assert mi == 0
        assert np.array_equal((x * np.uint64(mf)) >> np.uint64(32), x * np.uint64(gain) // np.uint64(scale))


def test_logging_format(cm):
    logs = []
    cm.set_log_callback(lambda lvl, msg: logs.append((lvl, msg)))
    try:
        mono = cm.Transform(48000, 1)
        mono.set_master_gain(2, 7, [3, 4])                  # logs at DEBUG (ref: src/transform.c:217)
        mono.unref()
    finally:
        cm.set_log_callback(None)
    assert logs and logs[0][0] == 4
    assert "libcoolmic-dsp/transform in " in logs[0][1] and "DEBUG: gain: scale=7, gain[0]=3 (in: 3, 4)" in logs[0][1]
    assert cm.lib.coolmic_logging_level2string(1) == b"ERROR"
    assert cm.lib.coolmic_logging_level2string(99) == b"(unknown)"


def test_tee_fanout(cm):
    # ref: src/tee.c:83-289
    assert not cm.lib.coolmic_tee_new(None, None, 0)
    assert not cm.lib.coolmic_tee_new(None, None, 5)
    data = bytes(range(256)) * 40                      # 10240 bytes
    tee = cm.Tee(2)
    src = cm.IoHandle.from_bytes(data, chunk=700)
    assert tee.attach(src) == 0
    src.unref()
    a = tee.get_iohandle(-1)
    b = tee.get_iohandle(-1)
    assert not cm.lib.coolmic_tee_get_iohandle(tee.ptr, -1)       # only two readers
    assert not cm.lib.coolmic_tee_get_iohandle(tee.ptr, 2)
    got_a, got_b = b"", b""
    # reader a runs ahead until the shared buffer (<= 8192 bytes) is full, then starves
    while True:
        n, d = a.read(1024)
        got_a += d
        if n == 0:
            break
    assert 1024 <= len(got_a) <= 8192 and a.eof() == 0
    n, d = b.read(300)
    got_b += d
    assert n == 300
    n, d = a.read(1024)                                # room again after b moved on
    got_a += d
    assert n > 0
    for h, got in ((a, got_a), (b, got_b)):
        pass
    # drain both alternately
    for _ in range(200):
        n, d = b.read(1000); got_b += d
        n2, d2 = a.read(1000); got_a += d2
        if n == 0 and n2 == 0 and a.eof() == 1 and b.eof() == 1:
            break
    assert got_a == data and got_b == data
    a.unref(); b.unref(); tee.unref()


def test_group_needs_a_gpu(cm):
    if cm.device_count() > 0:
        pytest.skip("a GPU is present")
    assert not cm.lib.coolmic_group_new(None, None, 48000, 2, 4, 512, 2)
    assert not cm.lib.coolmic_group_new(None, None, 0, 2, 4, 512, 2)
    assert cm.lib.coolmic_group_pump(None) == cm.ERROR_FAULT


def test_stdio_source_replays_a_raw_pcm_file(cm, tmp_path):
    # ref: src/snddev_stdio.c:50-78 -- device is the file name, read() is fread()
    data = bytes(range(256)) * 9
    f = tmp_path / "capture.pcm"
    f.write_bytes(data)
    new = cm.lib.coolmic_snddev_new
    assert not new(None, None, b"stdio", None, 48000, 2, 1, -1)                       # no file name
    assert not new(None, None, b"stdio", str(tmp_path / "missing").encode(), 48000, 2, 1, -1)
    name = str(f).encode()
    d = new(None, None, b"stdio", name, 48000, 2, 1, -1)
    assert d
    h = cm.IoHandle(cm.lib.coolmic_snddev_get_iohandle(d))
    cm.lib.coolmic_ro_unref(d)
    assert h.read(1000) == (1000, data[:1000])
    assert h.read(5000) == (len(data) - 1000, data[1000:])     # short read at end of file
    assert h.read(10) == (0, b"")
    h.unref()
    assert cm.lib.coolmic_feature_check(b"driver:stdio") == 1


def test_snddev_playback_side(cm, tmp_path):
    """coolmic_snddev_attach_iohandle + coolmic_snddev_iter (ref: src/snddev.c:143-152, 171-215): one round
    flushes what the device has not taken, then moves up to 1 KiB from the attached handle to the device;
    "stdio" opened for TX writes its file, "null" and "sine" discard (ref: src/snddev_null.c, snddev_sine.c)."""
    data = bytes((i * 7 + 3) & 0xff for i in range(5000))
    out = tmp_path / "playback.pcm"
    dev = cm.Snddev("stdio", 48000, 2, flags=2, device=str(out))                     # TX: "wb"
    assert cm.lib.coolmic_snddev_iter(None) == cm.ERROR_FAULT
    assert dev.iter() == cm.ERROR_GENERIC                                            # nothing attached: the read fails
    src = cm.IoHandle.from_bytes(data, chunk=700)
    assert dev.attach(src) == 0 and src.refcount() == 2
    src.unref()
    for _ in range(5):                                                               # 5 x 1 KiB >= 5000 bytes
        assert dev.iter() == cm.ERROR_NONE
    assert dev.iter() == cm.ERROR_NONE                                               # source exhausted: nothing to do
    assert dev.attach(None) == 0
    h = dev.get_iohandle()
    assert h.read(16) == (0, b"")                                                    # a file opened for writing reads nothing
    h.unref()
    dev.unref()                                                                      # closes the file
    assert out.read_bytes() == data
    for driver in ("null", "sine"):
        d = cm.Snddev(driver, 48000, 1, flags=3)
        s2 = cm.IoHandle.from_bytes(data[:1500])
        assert d.attach(s2) == 0
        s2.unref()
        assert d.iter() == cm.ERROR_NONE and d.iter() == cm.ERROR_NONE and d.iter() == cm.ERROR_NONE
        d.unref()
    # both directions on one file (ref: "w+b")
    both = cm.Snddev("stdio", 48000, 1, flags=3, device=str(tmp_path / "both.pcm"))
    s3 = cm.IoHandle.from_bytes(b"abcdefgh")
    assert both.attach(s3) == 0 and both.iter() == cm.ERROR_NONE
    s3.unref()
    both.unref()
    assert (tmp_path / "both.pcm").read_bytes() == b"abcdefgh"


def test_vu_colour_helpers_match_the_restatement(cm, oracle):
    """SURVEY 8f-4 (ref: src/util.c).  Product and oracle restatement compared over a grid, plus the values the
    source text fixes; tests/test_ref_util.py holds both against a build of the reference's own util.c."""
    import math
    lib, o = cm.lib, oracle.lib
    for p in [-200.0, -20.0001, -20.0, -19.9, -10.0, -3.0111266389980154, -1e-9, 0.0, 0.5, -math.inf]:
        assert lib.coolmic_util_power2hue(p, b"default") == o.oracle_power2hue(p)
    assert lib.coolmic_util_power2hue(-30.0, b"default") == math.pi * 2 / 3
    assert lib.coolmic_util_power2hue(-10.0, b"other") == 0.0
    for pk in [-32768, -32767, -30001, -30000, -28001, -28000, 0, 27999, 28000, 28001, 30000, 30001, 32766, 32767]:
        assert lib.coolmic_util_peak2hue(pk, b"default") == o.oracle_peak2hue(pk)
    assert lib.coolmic_util_peak2hue(32767, b"default") == 0.0
    assert lib.coolmic_util_peak2hue(30001, b"default") == 0.43
    assert lib.coolmic_util_peak2hue(100, b"default") == math.pi * 2 / 3
    for a in (0.0, 0.5, 1.0, 2.0):
        for h in np.linspace(-0.5, 7.0, 61):
            for sat in (0.0, 0.3, 1.0):
                for v in (0.0, 0.7, 1.0, 1.5):
                    assert lib.coolmic_util_ahsv2argb(a, h, sat, v) == o.oracle_ahsv2argb(a, h, sat, v)
    assert lib.coolmic_util_ahsv2argb(1.0, 0.0, 1.0, 1.0) == 0xFFFF0000          # red
    assert lib.coolmic_util_ahsv2argb(1.0, math.pi * 2 / 3, 1.0, 1.0) & 0xFF00FF00 == 0xFF00FF00   # green channel full
    # the batch form
    res = (cm.VuResult * 3)()
    res[0].global_power, res[0].global_peak = -30.0, 100
    res[1].global_power, res[1].global_peak = -3.0, 30500
    res[2].global_power, res[2].global_peak = 0.0, -32768
    pw = (C.c_uint32 * 3)()
    pk = (C.c_uint32 * 3)()
    lib.coolmic_util_vu_argb(res, 3, b"default", pw, pk)
    for i in range(3):
        assert pw[i] == o.oracle_ahsv2argb(1.0, o.oracle_power2hue(res[i].global_power), 1.0, 1.0)
        assert pk[i] == o.oracle_ahsv2argb(1.0, o.oracle_peak2hue(res[i].global_peak), 1.0, 1.0)
    assert pk[2] == 0xFFFF0000


def _raw_window(x, C):
    """one launch's raw VU window of the interleaved block x, as the device keeps it (cmhip_internal.h:
    key = |peak| << 47 | (~index & (2^46 - 1)) << 1 | negative, index in interleaved samples of the block)"""
    w = np.zeros(33, dtype=np.uint64)
    v = x.astype(np.int64)
    for c in range(C):
        col = v[c::C]
        w[c] = np.uint64(int((col * col).sum()))
        mag = np.abs(col)
        if mag.size and mag.max() > 0:
            f = int(np.argmax(mag))                      # the first of the largest
            idx = f * C + c
            w[16 + c] = np.uint64((int(mag[f]) << 47) | ((~idx & ((1 << 46) - 1)) << 1) | (1 if col[f] < 0 else 0))
    w[32] = np.uint64(x.size)
    return w


@pytest.mark.parametrize("C", [1, 2, 6])
def test_window_records_merge_like_one_window(cm, oracle, C):
    """The host arithmetic a meter behind a tee relies on (csrc/vumeter.c, cmhip_vu_raw_merge / _finish): the
    windows of consecutive launches, merged in stream order, give the reference's result over the whole stretch
    (ref: src/vumeter.c:161-177, 189-218) -- sums add, an equal peak in a later launch does not replace an
    earlier one, the global peak is the earliest of the largest across channels.  No GPU involved."""
    from oracle import oracle_ffi as of
    rng = np.random.default_rng(77 + C)
    for trial in range(40):
        nblk = int(rng.integers(1, 9))
        kind = trial % 4
        blocks = []
        for _ in range(nblk):
            n = int(rng.integers(0, 700)) * C
            if kind == 0:
                x = rng.integers(-32768, 32768, size=n, dtype=np.int64).astype(np.int16)
            elif kind == 1:                              # ties everywhere, both signs
                x = rng.choice(np.array([-32768, -32767, 0, 32767], dtype=np.int16), size=n)
            elif kind == 2:                              # small values: ties across launches and channels
                x = rng.integers(-3, 4, size=n, dtype=np.int64).astype(np.int16)
            else:
                x = np.zeros(n, dtype=np.int16)          # silence (maybe with one sample somewhere)
                if n and rng.random() < 0.5:
                    x[int(rng.integers(0, n))] = int(rng.choice([-5, 5]))
            blocks.append(x)
        v = oracle.vu_new(C)
        for x in blocks:
            if x.size:
                oracle.vu_accumulate(v, x)
        rc_o, r_o = oracle.vu_result(v)
        rc, r = cm.merge_windows(np.stack([_raw_window(x, C) for x in blocks]), C)
        assert rc == rc_o, (C, trial)
        if rc_o == 0:
            assert r.as_dict() == of.vu_result_dict(r_o), (C, trial)
