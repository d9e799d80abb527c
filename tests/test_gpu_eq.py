"""GPU parity for the EQ path (BASELINE config 3): int16 -> gain -> x/32768.f -> 3 biquads
-> float / int16 (+VU).  PARITY UNPINNED against the reference (it has no filter code);
the oracle's Direct-Form-I fmaf order is the specification and the bar is bit-equal
floats (both sides are built with -ffp-contract=off and use correctly rounded fma)."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle_ffi as of

pytestmark = pytest.mark.gpu


def _oracle_eq(oracle, coef, nsec, gain_args, blocks):
    q = (of.Biquad * max(nsec, 1))()
    for i in range(nsec):
        q[i].b0, q[i].b1, q[i].b2, q[i].a1, q[i].a2 = [float(v) for v in coef[5 * i:5 * i + 5]]
    g = None
    if gain_args is not None:
        rc, g = oracle.gain(1, *gain_args)
        assert rc == 0
    st = np.zeros(4 * max(nsec, 1), dtype=np.float32)
    outs = []
    for blk in blocks:
        outs.append(oracle.eq_run_mono(g, q, nsec, st, blk))
    return outs


def test_design_matches_oracle_design(gpu, oracle):
    cm = gpu
    q = oracle.eq3(48000.0)
    want = np.array([[q[i].b0, q[i].b1, q[i].b2, q[i].a1, q[i].a2] for i in range(3)],
                    dtype=np.float32).reshape(-1)
    assert np.array_equal(cm.eq3(48000.0).view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("nsec", [0, 1, 3, 4])
def test_eq_bit_exact_with_state_carry(gpu, oracle, nsec):
    cm = gpu
    rng = np.random.default_rng(7 + nsec)
    S, T = 70, 700                       # more than one 64-stream tile, ragged tiles
    coef3 = cm.eq3(48000.0)
    coef = np.concatenate([coef3, cm.design_biquad(1, 48000.0, 3000.0, 4.0, 2.0)])[: 5 * nsec]
    b = cm.Batch(S, 1, T, flags=cm.EQ | cm.OUT_F32 | cm.OUT_PCM | cm.VU)
    assert b.set_eq(-1, coef) == 0
    gains = []
    for s in range(S):
        ga = None if s % 3 == 0 else (1, 1000, [int(rng.integers(200, 2500))])
        if ga:
            assert b.set_gain(s, *ga) == 0
        gains.append(ga)
    lens1 = [int(v) for v in rng.integers(0, T + 1, S)]
    lens1[0], lens1[1], lens1[2] = T, 0, 1
    lens2 = [int(v) for v in rng.integers(0, T + 1, S)]
    blocks = [[rng.integers(-32768, 32768, n).astype(np.int16) for n in (lens1[s], lens2[s])]
              for s in range(S)]
    got = [[None, None] for _ in range(S)]
    for k, lens in enumerate((lens1, lens2)):
        for s in range(S):
            if lens[s]:
                b.upload(s, blocks[s][k])
        b.run(T, frames_per_stream=lens)
        for s in range(S):
            got[s][k] = (b.download_f32(s, 0, lens[s]), b.download(s, lens[s]))
    for s in range(S):
        want = _oracle_eq(oracle, coef, nsec, gains[s], blocks[s])
        v = oracle.vu_new(1)
        for k in range(2):
            wf, wi = want[k]
            gf, gi = got[s][k]
            assert np.array_equal(gf.view(np.uint32), wf.view(np.uint32)), (nsec, s, k)
            assert np.array_equal(gi, wi), (nsec, s, k)
            oracle.vu_accumulate(v, wi)
        rc_o, r_o = oracle.vu_result(v)
        rc_g, r_g = b.vu_result(s)
        assert rc_g == rc_o, (nsec, s)
        if rc_o == 0:
            assert r_g.as_dict() == of.vu_result_dict(r_o), (nsec, s)
    b.close()


def test_eq_config3_shape_sampled(gpu, oracle):
    """BASELINE config 3 shape (8192 mono streams) at a block the oracle finishes quickly;
    sampled streams compared bit for bit, float output only"""
    cm = gpu
    S, T = 8192, 2048
    coef = cm.eq3(48000.0)
    b = cm.Batch(S, 1, T, flags=cm.EQ | cm.OUT_F32)
    assert b.set_eq(-1, coef) == 0
    assert b.set_gain(-1, 1, 1000, [900]) == 0
    b.generate(cm.GEN_NOISE, 12345, T)
    b.run(T)
    for s in (0, 63, 64, 4097, 8191):
        x = oracle.lcg(12345 + s, T)
        (wf, _), = _oracle_eq(oracle, coef, 3, (1, 1000, [900]), [x])
        gf = b.download_f32(s, 0, T)
        assert np.array_equal(gf.view(np.uint32), wf.view(np.uint32)), s
    b.close()


def test_eq_config3_full_size_properties(gpu, oracle):
    """BASELINE config 3 at full size (8192 mono streams x 65 536 frames, 3-band EQ, float planes):
    sampled streams bit-equal to the oracle, and the same signal fed as two half blocks (filter
    state carried across the launches) bit-equal to the one-block run on those streams."""
    cm = gpu
    S, T = 8192, 65536
    coef = cm.eq3(48000.0)
    b = cm.Batch(S, 1, T, flags=cm.EQ | cm.OUT_F32)
    assert b.set_eq(-1, coef) == 0
    assert b.set_gain(-1, 1, 1000, [900]) == 0
    b.generate(cm.GEN_NOISE, 12345, T)
    b.run(T)
    pick = (0, 31, 32, 4095, 4096, 8191)
    full = {}
    for s in pick:
        (wf, _), = _oracle_eq(oracle, coef, 3, (1, 1000, [900]), [oracle.lcg(12345 + s, T)])
        full[s] = b.download_f32(s, 0, T).copy()
        assert np.array_equal(full[s].view(np.uint32), wf.view(np.uint32)), s
    b.close()
    h = cm.Batch(S, 1, T // 2, flags=cm.EQ | cm.OUT_F32)
    assert h.set_eq(-1, coef) == 0
    assert h.set_gain(-1, 1, 1000, [900]) == 0
    for half in range(2):
        h.generate(cm.GEN_NOISE, 12345, T // 2, frame_offset=half * (T // 2))
        h.run(T // 2)
        for s in pick:
            got = h.download_f32(s, 0, T // 2)
            want = full[s][half * (T // 2):(half + 1) * (T // 2)]
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (s, half)
    h.close()


@pytest.mark.parametrize("nsec", [1, 2, 3, 4])
def test_eq_pipelined_kernel_ragged_and_state_carry(gpu, oracle, nsec):
    """float-only batches take the pipelined kernel (k_eq_pipe): ragged stream ends inside
    and across 64-frame blocks, streams of length 0, three launches with carried state"""
    cm = gpu
    rng = np.random.default_rng(100 + nsec)
    S, T = 130, 333                      # 3 workgroups (64 + 64 + 2 streams), 6 blocks
    coef = np.concatenate([cm.eq3(48000.0), cm.design_biquad(1, 48000.0, 3000.0, 4.0, 2.0)])[: 5 * nsec]
    b = cm.Batch(S, 1, T, flags=cm.EQ | cm.OUT_F32)
    assert b.set_eq(-1, coef) == 0
    gains = []
    for s in range(S):
        ga = None if s % 4 == 0 else (1, 1000, [int(rng.integers(200, 2500))])
        if ga:
            assert b.set_gain(s, *ga) == 0
        gains.append(ga)
    all_lens = []
    for k in range(3):
        lens = [int(v) for v in rng.integers(0, T + 1, S)]
        lens[0], lens[1], lens[2], lens[3], lens[64], lens[129] = T, 0, 1, 64, 65, T
        if k == 1:
            lens = [T] * S               # a launch where no stream is ragged
        all_lens.append(lens)
    blocks = [[rng.integers(-32768, 32768, all_lens[k][s]).astype(np.int16) for k in range(3)]
              for s in range(S)]
    got = [[None] * 3 for _ in range(S)]
    for k in range(3):
        for s in range(S):
            if all_lens[k][s]:
                b.upload(s, blocks[s][k])
        b.run(T, frames_per_stream=all_lens[k])
        for s in range(S):
            got[s][k] = b.download_f32(s, 0, all_lens[k][s])
    for s in range(S):
        want = _oracle_eq(oracle, coef, nsec, gains[s], blocks[s])
        for k in range(3):
            wf, _ = want[k]
            assert np.array_equal(got[s][k].view(np.uint32), wf.view(np.uint32)), (nsec, s, k)
    b.close()


@pytest.mark.parametrize("gain_mode,inplace", [("off", False), ("unity", False), ("mixed", True)])
def test_eq_pipelined_int16_and_vu_only(gpu, oracle, gain_mode, inplace):
    """the chain a VU meter or encoder sits behind: EQ result as int16 (+VU window), no float
    planes.  gain off / gain == scale take the conversion short cut of the pipelined kernel;
    the in-place case overwrites the input PCM with the result"""
    cm = gpu
    rng = np.random.default_rng(4242 + len(gain_mode))
    S, T = 75, 400
    coef = cm.eq3(48000.0)
    flags = cm.EQ | cm.OUT_PCM | cm.VU | (cm.INPLACE if inplace else 0)
    b = cm.Batch(S, 1, T, flags=flags)
    assert b.set_eq(-1, coef) == 0
    gains = []
    for s in range(S):
        if gain_mode == "off":
            ga = None
        elif gain_mode == "unity":
            ga = (1, 777, [777])
        else:
            ga = None if s % 5 == 0 else (1, 1000, [int(rng.integers(100, 3000))])
        if ga:
            assert b.set_gain(s, *ga) == 0
        gains.append(ga)
    all_lens = []
    for k in range(2):
        lens = [int(v) for v in rng.integers(0, T + 1, S)]
        lens[0], lens[1], lens[2], lens[3] = T, 1 - k, 2, 64
        all_lens.append(lens)
    blocks = [[rng.integers(-32768, 32768, all_lens[k][s]).astype(np.int16) for k in range(2)]
              for s in range(S)]
    got = [[None] * 2 for _ in range(S)]
    for k in range(2):
        for s in range(S):
            if all_lens[k][s]:
                b.upload(s, blocks[s][k])
        b.run(T, frames_per_stream=all_lens[k])
        for s in range(S):
            got[s][k] = b.download(s, all_lens[k][s])
    for s in range(S):
        want = _oracle_eq(oracle, coef, 3, gains[s], blocks[s])
        v = oracle.vu_new(1)
        for k in range(2):
            assert np.array_equal(got[s][k], want[k][1]), (gain_mode, s, k)
            oracle.vu_accumulate(v, want[k][1])
        rc_o, r_o = oracle.vu_result(v)
        rc_g, r_g = b.vu_result(s)
        assert rc_g == rc_o, (gain_mode, s)
        if rc_o == 0:
            assert r_g.as_dict() == of.vu_result_dict(r_o), (gain_mode, s)
    b.close()


@pytest.mark.parametrize("C", [2, 3, 6, 16])
def test_eq_on_multichannel_streams(gpu, oracle, C):
    """every channel of a stream runs the stream's filter with state of its own, after the
    channel map and its own gain: float planes, interleaved int16 result, per-channel VU;
    expected values are the mono oracle run once per channel"""
    cm = gpu
    rng = np.random.default_rng(31 + C)
    S, T, nsec = 37, 300, 3
    coef = cm.eq3(48000.0)
    for flags in (cm.EQ | cm.OUT_F32 | cm.OUT_PCM | cm.VU, cm.EQ | cm.OUT_PCM | cm.VU | cm.INPLACE,
                  cm.EQ | cm.OUT_F32):
        b = cm.Batch(S, C, T, flags=flags)
        assert b.set_eq(-1, coef) == 0
        gains, maps = [], []
        for s in range(S):
            g = None if s % 4 == 0 else (int(rng.integers(1, 3000)), [int(v) for v in rng.integers(0, 4000, C)])
            m = None if s % 3 == 0 else [int(v) for v in rng.integers(0, C, C)]
            if g:
                assert b.set_gain(s, C, g[0], g[1]) == 0
            if m:
                assert b.set_chmap(s, m) == 0
            gains.append(g)
            maps.append(m)
        all_lens = []
        for k in range(2):
            lens = [int(v) for v in rng.integers(0, T + 1, S)]
            lens[0], lens[1], lens[2], lens[3] = T, k, 2, 64
            all_lens.append(lens)
        blocks = [[rng.integers(-32768, 32768, all_lens[k][s] * C).astype(np.int16) for k in range(2)]
                  for s in range(S)]
        got = [[None] * 2 for _ in range(S)]
        for k in range(2):
            for s in range(S):
                if all_lens[k][s]:
                    b.upload(s, blocks[s][k])
            b.run(T, frames_per_stream=all_lens[k])
            for s in range(S):
                n = all_lens[k][s]
                pcm = b.download(s, n) if (flags & cm.OUT_PCM) else None
                planes = [b.download_f32(s, c, n) for c in range(C)] if (flags & cm.OUT_F32) else None
                got[s][k] = (pcm, planes)
        for s in range(S):
            per_ch = []
            for c in range(C):
                src = c if maps[s] is None else maps[s][c]
                chans = [blocks[s][k].reshape(-1, C)[:, src].copy() for k in range(2)]
                ga = None if gains[s] is None else (1, gains[s][0], [gains[s][1][c]])
                per_ch.append(_oracle_eq(oracle, coef, nsec, ga, chans))
            v = oracle.vu_new(C)
            for k in range(2):
                n = all_lens[k][s]
                want_i = np.stack([per_ch[c][k][1] for c in range(C)], axis=1).reshape(-1) if n else np.zeros(0, np.int16)
                pcm, planes = got[s][k]
                if pcm is not None:
                    assert np.array_equal(pcm, want_i), (C, flags, s, k)
                if planes is not None:
                    for c in range(C):
                        assert np.array_equal(planes[c].view(np.uint32), per_ch[c][k][0].view(np.uint32)), (C, s, k, c)
                oracle.vu_accumulate(v, want_i)
            if flags & cm.VU:
                rc_o, r_o = oracle.vu_result(v)
                rc_g, r_g = b.vu_result(s)
                assert rc_g == rc_o, (C, s)
                if rc_o == 0:
                    assert r_g.as_dict() == of.vu_result_dict(r_o), (C, flags, s)
        b.close()


def test_eq_non_finite_results_saturate_like_the_oracle(gpu, oracle):
    """a filter that overflows: +-inf and NaN in the float result; the int16 result saturates
    and takes 0 for NaN exactly as oracle_f32_to_i16 says (the S waves use the hardware's
    saturating conversions for this), and the VU window follows the int16 result"""
    cm = gpu
    rng = np.random.default_rng(99)
    S, T = 40, 500
    coef = np.array([3.0e38, -3.0e38, 3.0e38, 0.25, 0.5,        # overflows within a few samples
                     1.0, 0.0, 0.0, 0.0, 0.0], dtype=np.float32)
    b = cm.Batch(S, 1, T, flags=cm.EQ | cm.OUT_F32 | cm.OUT_PCM | cm.VU)
    assert b.set_eq(-1, coef) == 0
    xs = [rng.integers(-32768, 32768, T).astype(np.int16) for _ in range(S)]
    for s in range(S):
        b.upload(s, xs[s])
    b.run(T)
    saw_nan = saw_inf = False
    for s in range(S):
        (wf, wi), = _oracle_eq(oracle, coef, 2, None, [xs[s]])
        gf, gi = b.download_f32(s, 0, T), b.download(s, T)
        saw_nan |= bool(np.isnan(wf).any())
        saw_inf |= bool(np.isinf(wf).any())
        assert np.array_equal(np.isnan(gf), np.isnan(wf)), s
        fin = ~np.isnan(wf)
        assert np.array_equal(gf[fin].view(np.uint32), wf[fin].view(np.uint32)), s
        assert np.array_equal(gi, wi), s
        v = oracle.vu_new(1)
        oracle.vu_accumulate(v, wi)
        rc_o, r_o = oracle.vu_result(v)
        rc_g, r_g = b.vu_result(s)
        assert rc_g == rc_o == 0 and r_g.as_dict() == of.vu_result_dict(r_o), s
    assert saw_nan and saw_inf
    b.close()


def test_eq_on_a_second_device_of_the_process(gpu, oracle):
    """The EQ kernels need their dynamic-LDS limit raised per function AND per device (prepare_eq,
    csrc/k_eq.hip): a host that drives several GPUs from one process creates EQ batches on each.
    Needs two GPUs; the 1-GPU test box skips it."""
    cm = gpu
    if cm.device_count() < 2:
        pytest.skip("one GPU visible")
    S, T = 40, 700
    coef = cm.eq3(48000.0)
    rng = np.random.default_rng(12)
    xs = [rng.integers(-32768, 32768, T).astype(np.int16) for _ in range(S)]
    outs = []
    for dev in (0, 1, 0):
        b = cm.Batch(S, 1, T, flags=cm.EQ | cm.OUT_F32, device=dev)
        assert b.set_eq(-1, coef) == 0
        for s in range(S):
            b.upload(s, xs[s])
        b.run(T)
        outs.append([b.download_f32(s, 0, T) for s in range(S)])
        b.close()
    for s in range(S):
        (wf, _), = _oracle_eq(oracle, coef, 3, None, [xs[s]])
        for k in range(3):
            assert np.array_equal(outs[k][s].view(np.uint32), wf.view(np.uint32)), (k, s)
