"""A second, independent witness for the oracle (test infrastructure checking test
infrastructure): the integer part of the path restated once more in pure Python -- written
from the reference's text (ref: src/transform.c:101-124 `__process`, src/transform.c:195-222 the
gain rules, src/vumeter.c:161-177 the accumulate loop, src/vumeter.c:201-212 the dB finish), NOT
from oracle/oracle.c -- and run against liboracle.so on randomised blocks, so that a slip in
the C restatement cannot hide behind agreeing with itself.  Python integers are unbounded and
`math.sqrt` / `math.log10` are the same libm doubles, so this witness has no overflow or
truncation rules of its own: C's truncating division is written out."""
import math

import numpy as np
import pytest

from oracle import oracle_ffi as of


def c_div(a, b):
    """C's `/` on signed integers: truncation toward zero"""
    q = abs(a) // abs(b)
    return q if (a < 0) == (b < 0) else -q


def witness_process(samples, channels, scale, gain):
    """ref: src/transform.c:101-124"""
    if not scale:
        return list(samples)
    out = []
    for i, x in enumerate(samples):
        tmp = c_div(x * gain[i % channels], scale)
        if tmp >= 32767:
            tmp = 32767
        elif tmp <= -32768:
            tmp = -32768
        out.append(tmp)
    return out


def witness_gain_rules(stream_channels, channels, scale, gain):
    """ref: src/transform.c:195-222 -> (rc, scale, gain list) for a stream of stream_channels"""
    if not channels or not scale or gain is None:
        return 0, 0, None
    if channels == stream_channels:
        return 0, scale, list(gain[:channels])
    if channels == 1:
        return 0, scale, [gain[0]] * stream_channels
    if channels == 2 and stream_channels == 1:
        return 0, scale, [(gain[0] + gain[1]) // 2]
    return -10, None, None


class WitnessVu:
    """ref: src/vumeter.c:161-177 (accumulate), :189-218 (result)"""

    def __init__(self, channels):
        self.channels = channels
        self.clear()

    def clear(self):
        self.channel_peak = [0] * self.channels
        self.global_peak = 0
        self.power = [0] * self.channels
        self.frames = 0

    def accumulate(self, samples):
        c = 0
        for x in samples:
            if abs(x) > abs(self.channel_peak[c]):
                self.channel_peak[c] = x
                if abs(x) > abs(self.global_peak):
                    self.global_peak = x
            self.power[c] += x * x
            c += 1
            if c == self.channels:
                c = 0
                self.frames += 1

    @staticmethod
    def _db(p_int):
        if p_int == 0:
            return -math.inf
        return min(20.0 * math.log10(math.sqrt(float(p_int)) / 32768.0), 0.0)

    def result(self):
        if not self.frames:
            return -10, None
        r = {"frames": self.frames, "global_peak": self.global_peak,
             "channel_peak": list(self.channel_peak),
             "channel_power": [self._db(c_div(p, self.frames)) for p in self.power],
             "global_power": self._db(sum(self.power) // (self.frames * self.channels))}
        self.clear()
        return 0, r


EDGE = [-32768, -32767, -1, 0, 1, 32766, 32767]


def _block(rng, n):
    kind = rng.integers(0, 4)
    if kind == 0:
        x = rng.integers(-32768, 32768, n)
    elif kind == 1:
        x = rng.choice(EDGE, n)                                  # ties and the asymmetric ends
    elif kind == 2:
        x = (rng.normal(0, 3000, n)).astype(np.int64).clip(-32768, 32767)
    else:
        x = rng.integers(-3, 4, n)                               # many equal magnitudes: first-max rule
    return np.asarray(x, dtype=np.int16)


@pytest.mark.parametrize("seed", range(12))
def test_oracle_agrees_with_the_python_witness(oracle, seed):
    rng = np.random.default_rng(7000 + seed)
    C_ = int(rng.choice([1, 2, 3, 6, 16]))
    scale = int(rng.choice([0, 1, 2, 3, 7, 1000, 32768, 65535, int(rng.integers(1, 65536))]))
    gshape = int(rng.choice([C_, 1, 2]))
    gains = [int(v) for v in rng.choice([0, 1, 2, 999, 1000, 1001, 65535, int(rng.integers(0, 65536))], gshape)]
    rc_w, w_scale, w_gain = witness_gain_rules(C_, gshape, scale, gains)
    rc_o, g = oracle.gain(C_, gshape, scale, gains)
    assert rc_o == rc_w
    if rc_w != 0:
        return
    vu_w = WitnessVu(C_)
    vu_o = oracle.vu_new(C_)
    for _ in range(3):
        frames = int(rng.integers(1, 700))
        x = _block(rng, frames * C_)
        want = witness_process([int(v) for v in x], C_, w_scale, w_gain)
        got = oracle.gain_apply(g, x, C_)
        assert got.tolist() == want
        vu_w.accumulate(want)
        oracle.vu_accumulate(vu_o, got)
    rc1, r_w = vu_w.result()
    rc2, r_o = oracle.vu_result(vu_o)
    assert rc1 == rc2 == 0
    d = of.vu_result_dict(r_o)
    assert d["frames"] == r_w["frames"]
    assert d["global_peak"] == r_w["global_peak"]
    assert d["channel_peak"] == r_w["channel_peak"]
    # dB doubles: equal to the last bit (same libm, same integer mean first)
    assert d["global_power"] == r_w["global_power"]
    assert d["channel_power"] == r_w["channel_power"]


def test_witness_reproduces_the_hand_checkable_vectors(golden):
    """K1-K5, K9 of SURVEY 8(c) by the witness alone (short enough to check by hand)"""
    for name in ("K1", "K2", "K3", "K4", "K5", "K9"):
        case = golden["cases"][name]
        ch = case["channels"]
        gs = case["gain"]
        rc, sc, gn = witness_gain_rules(ch, gs["channels"], gs["scale"], gs["gain"])
        assert rc == 0
        out = witness_process(golden[case["input"]], ch, sc, gn)
        assert out == case["pcm"], name
        v = WitnessVu(ch)
        v.accumulate(out)
        _, r = v.result()
        for key, want in case["vu"].items():
            assert r[key] == want, (name, key)


# ---- the VU colour helpers, written from ref: src/util.c:30-138 ---------------------

def _x255(x):
    x = 1.0 if x >= 1.0 else (0.0 if x <= 0.0 else x)
    return min(int(x * 255.0), 255)


def witness_ahsv2argb(alpha, hue, sat, value):
    hue1 = int(hue / (math.pi / 3.0))                     # C's (int): truncation toward zero
    f = hue - float(hue1)
    p_ = value * (1.0 - sat)
    q_ = value * (1.0 - sat * f)
    t_ = value * (1.0 - sat * (1.0 - f))
    rgb = {0: (value, t_, p_), 6: (value, t_, p_), 1: (q_, value, p_), 2: (p_, value, t_),
           3: (p_, q_, value), 4: (t_, p_, value), 5: (value, p_, q_)}.get(hue1, (0.0, 0.0, 0.0))
    return (_x255(alpha) << 24) + (_x255(rgb[0]) << 16) + (_x255(rgb[1]) << 8) + _x255(rgb[2])


def witness_power2hue(power):
    if power < -20.0:
        return math.pi * 2.0 / 3.0
    if power >= 0:
        return 0.0
    return math.pow(math.sin(math.pi * power / 40.0), 2.0) * math.pi * 2.0 / 3.0


def witness_peak2hue(peak):
    if peak in (-32768, 32767):
        return 0.0
    if peak < -30000 or peak > 30000:
        return 0.43
    if peak < -28000 or peak > 28000:
        return 1.0
    return math.pi * 2.0 / 3.0


def test_colour_helpers_against_the_python_witness(oracle, cm):
    """product (csrc/util.c) and oracle restatement against a third, independent one"""
    rng = np.random.default_rng(99)
    for _ in range(400):
        a, h, sat, v = rng.uniform(-0.2, 1.3), rng.uniform(0.0, 2.0 * math.pi + 0.2), rng.uniform(0, 1), rng.uniform(-0.1, 1.4)
        want = witness_ahsv2argb(a, h, sat, v)
        assert oracle.lib.oracle_ahsv2argb(a, h, sat, v) == want
        assert cm.lib.coolmic_util_ahsv2argb(a, h, sat, v) == want
    for p in list(rng.uniform(-60.0, 3.0, 200)) + [-20.0, -20.000001, 0.0, -1e-12, -math.inf]:
        want = witness_power2hue(float(p))
        assert oracle.lib.oracle_power2hue(float(p)) == want
        assert cm.lib.coolmic_util_power2hue(float(p), b"default") == want
    for pk in list(rng.integers(-32768, 32768, 300)) + [-32768, 32767, -30001, -30000, 30000, 30001, -28001, 28001, 28000]:
        want = witness_peak2hue(int(pk))
        assert oracle.lib.oracle_peak2hue(int(pk)) == want
        assert cm.lib.coolmic_util_peak2hue(int(pk), b"default") == want
