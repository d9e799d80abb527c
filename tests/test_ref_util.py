"""SURVEY 8f-4, the VU colour helpers (ref: src/util.c:59-138) -- the one file at the path's edge that the image
can compile from the reference's own source (it needs libm and the reference's <coolmic-dsp/util.h> only):
`make -C oracle _ref` builds oracle/_ref/libref_util.so from it, unmodified.  Here the oracle's restatement AND the
product's helpers are held against that build over dense grids, and against tests/golden/ref_util.json, the vectors
tests/golden/make_ref_util.py took from it (for machines where neither the reference nor its build is present).
All comparisons are bit-exact (doubles by their bits)."""
import ctypes as C
import json
import math
import os
import struct

import numpy as np
import pytest

from oracle import oracle_ffi as of

HERE = os.path.dirname(os.path.abspath(__file__))


def _bits(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


@pytest.fixture(scope="module")
def ref():
    r = of.load_ref_util()
    if r is None:
        pytest.skip("oracle/_ref/libref_util.so not built (no reference sources here)")
    return r


@pytest.fixture(scope="module")
def vectors():
    with open(os.path.join(HERE, "golden", "ref_util.json")) as f:
        return json.load(f)


def _sides(cm, oracle):
    lib, o = cm.lib, oracle.lib
    return (("oracle", lambda p: o.oracle_peak2hue(p), lambda p: o.oracle_power2hue(p),
             lambda a, h, s, v: o.oracle_ahsv2argb(a, h, s, v)),
            ("product", lambda p: lib.coolmic_util_peak2hue(p, b"default"),
             lambda p: lib.coolmic_util_power2hue(p, b"default"),
             lambda a, h, s, v: lib.coolmic_util_ahsv2argb(a, h, s, v)))


def test_colour_helpers_against_the_reference_build(cm, oracle, ref):
    """every peak value; powers across and around the profile's break points; the colour wheel on a grid that
    touches every sector border from both sides"""
    rng = np.random.default_rng(8)
    powers = [-math.inf, -200.0, -20.000000000000004, -20.0, -19.999999999999996, -5e-324, 0.0, 5e-324, 3.0]
    powers += [float(x) for x in np.linspace(-45.0, 5.0, 2001)] + [float(x) for x in rng.uniform(-21.0, 0.5, 3000)]
    hues = [-0.5, -1e-9, 0.0] + [i * math.pi / 3 for i in range(8)]
    hues += [math.nextafter(i * math.pi / 3, 0.0) for i in range(1, 8)] + [math.nextafter(i * math.pi / 3, 9.0) for i in range(7)]
    hues += [float(x) for x in rng.uniform(0.0, 2 * math.pi, 300)]
    for name, peak2hue, power2hue, ahsv2argb in _sides(cm, oracle):
        for p in range(-32768, 32768):
            assert _bits(peak2hue(p)) == _bits(ref.coolmic_util_peak2hue(p, b"default")), (name, p)
        for p in powers:
            assert _bits(power2hue(p)) == _bits(ref.coolmic_util_power2hue(p, b"default")), (name, p)
        for h in hues:
            for a in (0.0, 0.25, 1.0, 1.5):
                for s in (0.0, 0.3, 0.999, 1.0):
                    for v in (0.0, 0.004, 0.7, 1.0, 1.5):
                        assert ahsv2argb(a, h, s, v) == ref.coolmic_util_ahsv2argb(a, h, s, v), (name, a, h, s, v)
    # the profile argument: anything but "default" is red in the reference; the product's helpers take it too
    for prof in (b"other", b""):
        for p in (-32768, 0, 29000):
            assert _bits(cm.lib.coolmic_util_peak2hue(p, prof)) == _bits(ref.coolmic_util_peak2hue(p, prof))
        for p in (-30.0, -10.0, 0.0):
            assert _bits(cm.lib.coolmic_util_power2hue(p, prof)) == _bits(ref.coolmic_util_power2hue(p, prof))


def test_colour_helpers_against_the_committed_reference_vectors(cm, oracle, vectors):
    for name, peak2hue, power2hue, ahsv2argb in _sides(cm, oracle):
        for p, want in vectors["peak2hue"]:
            assert _bits(peak2hue(p)) == _bits(float.fromhex(want)), (name, p)
        for p, want in vectors["power2hue"]:
            assert _bits(power2hue(float.fromhex(p))) == _bits(float.fromhex(want)), (name, p)
        for a, h, s, v, want in vectors["ahsv2argb"]:
            assert ahsv2argb(*(float.fromhex(x) for x in (a, h, s, v))) == want, (name, a, h, s, v)
    for p, want in vectors["peak2hue_other_profile"]:
        assert _bits(cm.lib.coolmic_util_peak2hue(p, b"other")) == _bits(float.fromhex(want))
    for p, want in vectors["power2hue_other_profile"]:
        assert _bits(cm.lib.coolmic_util_power2hue(float.fromhex(p), b"other")) == _bits(float.fromhex(want))


def test_the_committed_vectors_are_what_the_reference_build_gives(ref, vectors):
    """(where the build is present: the fixture has not drifted from it)"""
    for p, want in vectors["peak2hue"]:
        assert _bits(ref.coolmic_util_peak2hue(p, b"default")) == _bits(float.fromhex(want))
    for p, want in vectors["power2hue"]:
        assert _bits(ref.coolmic_util_power2hue(float.fromhex(p), b"default")) == _bits(float.fromhex(want))
    for a, h, s, v, want in vectors["ahsv2argb"]:
        assert ref.coolmic_util_ahsv2argb(*(float.fromhex(x) for x in (a, h, s, v))) == want


def test_batch_colour_form_against_the_reference_build(cm, ref):
    """the batch form the GPU test uses (coolmic_util_vu_argb) on results that cover every branch"""
    vals = [(-30.0, 100), (-3.0111266389980154, 30500), (0.0, -32768), (-19.5, 28001), (-math.inf, 0), (-0.25, 32767),
            (-12.0, -30001), (-20.0, -28000)]
    res = (cm.VuResult * len(vals))()
    for i, (pw, pk) in enumerate(vals):
        res[i].global_power, res[i].global_peak = pw, pk
    a = (C.c_uint32 * len(vals))()
    b = (C.c_uint32 * len(vals))()
    cm.lib.coolmic_util_vu_argb(res, len(vals), b"default", a, b)
    for i, (pw, pk) in enumerate(vals):
        assert a[i] == ref.coolmic_util_ahsv2argb(1.0, ref.coolmic_util_power2hue(pw, b"default"), 1.0, 1.0), i
        assert b[i] == ref.coolmic_util_ahsv2argb(1.0, ref.coolmic_util_peak2hue(pk, b"default"), 1.0, 1.0), i
