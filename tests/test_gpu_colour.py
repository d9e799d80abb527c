"""SURVEY 8f-4 on real results: the VU colours of a whole batch (ref: src/util.c:59-138, delivered per
event at ref: src/simple.c:486-491).  The windows come from the HIP path; the batch form of the helpers
(coolmic_util_vu_argb, one call for all meters) is compared with the per-event helpers the reference
has and with the oracle's restatement.  Parity unpinned: the reference holds no vectors for util.c."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_batch_results_coloured_in_one_call(gpu, oracle):
    cm, lib, o = gpu, gpu.lib, oracle.lib
    # the library says which path computes the windows (ref: src/coolmic-dsp.c:64-83)
    assert lib.coolmic_feature_check(b"accel:hip/gfx950") == 1
    assert b"accel:hip/gfx950" in lib.coolmic_features().split(b" ")
    S, Cn, T = 2048, 2, 4800
    b = cm.Batch(S, Cn, T, flags=cm.VU)
    # levels from silence to clipping, so that every branch of power2hue / peak2hue is met
    rng = np.random.default_rng(5)
    gains = rng.integers(0, 2200, size=S)
    for s in range(S):
        assert b.set_gain(s, 1, 1000, [int(gains[s])]) == 0
    b.generate(cm.GEN_NOISE, 4242, T)
    for s in (0, 7):                              # silence: power -inf, peak 0
        b.upload(s, np.zeros(T * Cn, dtype=np.int16))
    b.run(T)
    res, rcs = b.vu_results()
    assert all(rc == 0 for rc in rcs)
    pw = (C.c_uint32 * S)()
    pk = (C.c_uint32 * S)()
    lib.coolmic_util_vu_argb(res, S, b"default", pw, pk)
    seen = set()
    for s in range(S):
        r = res[s]
        hp = lib.coolmic_util_power2hue(r.global_power, b"default")
        hk = lib.coolmic_util_peak2hue(r.global_peak, b"default")
        assert pw[s] == lib.coolmic_util_ahsv2argb(1.0, hp, 1.0, 1.0), s
        assert pk[s] == lib.coolmic_util_ahsv2argb(1.0, hk, 1.0, 1.0), s
        assert pw[s] == o.oracle_ahsv2argb(1.0, o.oracle_power2hue(r.global_power), 1.0, 1.0), s
        assert pk[s] == o.oracle_ahsv2argb(1.0, o.oracle_peak2hue(r.global_peak), 1.0, 1.0), s
        seen.add(hk)
    assert len(seen) >= 3                         # clipped, hot, and quiet meters were all there
    assert res[0].global_peak == 0 and pk[0] == lib.coolmic_util_ahsv2argb(1.0, 2 * np.pi / 3, 1.0, 1.0)
    # either output may be left out
    lib.coolmic_util_vu_argb(res, S, b"default", None, pk)
    lib.coolmic_util_vu_argb(res, S, b"default", pw, None)
    b.close()
