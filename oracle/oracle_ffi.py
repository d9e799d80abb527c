"""ctypes view of oracle/liboracle.so (the CPU restatement of the reference path).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product (libcoolmic-dsp_amd/) never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

MAX_CH = 16
ssize_t = C.c_ssize_t
READ_FN = C.CFUNCTYPE(ssize_t, C.c_void_p, C.c_void_p, C.c_size_t)
EOF_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)


class Handle(C.Structure):
    _fields_ = [("userdata", C.c_void_p), ("read", READ_FN), ("eof", EOF_FN)]


class Gain(C.Structure):
    _fields_ = [("scale", C.c_uint16), ("gain", C.c_uint16 * MAX_CH)]


class Transform(C.Structure):
    _fields_ = [("io", C.POINTER(Handle)), ("carry", C.c_ubyte * (2 * MAX_CH - 1)),
                ("carry_fill", C.c_size_t), ("channels", C.c_uint), ("gain", Gain)]


class VuResult(C.Structure):
    # same layout as coolmic_vumeter_result_t (ref: include/coolmic-dsp/vumeter.h:48-83)
    _fields_ = [("rate", C.c_uint32), ("channels", C.c_uint), ("frames", C.c_size_t),
                ("global_peak", C.c_int16), ("global_power", C.c_double),
                ("channel_peak", C.c_int16 * MAX_CH), ("channel_power", C.c_double * MAX_CH)]


class Vumeter(C.Structure):
    _fields_ = [("inp", C.POINTER(Handle)), ("rate", C.c_uint32), ("channels", C.c_uint),
                ("buffer", C.c_ubyte * (2 * MAX_CH * 32)), ("fill", C.c_size_t),
                ("power", C.c_int64 * MAX_CH), ("result", VuResult)]


class Sine(C.Structure):
    _fields_ = [("table", C.c_int16 * 96), ("len", C.c_size_t), ("pos", C.c_size_t)]


class MemSrc(C.Structure):
    _fields_ = [("data", C.c_void_p), ("len", C.c_size_t), ("pos", C.c_size_t),
                ("chunk", C.c_size_t)]


class Biquad(C.Structure):
    _fields_ = [("b0", C.c_float), ("b1", C.c_float), ("b2", C.c_float),
                ("a1", C.c_float), ("a2", C.c_float)]


def build():
    """(Re)build liboracle.so with the committed Makefile.  Building is not using."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def load():
    if not os.path.exists(_SO):
        build()
    lib = C.CDLL(_SO)
    P = C.POINTER
    sig = {
        "oracle_handle_read": (ssize_t, [P(Handle), C.c_void_p, C.c_size_t]),
        "oracle_handle_eof": (C.c_int, [P(Handle)]),
        "oracle_gain_set": (C.c_int, [P(Gain), C.c_uint, C.c_uint, C.c_uint16, P(C.c_uint16)]),
        "oracle_gain_apply": (None, [P(Gain), C.c_void_p, C.c_size_t, C.c_uint]),
        "oracle_transform_init": (None, [P(Transform), C.c_uint, P(Handle)]),
        "oracle_transform_read": (ssize_t, [C.c_void_p, C.c_void_p, C.c_size_t]),
        "oracle_transform_eof": (C.c_int, [C.c_void_p]),
        "oracle_vumeter_init": (None, [P(Vumeter), C.c_uint32, C.c_uint, P(Handle)]),
        "oracle_vumeter_reset": (None, [P(Vumeter)]),
        "oracle_vumeter_read": (ssize_t, [P(Vumeter), ssize_t]),
        "oracle_vumeter_result": (C.c_int, [P(Vumeter), P(VuResult)]),
        "oracle_vumeter_accumulate": (None, [P(Vumeter), C.c_void_p, C.c_size_t]),
        "oracle_power_db": (C.c_double, [C.c_int64, C.c_uint64]),
        "oracle_sine_table": (C.c_int, [C.c_uint32, C.c_void_p, P(C.c_size_t)]),
        "oracle_sine_init": (C.c_int, [P(Sine), C.c_uint32]),
        "oracle_sine_read": (ssize_t, [C.c_void_p, C.c_void_p, C.c_size_t]),
        "oracle_null_read": (ssize_t, [C.c_void_p, C.c_void_p, C.c_size_t]),
        "oracle_memsrc_read": (ssize_t, [C.c_void_p, C.c_void_p, C.c_size_t]),
        "oracle_memsrc_eof": (C.c_int, [C.c_void_p]),
        "oracle_lcg_fill": (C.c_uint32, [C.c_uint32, C.c_void_p, C.c_size_t]),
        "oracle_lcg_skip": (C.c_uint32, [C.c_uint32, C.c_uint64]),
        "oracle_chmap_apply": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint]),
        "oracle_i16_to_f32_planar": (None, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t,
                                            C.c_uint]),
        "oracle_f32_to_i16": (C.c_int16, [C.c_float]),
        "oracle_biquad_run": (None, [P(Biquad), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
        "oracle_biquad_design": (None, [P(Biquad), C.c_int, C.c_double, C.c_double, C.c_double,
                                        C.c_double]),
        "oracle_eq3_design": (None, [P(Biquad), C.c_double]),
        "oracle_eq_run_mono": (None, [P(Gain), P(Biquad), C.c_uint, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_size_t]),
        "oracle_ahsv2argb": (C.c_uint32, [C.c_double, C.c_double, C.c_double, C.c_double]),
        "oracle_power2hue": (C.c_double, [C.c_double]),
        "oracle_peak2hue": (C.c_double, [C.c_int16]),
        "oracle_bench_block": (C.c_double, [C.c_uint, C.c_uint, C.c_uint, C.c_size_t, C.c_void_p,
                                            C.c_uint16, P(C.c_uint16), C.c_uint32,
                                            P(C.c_uint64)]),
        "oracle_bench_chain": (C.c_double, [C.c_size_t, C.c_uint16, C.c_uint16, P(C.c_uint64)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


def load_ref_util():
    """oracle/_ref/libref_util.so -- the reference's own src/util.c compiled by `make -C oracle _ref`
    (only where the reference's sources are present; the built file travels to the GPU box).  None when absent."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libref_util.so")
    if not os.path.exists(path):
        return None
    ref = C.CDLL(path)
    ref.coolmic_util_ahsv2argb.restype = C.c_uint32
    ref.coolmic_util_ahsv2argb.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double]
    ref.coolmic_util_power2hue.restype = C.c_double
    ref.coolmic_util_power2hue.argtypes = [C.c_double, C.c_char_p]
    ref.coolmic_util_peak2hue.restype = C.c_double
    ref.coolmic_util_peak2hue.argtypes = [C.c_int16, C.c_char_p]
    return ref


LOG_CB = C.CFUNCTYPE(C.c_int, C.c_int, C.c_char_p)


def bind_core(lib):
    """prototypes of <coolmic-dsp/coolmic-dsp.h> / <coolmic-dsp/logging.h> on a library that defines them"""
    lib.coolmic_error2string.restype = C.c_char_p
    lib.coolmic_error2string.argtypes = [C.c_int]
    lib.coolmic_features.restype = C.c_char_p
    lib.coolmic_features.argtypes = []
    lib.coolmic_feature_check.restype = C.c_int
    lib.coolmic_feature_check.argtypes = [C.c_char_p]
    lib.coolmic_logging_level2string.restype = C.c_char_p
    lib.coolmic_logging_level2string.argtypes = [C.c_int]
    lib.coolmic_logging_set_cb_simple.restype = C.c_int
    lib.coolmic_logging_set_cb_simple.argtypes = [LOG_CB]
    lib.coolmic_logging_log_real.restype = C.c_int       # (variadic: arguments are converted per call)
    return lib


def load_ref_core():
    """oracle/_ref/libref_core.so -- the reference's own src/coolmic-dsp.c + src/logging.c compiled by
    `make -C oracle _ref` (with HAVE_ENC_OPUS and HAVE_SNDDRV_DRIVER_STDIO).  None when absent."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libref_core.so")
    if not os.path.exists(path):
        return None
    return bind_core(C.CDLL(path))


def log_message(lib, file, line, component, level, error, text):
    """what `lib` hands its log callback for one coolmic_logging_log_real("%s", text) call: (rc, [(level, msg)])"""
    seen = []
    cb = LOG_CB(lambda lvl, msg: seen.append((lvl, msg)) or 0)
    lib.coolmic_logging_set_cb_simple(cb)
    try:
        rc = lib.coolmic_logging_log_real(C.c_char_p(file), C.c_ulong(line), C.c_char_p(component), C.c_int(level),
                                          C.c_int(error), C.c_char_p(b"%s" if text is not None else None),
                                          C.c_char_p(text))
    finally:
        lib.coolmic_logging_set_cb_simple(LOG_CB())
    return rc, seen


# ---------------------------------------------------------------------------
# numpy-level helpers used by the parity tests


class Oracle:
    """Convenience wrapper: block-level reference results for numpy inputs."""

    def __init__(self):
        self.lib = load()

    # -- parameters -----------------------------------------------------
    def gain(self, stream_channels, channels, scale, gains):
        g = Gain()
        arr = (C.c_uint16 * max(1, len(gains)))(*gains) if gains is not None else None
        rc = self.lib.oracle_gain_set(C.byref(g), stream_channels, channels, scale, arr)
        return rc, g

    # -- block ops ------------------------------------------------------
    def gain_apply(self, g, pcm, channels):
        out = np.ascontiguousarray(pcm, dtype=np.int16).copy()
        self.lib.oracle_gain_apply(C.byref(g), out.ctypes.data, out.size // channels, channels)
        return out

    def chmap(self, cmap, pcm, channels):
        src = np.ascontiguousarray(pcm, dtype=np.int16)
        out = np.empty_like(src)
        m = np.asarray(cmap, dtype=np.uint8)
        self.lib.oracle_chmap_apply(m.ctypes.data, src.ctypes.data, out.ctypes.data,
                                    src.size // channels, channels)
        return out

    def vu_new(self, channels, rate=48000):
        v = Vumeter()
        self.lib.oracle_vumeter_init(C.byref(v), rate, channels, None)
        return v

    def vu_accumulate(self, v, pcm):
        src = np.ascontiguousarray(pcm, dtype=np.int16)
        self.lib.oracle_vumeter_accumulate(C.byref(v), src.ctypes.data, src.size // v.channels)

    def vu_result(self, v):
        r = VuResult()
        rc = self.lib.oracle_vumeter_result(C.byref(v), C.byref(r))
        return rc, r

    def lcg(self, seed, samples):
        out = np.empty(samples, dtype=np.int16)
        self.lib.oracle_lcg_fill(seed & 0xFFFFFFFF, out.ctypes.data, samples)
        return out

    def sine_table(self, rate):
        buf = np.zeros(96, dtype=np.int16)
        n = C.c_size_t()
        rc = self.lib.oracle_sine_table(rate, buf.ctypes.data, C.byref(n))
        return rc, buf[: n.value].copy()

    def to_f32_planar(self, pcm, channels):
        src = np.ascontiguousarray(pcm, dtype=np.int16)
        frames = src.size // channels
        out = np.empty((channels, frames), dtype=np.float32)
        self.lib.oracle_i16_to_f32_planar(src.ctypes.data, out.ctypes.data, frames, frames,
                                          channels)
        return out

    def eq3(self, rate=48000.0):
        q = (Biquad * 3)()
        self.lib.oracle_eq3_design(q, rate)
        return q

    def eq_run_mono(self, g, q, nsec, state, pcm, want_i16=True):
        src = np.ascontiguousarray(pcm, dtype=np.int16)
        of = np.empty(src.size, dtype=np.float32)
        oi = np.empty(src.size, dtype=np.int16) if want_i16 else None
        self.lib.oracle_eq_run_mono(C.byref(g) if g is not None else None, q, nsec,
                                    state.ctypes.data, src.ctypes.data, of.ctypes.data,
                                    oi.ctypes.data if oi is not None else None, src.size)
        return of, oi


def vu_result_dict(r):
    ch = r.channels
    return {
        "rate": r.rate, "channels": ch, "frames": r.frames, "global_peak": r.global_peak,
        "global_power": r.global_power,
        "channel_peak": [r.channel_peak[i] for i in range(ch)],
        "channel_power": [r.channel_power[i] for i in range(ch)],
    }
