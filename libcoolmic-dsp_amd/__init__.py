"""ctypes view of lib/libcoolmic-dsp-hip.so for tests and bench.py.

Two layers, both thin:
  * `Batch` mirrors the C ABI of include/coolmic_hip.h (the MI355X batch engine);
  * `Transform`, `Vumeter`, `Snddev`, `IoHandle` mirror the reference's per-stream
    operator API of include/coolmic-dsp/*.h (same function names underneath, same
    argument meaning and error numbers), so tests read like tests of the reference.

There is no fallback of any kind: if the shared object is missing, import fails.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("COOLMIC_HIP_LIB") or os.path.join(_HERE, "lib", "libcoolmic-dsp-hip.so")
if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make -C {_HERE}` "
        "(or __graft_entry__.build()); there is no CPU fallback")

lib = C.CDLL(LIB_PATH)

MAX_CH = 16
ssize_t = C.c_ssize_t

# error numbers (include/coolmic-dsp/coolmic-dsp.h)
ERROR_NONE, ERROR_GENERIC, ERROR_NOSYS, ERROR_FAULT = 0, -1, -8, -9
ERROR_INVAL, ERROR_NOMEM, ERROR_BUSY = -10, -11, -12

OUT_PCM, OUT_F32, VU, INPLACE, EQ, HOSTPCM, EXTSLOTS = 0x1, 0x2, 0x4, 0x8, 0x10, 0x20, 0x40
PLACE_SEARCH = 0x80
GEN_NULL, GEN_SINE, GEN_NOISE = 0, 1, 2
NODE_WORDS = 34

READ_FN = C.CFUNCTYPE(ssize_t, C.c_void_p, C.c_void_p, C.c_size_t)
EOF_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)
FREE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)
LOG_FN = C.CFUNCTYPE(C.c_int, C.c_int, C.c_char_p)


class VuResult(C.Structure):
    """coolmic_vumeter_result_t (include/coolmic-dsp/vumeter.h)"""
    _fields_ = [("rate", C.c_uint32), ("channels", C.c_uint), ("frames", C.c_size_t),
                ("global_peak", C.c_int16), ("global_power", C.c_double),
                ("channel_peak", C.c_int16 * MAX_CH), ("channel_power", C.c_double * MAX_CH)]

    def as_dict(self):
        ch = self.channels
        return {"rate": self.rate, "channels": ch, "frames": self.frames,
                "global_peak": self.global_peak, "global_power": self.global_power,
                "channel_peak": [self.channel_peak[i] for i in range(ch)],
                "channel_power": [self.channel_power[i] for i in range(ch)]}


class BatchDesc(C.Structure):
    _fields_ = [("device", C.c_int), ("streams", C.c_uint), ("channels", C.c_uint),
                ("rate", C.c_uint), ("max_frames", C.c_size_t), ("flags", C.c_uint),
                ("hip_stream", C.c_void_p)]


class Placement(C.Structure):
    """cmhip_placement_t (include/coolmic_hip.h)"""
    _fields_ = [("searched", C.c_int), ("candidates", C.c_int), ("chosen_in", C.c_int),
                ("chosen_out", C.c_int), ("probe_launches", C.c_int), ("first_pair_ms", C.c_double),
                ("best_pair_ms", C.c_double), ("search_ms", C.c_double),
                ("bytes_requested", C.c_uint64), ("bytes_free_before", C.c_uint64)]

    def as_dict(self):
        return {"searched": bool(self.searched), "candidates": self.candidates,
                "chosen": [self.chosen_in, self.chosen_out], "probe_launches": self.probe_launches,
                "first_pair_ms": round(self.first_pair_ms, 4), "best_pair_ms": round(self.best_pair_ms, 4),
                "search_ms": round(self.search_ms, 1),
                "GiB_requested": round(self.bytes_requested / 2**30, 2),
                "GiB_free_before": round(self.bytes_free_before / 2**30, 2)}


def _sig(name, res, args):
    fn = getattr(lib, name)
    fn.restype = res
    fn.argtypes = args
    return fn


_P = C.POINTER
_vp = C.c_void_p
# every symbol include/*.h declares; tests/test_abi.py checks this table against the headers
SIGNATURES = {
    # include/coolmic_hip.h
    "cmhip_device_count": (C.c_int, []),
    "cmhip_device_synchronize": (C.c_int, [C.c_int]),
    "cmhip_device_mem_info": (C.c_int, [C.c_int, _P(C.c_size_t), _P(C.c_size_t)]),
    "cmhip_device_alloc": (_vp, [C.c_int, C.c_size_t]),
    "cmhip_device_free": (None, [C.c_int, _vp]),
    "cmhip_device_read": (C.c_int, [C.c_int, _vp, _vp, C.c_size_t]),
    "cmhip_last_error": (C.c_char_p, []),
    "cmhip_version": (C.c_char_p, []),
    "cmhip_batch_new": (_vp, [_P(BatchDesc)]),
    "cmhip_batch_free": (None, [_vp]),
    "cmhip_batch_placement": (C.c_int, [_vp, _P(Placement)]),
    "cmhip_batch_set_gain": (C.c_int, [_vp, C.c_long, C.c_uint, C.c_uint16, _P(C.c_uint16)]),
    "cmhip_batch_set_chmap": (C.c_int, [_vp, C.c_long, _vp]),
    "cmhip_batch_set_eq": (C.c_int, [_vp, C.c_long, C.c_uint, _vp]),
    "cmhip_batch_eq_reset": (C.c_int, [_vp, C.c_long]),
    "cmhip_design_biquad": (None, [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, _vp]),
    "cmhip_batch_stride": (C.c_size_t, [_vp]),
    "cmhip_batch_max_frames": (C.c_size_t, [_vp]),
    "cmhip_batch_dev_in": (_vp, [_vp]),
    "cmhip_batch_dev_out": (_vp, [_vp]),
    "cmhip_batch_dev_f32": (_vp, [_vp]),
    "cmhip_batch_hip_stream": (_vp, [_vp]),
    "cmhip_batch_upload": (C.c_int, [_vp, C.c_uint, _vp, C.c_size_t]),
    "cmhip_batch_download": (C.c_int, [_vp, C.c_uint, _vp, C.c_size_t]),
    "cmhip_batch_upload_all": (C.c_int, [_vp, _vp, C.c_size_t]),
    "cmhip_batch_download_all": (C.c_int, [_vp, _vp, C.c_size_t]),
    "cmhip_host_alloc": (_vp, [C.c_size_t]),
    "cmhip_host_free": (None, [_vp]),
    "cmhip_batch_download_input": (C.c_int, [_vp, C.c_uint, _vp, C.c_size_t]),
    "cmhip_batch_download_f32": (C.c_int, [_vp, C.c_uint, C.c_uint, _vp, C.c_size_t]),
    "cmhip_batch_generate": (C.c_int, [_vp, C.c_int, C.c_uint32, C.c_size_t, C.c_uint64,
                                       C.c_uint64, C.c_uint64]),
    "cmhip_batch_run": (C.c_int, [_vp, C.c_size_t, _vp]),
    "cmhip_batch_sync": (C.c_int, [_vp]),
    "cmhip_batch_vu_result": (C.c_int, [_vp, C.c_uint, _P(VuResult)]),
    "cmhip_batch_vu_results": (C.c_int, [_vp, _vp, _vp]),
    "cmhip_batch_vu_snapshot": (C.c_int, [_vp]),
    "cmhip_batch_vu_collect": (C.c_int, [_vp, _vp, _vp]),
    "cmhip_batch_vu_collect_begin": (C.c_int, [_vp, _vp, _vp]),
    "cmhip_batch_vu_collect_end": (C.c_int, [_vp]),
    "cmhip_batch_vu_reset": (C.c_int, [_vp, C.c_long]),
    "cmhip_batch_vu_raw": (C.c_int, [_vp, C.c_uint, _vp, _vp, _P(C.c_uint64)]),
    "cmhip_batch_vu_node_partial": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint64]),
    "cmhip_batch_vu_node_record": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint64]),
    "cmhip_node_finish": (C.c_int, [_vp, C.c_uint, C.c_uint, _P(VuResult)]),
    "cmhip_batch_run_slots": (C.c_int, [_vp, C.c_size_t, _vp, _vp, _vp]),
    "cmhip_host_alloc_mapped": (_vp, [C.c_size_t, _P(_vp)]),
    "cmhip_node_unique_id": (C.c_int, [_vp]),
    "cmhip_node_new": (_vp, [C.c_int, C.c_int, C.c_int, _vp, C.c_uint]),
    "cmhip_node_free": (None, [_vp]),
    "cmhip_node_ranks": (C.c_int, [_vp]),
    "cmhip_node_runtime": (C.c_char_p, []),
    "cmhip_node_partial": (C.c_int, [_vp, _vp, C.c_uint, C.c_uint, C.c_uint64, C.c_uint64]),
    "cmhip_node_allreduce": (C.c_int, [_vp, C.c_uint, C.c_uint, _vp]),
    "cmhip_node_fetch": (C.c_int, [_vp, C.c_uint, C.c_uint, _vp]),
    "cmhip_node_merge_host": (C.c_int, [_vp, C.c_uint, _vp]),
    "cmhip_batch_timing": (C.c_int, [_vp, C.c_int]),
    "cmhip_batch_timing_read": (C.c_int, [_vp, _P(C.c_double), _P(C.c_uint)]),
    "cmhip_batch_ceiling": (C.c_double, [_vp, C.c_int, C.c_size_t, C.c_int]),
    # include/coolmic-dsp/ro-compat.h
    "coolmic_ro_new_raw": (_vp, [_vp, C.c_char_p, _vp]),
    "coolmic_ro_ref": (C.c_int, [_vp]),
    "coolmic_ro_unref": (C.c_int, [_vp]),
    "coolmic_ro_refcount": (C.c_uint, [_vp]),
    # include/coolmic-dsp/coolmic-dsp.h
    "coolmic_error2string": (C.c_char_p, [C.c_int]),
    "coolmic_features": (C.c_char_p, []),
    "coolmic_feature_check": (C.c_int, [C.c_char_p]),
    # include/coolmic-dsp/logging.h
    "coolmic_logging_level2string": (C.c_char_p, [C.c_int]),
    "coolmic_logging_log_real": (C.c_int, None),
    "coolmic_logging_set_cb_simple": (C.c_int, [LOG_FN]),
    # include/coolmic-dsp/iohandle.h
    "coolmic_iohandle_new": (_vp, [C.c_char_p, _vp, _vp, FREE_FN, READ_FN, EOF_FN]),
    "coolmic_iohandle_read": (ssize_t, [_vp, _vp, C.c_size_t]),
    "coolmic_iohandle_eof": (C.c_int, [_vp]),
    # include/coolmic-dsp/transform.h
    "coolmic_transform_new": (_vp, [C.c_char_p, _vp, C.c_uint32, C.c_uint]),
    "coolmic_transform_attach_iohandle": (C.c_int, [_vp, _vp]),
    "coolmic_transform_get_iohandle": (_vp, [_vp]),
    "coolmic_transform_set_master_gain": (C.c_int, [_vp, C.c_uint, C.c_uint16, _P(C.c_uint16)]),
    "coolmic_transform_set_channel_map": (C.c_int, [_vp, _vp]),
    "coolmic_transform_set_eq": (C.c_int, [_vp, C.c_uint, _vp]),
    # include/coolmic-dsp/vumeter.h
    "coolmic_vumeter_new": (_vp, [C.c_char_p, _vp, C.c_uint32, C.c_uint]),
    "coolmic_vumeter_reset": (C.c_int, [_vp]),
    "coolmic_vumeter_attach_iohandle": (C.c_int, [_vp, _vp]),
    "coolmic_vumeter_read": (ssize_t, [_vp, ssize_t]),
    "coolmic_vumeter_result": (C.c_int, [_vp, _P(VuResult)]),
    # include/coolmic-dsp/snddev.h
    "coolmic_snddev_new": (_vp, [C.c_char_p, _vp, C.c_char_p, _vp, C.c_uint32, C.c_uint, C.c_int,
                                 ssize_t]),
    "coolmic_snddev_get_iohandle": (_vp, [_vp]),
    "coolmic_snddev_attach_iohandle": (C.c_int, [_vp, _vp]),
    "coolmic_snddev_iter": (C.c_int, [_vp]),
    # include/coolmic-dsp/tee.h
    "coolmic_tee_new": (_vp, [C.c_char_p, _vp, C.c_size_t]),
    "coolmic_tee_attach_iohandle": (C.c_int, [_vp, _vp]),
    "coolmic_tee_get_iohandle": (_vp, [_vp, ssize_t]),
    # include/coolmic-dsp/util.h
    "coolmic_util_ahsv2argb": (C.c_uint32, [C.c_double, C.c_double, C.c_double, C.c_double]),
    "coolmic_util_power2hue": (C.c_double, [C.c_double, C.c_char_p]),
    "coolmic_util_peak2hue": (C.c_double, [C.c_int16, C.c_char_p]),
    "coolmic_util_vu_argb": (None, [_vp, C.c_size_t, C.c_char_p, _vp, _vp]),
    # include/coolmic-dsp/group.h
    "coolmic_group_new": (_vp, [C.c_char_p, _vp, C.c_uint32, C.c_uint, C.c_uint, C.c_size_t, C.c_uint]),
    "coolmic_group_new_on": (_vp, [C.c_int, C.c_char_p, _vp, C.c_uint32, C.c_uint, C.c_uint, C.c_size_t, C.c_uint]),
    "coolmic_group_device": (C.c_int, [_vp]),
    "coolmic_group_engine": (_vp, [_vp]),
    "coolmic_transform_set_device": (C.c_int, [_vp, C.c_int]),
    "coolmic_vumeter_set_device": (C.c_int, [_vp, C.c_int]),
    "cmhip_host_alloc_mapped_on": (_vp, [C.c_int, C.c_size_t, _P(_vp)]),
    "coolmic_group_add_stream": (C.c_int, [_vp, _vp]),
    "coolmic_group_set_master_gain": (C.c_int, [_vp, C.c_uint, C.c_uint, C.c_uint16, _P(C.c_uint16)]),
    "coolmic_group_set_channel_map": (C.c_int, [_vp, C.c_uint, _vp]),
    "coolmic_group_set_eq": (C.c_int, [_vp, C.c_int, C.c_uint, _vp]),
    "coolmic_group_get_iohandle": (_vp, [_vp, C.c_uint]),
    "coolmic_group_pump": (C.c_int, [_vp]),
    "coolmic_group_set_pull_threads": (C.c_int, [_vp, C.c_uint]),
    "coolmic_group_vumeter_result": (C.c_int, [_vp, C.c_uint, _P(VuResult)]),
    "coolmic_group_streams": (C.c_uint, [_vp]),
}
MISSING = []        # entry points this build of the library lacks (an older build under tools/ab_two_libs.py)
for _name, (_res, _args) in SIGNATURES.items():
    if not hasattr(lib, _name):
        MISSING.append(_name)
        continue
    _fn = getattr(lib, _name)
    _fn.restype = _res
    if _args is not None:
        _fn.argtypes = _args

# not in a public header: host-logic test hooks
if hasattr(lib, "cmhip_test_gain_consts"):      # (an older build loaded by tools/ab_two_libs.py has another hook)
    lib.cmhip_test_gain_consts.restype = None
    lib.cmhip_test_gain_consts.argtypes = [C.c_uint16, C.c_uint16, _P(C.c_uint16), _P(C.c_uint32)]
lib.cmhip_test_merge_windows.restype = C.c_int
lib.cmhip_test_merge_windows.argtypes = [_P(C.c_uint64), C.c_uint, C.c_uint, C.c_uint, _P(VuResult)]
lib.cmhip_debug_run_count.restype = C.c_ulonglong
lib.cmhip_debug_run_count.argtypes = []
lib.coolmic_debug_vumeter_mode.restype = C.c_int
lib.coolmic_debug_vumeter_mode.argtypes = [_vp]
lib.coolmic_sine_period.restype = C.c_int
lib.coolmic_sine_period.argtypes = [C.c_uint32, _vp, _P(C.c_size_t)]


class CoolmicError(RuntimeError):
    def __init__(self, what, code):
        text = lib.coolmic_error2string(code).decode()
        detail = lib.cmhip_last_error().decode()
        super().__init__(f"{what}: {code} ({text}) {detail}")
        self.code = code


def _check(what, rc):
    if rc != ERROR_NONE:
        raise CoolmicError(what, rc)


def device_count():
    return lib.cmhip_device_count()


def device_synchronize(device=0):
    _check("device_synchronize", lib.cmhip_device_synchronize(device))


def device_mem_info(device=0):
    f, t = C.c_size_t(), C.c_size_t()
    _check("device_mem_info", lib.cmhip_device_mem_info(device, C.byref(f), C.byref(t)))
    return f.value, t.value


def last_error():
    return lib.cmhip_last_error().decode()


def gain_consts(gain, scale):
    """Test hook: the division constants the kernels use for gain / scale -- integer part mi and fraction mf with
    floor(|x| * gain / scale) == |x| * mi + ((|x| * mf) >> 32) for every |x| <= 32768 (StreamParam, cmhip_internal.h)."""
    mi, mf = C.c_uint16(), C.c_uint32()
    lib.cmhip_test_gain_consts(gain, scale, C.byref(mi), C.byref(mf))
    return mi.value, mf.value


def merge_windows(windows, channels, rate=48000):
    """Test hook: raw VU windows (rows of 33 uint64: 16 sums of squares, 16 peak keys, samples accounted), one
    per launch in stream order, merged on the host as a meter behind a tee merges them, and finished."""
    import numpy as np
    w = np.ascontiguousarray(windows, dtype=np.uint64).reshape(-1, 33)
    r = VuResult()
    rc = lib.cmhip_test_merge_windows(w.ctypes.data_as(_P(C.c_uint64)), w.shape[0], channels, rate, C.byref(r))
    return rc, r


def sine_period(rate):
    buf = np.zeros(96, dtype=np.int16)
    n = C.c_size_t()
    rc = lib.coolmic_sine_period(rate, buf.ctypes.data, C.byref(n))
    return rc, buf[: n.value].copy()


def design_biquad(kind, rate, freq, gain_db, q=0.0):
    out = np.zeros(5, dtype=np.float32)
    lib.cmhip_design_biquad(kind, rate, freq, gain_db, q, out.ctypes.data)
    return out


def eq3(rate=48000.0):
    """the 3-band EQ of BASELINE config 3"""
    return np.concatenate([design_biquad(0, rate, 200.0, 3.0),
                           design_biquad(1, rate, 1000.0, -2.0, 1.0),
                           design_biquad(2, rate, 6000.0, 2.0)]).astype(np.float32)


# ---------------------------------------------------------------------------
# the batch engine


class Batch:
    def __init__(self, streams, channels, max_frames, flags=OUT_PCM | VU, rate=48000, device=0,
                 hip_stream=None):
        d = BatchDesc(device, streams, channels, rate, max_frames, flags, hip_stream)
        self.h = lib.cmhip_batch_new(C.byref(d))
        if not self.h:
            raise CoolmicError("cmhip_batch_new", ERROR_GENERIC)
        self.streams, self.channels, self.max_frames, self.flags = streams, channels, max_frames, flags
        self.rate = rate

    def close(self):
        if self.h:
            lib.cmhip_batch_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def placement(self):
        """what the placement search of CMHIP_PLACE_SEARCH did at creation (a dict)"""
        p = Placement()
        _check("placement", lib.cmhip_batch_placement(self.h, C.byref(p)))
        return p.as_dict()

    # parameters
    def set_gain(self, stream, channels, scale, gains):
        arr = (C.c_uint16 * len(gains))(*gains) if gains is not None and len(gains) else None
        return lib.cmhip_batch_set_gain(self.h, stream, channels, scale, arr)

    def set_chmap(self, stream, cmap):
        if cmap is None:
            return lib.cmhip_batch_set_chmap(self.h, stream, None)
        m = np.asarray(cmap, dtype=np.uint8)
        return lib.cmhip_batch_set_chmap(self.h, stream, m.ctypes.data)

    def set_eq(self, stream, coef):
        c = np.ascontiguousarray(coef, dtype=np.float32).reshape(-1)
        return lib.cmhip_batch_set_eq(self.h, stream, c.size // 5, c.ctypes.data if c.size else None)

    def eq_reset(self, stream=-1):
        _check("eq_reset", lib.cmhip_batch_eq_reset(self.h, stream))

    # data
    def upload(self, stream, pcm):
        a = np.ascontiguousarray(pcm, dtype=np.int16)
        _check("upload", lib.cmhip_batch_upload(self.h, stream, a.ctypes.data, a.size // self.channels))

    def download(self, stream, frames):
        out = np.empty(frames * self.channels, dtype=np.int16)
        _check("download", lib.cmhip_batch_download(self.h, stream, out.ctypes.data, frames))
        return out

    def download_f32(self, stream, channel, frames):
        out = np.empty(frames, dtype=np.float32)
        _check("download_f32", lib.cmhip_batch_download_f32(self.h, stream, channel,
                                                            out.ctypes.data, frames))
        return out

    def upload_all(self, host_ptr, frames):
        _check("upload_all", lib.cmhip_batch_upload_all(self.h, host_ptr, frames))

    def download_all(self, host_ptr, frames):
        _check("download_all", lib.cmhip_batch_download_all(self.h, host_ptr, frames))

    def generate(self, mode, seed, frames, first_global=0, global_step=1, frame_offset=0):
        _check("generate", lib.cmhip_batch_generate(self.h, mode, seed & 0xFFFFFFFF, frames,
                                                    first_global, global_step, frame_offset))

    def download_input(self, stream, frames):
        out = np.empty(frames * self.channels, dtype=np.int16)
        _check("download_input", lib.cmhip_batch_download_input(self.h, stream, out.ctypes.data,
                                                                frames))
        return out

    # hot path
    def run(self, frames, frames_per_stream=None):
        if frames_per_stream is None:
            _check("run", lib.cmhip_batch_run(self.h, frames, None))
        else:
            a = np.ascontiguousarray(frames_per_stream, dtype=np.uint32)
            assert a.size == self.streams
            _check("run", lib.cmhip_batch_run(self.h, frames, a.ctypes.data))

    def run_slots(self, frames, slots_in, slots_out, frames_per_stream=None):
        """one pass over PCM arrays named for this run (device pointers; MappedPcm.dev)"""
        fps = None
        if frames_per_stream is not None:
            fps = (C.c_uint32 * self.streams)(*frames_per_stream)
        _check("run_slots", lib.cmhip_batch_run_slots(self.h, frames, fps, slots_in, slots_out))

    def hip_stream(self):
        """the hipStream_t (as an integer) this batch launches on"""
        return lib.cmhip_batch_hip_stream(self.h) or 0

    def sync(self):
        _check("sync", lib.cmhip_batch_sync(self.h))

    # VU
    def vu_result(self, stream):
        r = VuResult()
        rc = lib.cmhip_batch_vu_result(self.h, stream, C.byref(r))
        return rc, r

    def vu_results(self):
        out = (VuResult * self.streams)()
        rc = (C.c_int * self.streams)()
        _check("vu_results", lib.cmhip_batch_vu_results(self.h, out, rc))
        return out, list(rc)

    def vu_snapshot(self):
        _check("vu_snapshot", lib.cmhip_batch_vu_snapshot(self.h))

    def vu_collect(self, out=None, rc=None):
        out = out if out is not None else (VuResult * self.streams)()
        rc = rc if rc is not None else (C.c_int * self.streams)()
        _check("vu_collect", lib.cmhip_batch_vu_collect(self.h, out, rc))
        return out, rc

    def vu_collect_begin(self, out, rc):
        """first half of vu_collect: the helper threads finish the oldest snapshot into out / rc (ctypes arrays
        that must stay alive) while the caller goes on; vu_collect_end() returns when they are complete"""
        _check("vu_collect_begin", lib.cmhip_batch_vu_collect_begin(self.h, out, rc))

    def vu_collect_end(self):
        _check("vu_collect_end", lib.cmhip_batch_vu_collect_end(self.h))

    def vu_reset(self, stream=-1):
        _check("vu_reset", lib.cmhip_batch_vu_reset(self.h, stream))

    def vu_raw(self, stream):
        power = np.zeros(MAX_CH, dtype=np.int64)
        peak = np.zeros(MAX_CH, dtype=np.int16)
        frames = C.c_uint64()
        _check("vu_raw", lib.cmhip_batch_vu_raw(self.h, stream, power.ctypes.data, peak.ctypes.data,
                                                C.byref(frames)))
        return power, peak, frames.value

    def node_partial(self, dst_device_ptr, first_global=0, global_step=1):
        _check("node_partial", lib.cmhip_batch_vu_node_partial(self.h, dst_device_ptr, first_global,
                                                               global_step))

    def node_record(self, first_global=0, global_step=1):
        """this batch's un-reduced node record in host memory (waits for the last run)"""
        out = np.zeros(NODE_WORDS, dtype=np.int64)
        _check("node_record", lib.cmhip_batch_vu_node_record(self.h, out.ctypes.data, first_global, global_step))
        return out

    # measurement
    def timing(self, enable):
        """True / 1: every run carries events; n > 1: every n-th run; False / 0: off"""
        _check("timing", lib.cmhip_batch_timing(self.h, int(enable)))

    def timing_read(self):
        ms, n = C.c_double(), C.c_uint()
        _check("timing_read", lib.cmhip_batch_timing_read(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def ceiling(self, mode, iters=10):
        return lib.cmhip_batch_ceiling(self.h, mode, self.max_frames, iters)

    @property
    def stride(self):
        return lib.cmhip_batch_stride(self.h)

    @property
    def dev_in(self):
        return lib.cmhip_batch_dev_in(self.h)

    @property
    def dev_out(self):
        return lib.cmhip_batch_dev_out(self.h)


class PinnedPcm:
    """pinned host mirror of a batch's PCM slots: numpy view int16 [S][stride]"""

    def __init__(self, batch):
        self.shape = (batch.streams, batch.stride)
        self.nbytes = batch.streams * batch.stride * 2
        self.ptr = lib.cmhip_host_alloc(self.nbytes)
        if not self.ptr:
            raise CoolmicError("cmhip_host_alloc", ERROR_NOMEM)
        buf = (C.c_int16 * (self.nbytes // 2)).from_address(self.ptr)
        self.array = np.frombuffer(buf, dtype=np.int16).reshape(self.shape)

    def free(self):
        if self.ptr:
            self.array = None
            lib.cmhip_host_free(self.ptr)
            self.ptr = None


class DeviceWords:
    """n int64 words of plain device memory (cmhip_device_alloc): `.dev` for the engine, read() for the host"""

    def __init__(self, n, device=0):
        self.n, self.device = n, device
        self.dev = lib.cmhip_device_alloc(device, n * 8)
        if not self.dev:
            raise CoolmicError("cmhip_device_alloc", ERROR_NOMEM)

    def read(self):
        out = np.zeros(self.n, dtype=np.int64)
        _check("device_read", lib.cmhip_device_read(self.device, out.ctypes.data, self.dev, self.n * 8))
        return out

    def free(self):
        if self.dev:
            lib.cmhip_device_free(self.device, self.dev)
            self.dev = None


class MappedPcm:
    """pinned, device-mapped host mirror of a batch's slot layout: numpy view int16 [S][stride] on the
    host, `.dev` for cmhip_batch_run_slots"""

    def __init__(self, batch):
        self.shape = (batch.streams, batch.stride)
        self.nbytes = batch.streams * batch.stride * 2
        dev = C.c_void_p()
        self.ptr = lib.cmhip_host_alloc_mapped(self.nbytes, C.byref(dev))
        if not self.ptr:
            raise CoolmicError("cmhip_host_alloc_mapped", ERROR_NOMEM)
        self.dev = dev.value
        buf = (C.c_int16 * (self.nbytes // 2)).from_address(self.ptr)
        self.array = np.frombuffer(buf, dtype=np.int16).reshape(self.shape)

    def free(self):
        if self.ptr:
            self.array = None
            lib.cmhip_host_free(self.ptr)
            self.ptr = None


def node_finish(words, channels, rate=48000):
    w = np.ascontiguousarray(words, dtype=np.int64)
    assert w.size == NODE_WORDS
    r = VuResult()
    rc = lib.cmhip_node_finish(w.ctypes.data, channels, rate, C.byref(r))
    return rc, r


NODE_ID_BYTES = 128


def node_unique_id():
    """rank 0: the 128-byte id every rank hands to Node() (ncclGetUniqueId)"""
    buf = (C.c_ubyte * NODE_ID_BYTES)()
    _check("node_unique_id", lib.cmhip_node_unique_id(buf))
    return bytes(buf)


def node_runtime():
    """'hip=<path> rccl=<path>': the HIP runtime the engine is bound to and the librccl it resolved"""
    return lib.cmhip_node_runtime().decode()


def node_merge_host(records):
    """SUM / MAX of per-rank node records on the host (the no-collective form)"""
    w = np.ascontiguousarray(records, dtype=np.int64).reshape(-1, NODE_WORDS)
    out = np.zeros(NODE_WORDS, dtype=np.int64)
    _check("node_merge_host", lib.cmhip_node_merge_host(w.ctypes.data, w.shape[0], out.ctypes.data))
    return out


class Node:
    """cmhip_node_t: the node-global VU exchange over RCCL, one per GPU / rank"""

    def __init__(self, device, nranks, rank, unique_id, max_records=8):
        assert len(unique_id) == NODE_ID_BYTES
        self._id = (C.c_ubyte * NODE_ID_BYTES).from_buffer_copy(unique_id)
        self.max_records = max_records
        self.h = lib.cmhip_node_new(device, nranks, rank, self._id, max_records)
        if not self.h:
            raise CoolmicError("cmhip_node_new: " + last_error(), ERROR_GENERIC)

    def close(self):
        if self.h:
            lib.cmhip_node_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def ranks(self):
        return lib.cmhip_node_ranks(self.h)

    def partial(self, batch, set_, slot, first_global=0, global_step=1):
        _check("node_partial", lib.cmhip_node_partial(self.h, batch.h, set_, slot, first_global, global_step))

    def allreduce(self, set_, count, after=None):
        _check("node_allreduce", lib.cmhip_node_allreduce(self.h, set_, count, after.h if after else None))

    def fetch(self, set_, count):
        out = np.zeros((count, NODE_WORDS), dtype=np.int64)
        _check("node_fetch", lib.cmhip_node_fetch(self.h, set_, count, out.ctypes.data))
        return out


# ---------------------------------------------------------------------------
# the reference's per-stream operator API


_alive = {}


class IoHandle:
    """Owns one reference to a coolmic_iohandle_t."""

    def __init__(self, ptr, keep=()):
        if not ptr:
            raise CoolmicError("iohandle", ERROR_GENERIC)
        self.ptr = ptr
        self._keep = keep

    @classmethod
    def from_callbacks(cls, read, eof=None, free=None):
        """read(nbytes) -> bytes | int(<=0); eof() -> int.  The ctypes thunks stay alive
        until the C side runs the handle's free callback (i.e. until the last unref)."""
        token = object()

        def _read(_ud, buf, n):
            got = read(n)
            if isinstance(got, int):
                return got
            got = bytes(got)[:n]
            C.memmove(buf, got, len(got))
            return len(got)

        def _free(_ud):
            if free:
                free()
            _alive.pop(id(token), None)
            return 0

        rcb = READ_FN(_read)
        ecb = EOF_FN((lambda _ud: eof())) if eof else EOF_FN()
        fcb = FREE_FN(_free)
        _alive[id(token)] = (token, rcb, ecb, fcb)
        ptr = lib.coolmic_iohandle_new(None, None, None, fcb, rcb, ecb)
        if not ptr:
            _alive.pop(id(token), None)
        return cls(ptr)

    @classmethod
    def from_bytes(cls, data, chunk=0):
        """serves `data` in pieces of at most `chunk` bytes (0: as asked), then reports EOF"""
        state = {"pos": 0}
        data = bytes(data)

        def read(n):
            k = min(n, len(data) - state["pos"])
            if chunk:
                k = min(k, chunk)
            out = data[state["pos"]: state["pos"] + k]
            state["pos"] += k
            return out

        return cls.from_callbacks(read, eof=lambda: 1 if state["pos"] >= len(data) else 0)

    def read(self, nbytes):
        buf = (C.c_ubyte * max(nbytes, 1))()
        n = lib.coolmic_iohandle_read(self.ptr, buf, nbytes)
        return n, bytes(buf[: max(n, 0)])

    def eof(self):
        return lib.coolmic_iohandle_eof(self.ptr)

    def refcount(self):
        return lib.coolmic_ro_refcount(self.ptr)

    def unref(self):
        if self.ptr:
            lib.coolmic_ro_unref(self.ptr)
            self.ptr = None


class Snddev:
    def __init__(self, driver, rate=48000, channels=1, flags=1, device=None):
        self.ptr = lib.coolmic_snddev_new(None, None, driver.encode() if driver else None,
                                          C.c_char_p(device.encode()) if device else None,
                                          rate, channels, flags, -1)
        if not self.ptr:
            raise CoolmicError("coolmic_snddev_new", ERROR_GENERIC)

    def get_iohandle(self):
        return IoHandle(lib.coolmic_snddev_get_iohandle(self.ptr))

    def attach(self, handle):
        return lib.coolmic_snddev_attach_iohandle(self.ptr, handle.ptr if handle else None)

    def iter(self):
        return lib.coolmic_snddev_iter(self.ptr)

    def unref(self):
        if self.ptr:
            lib.coolmic_ro_unref(self.ptr)
            self.ptr = None


class Transform:
    def __init__(self, rate=48000, channels=1):
        self.ptr = lib.coolmic_transform_new(None, None, rate, channels)
        if not self.ptr:
            raise CoolmicError("coolmic_transform_new", ERROR_GENERIC)
        self.channels = channels

    def attach(self, handle):
        return lib.coolmic_transform_attach_iohandle(self.ptr, handle.ptr if handle else None)

    def get_iohandle(self):
        return IoHandle(lib.coolmic_transform_get_iohandle(self.ptr))

    def set_master_gain(self, channels, scale, gains):
        arr = (C.c_uint16 * len(gains))(*gains) if gains is not None and len(gains) else None
        return lib.coolmic_transform_set_master_gain(self.ptr, channels, scale, arr)

    def set_channel_map(self, cmap):
        if cmap is None:
            return lib.coolmic_transform_set_channel_map(self.ptr, None)
        m = np.asarray(cmap, dtype=np.uint8)
        return lib.coolmic_transform_set_channel_map(self.ptr, m.ctypes.data)

    def set_device(self, device):
        return lib.coolmic_transform_set_device(self.ptr, device)

    def set_eq(self, coef):
        """coef: 5 floats per section (b0 b1 b2 a1 a2), or None / empty to switch the filter off"""
        if coef is None or len(coef) == 0:
            return lib.coolmic_transform_set_eq(self.ptr, 0, None)
        c = np.ascontiguousarray(coef, dtype=np.float32)
        return lib.coolmic_transform_set_eq(self.ptr, c.size // 5, c.ctypes.data)

    def refcount(self):
        return lib.coolmic_ro_refcount(self.ptr)

    def unref(self):
        if self.ptr:
            lib.coolmic_ro_unref(self.ptr)
            self.ptr = None


class Vumeter:
    def __init__(self, rate=48000, channels=1):
        self.ptr = lib.coolmic_vumeter_new(None, None, rate, channels)
        if not self.ptr:
            raise CoolmicError("coolmic_vumeter_new", ERROR_GENERIC)
        self.channels = channels

    def attach(self, handle):
        return lib.coolmic_vumeter_attach_iohandle(self.ptr, handle.ptr if handle else None)

    def read(self, maxlen=-1):
        return lib.coolmic_vumeter_read(self.ptr, maxlen)

    def result(self):
        r = VuResult()
        rc = lib.coolmic_vumeter_result(self.ptr, C.byref(r))
        return rc, r

    def reset(self):
        return lib.coolmic_vumeter_reset(self.ptr)

    def set_device(self, device):
        return lib.coolmic_vumeter_set_device(self.ptr, device)

    def mode(self):
        """test hook: 0 own batch, 1 shares the launch of the transform right above, 2 shares it through a tee"""
        return lib.coolmic_debug_vumeter_mode(self.ptr)

    def unref(self):
        if self.ptr:
            lib.coolmic_ro_unref(self.ptr)
            self.ptr = None


class Tee:
    def __init__(self, readers):
        self.ptr = lib.coolmic_tee_new(None, None, readers)
        if not self.ptr:
            raise CoolmicError("coolmic_tee_new", ERROR_GENERIC)

    def attach(self, handle):
        return lib.coolmic_tee_attach_iohandle(self.ptr, handle.ptr if handle else None)

    def get_iohandle(self, index=-1):
        return IoHandle(lib.coolmic_tee_get_iohandle(self.ptr, index))

    def unref(self):
        if self.ptr:
            lib.coolmic_ro_unref(self.ptr)
            self.ptr = None


class Group:
    """coolmic_group_t: many transform -> vumeter pipelines, one launch per block"""

    def __init__(self, channels, max_streams, block_frames, queue_blocks=2, rate=48000, device=None):
        if device is None:
            self.ptr = lib.coolmic_group_new(None, None, rate, channels, max_streams, block_frames,
                                             queue_blocks)
        else:
            self.ptr = lib.coolmic_group_new_on(device, None, None, rate, channels, max_streams, block_frames,
                                                queue_blocks)
        if not self.ptr:
            raise CoolmicError("coolmic_group_new", ERROR_GENERIC)
        self.channels = channels

    @property
    def device(self):
        return lib.coolmic_group_device(self.ptr)

    def engine(self):
        """the group's cmhip_batch_t (a borrowed pointer: cmhip_node_partial and friends)"""
        return lib.coolmic_group_engine(self.ptr)

    def add_stream(self, handle):
        return lib.coolmic_group_add_stream(self.ptr, handle.ptr if handle else None)

    def set_master_gain(self, slot, channels, scale, gains):
        arr = (C.c_uint16 * len(gains))(*gains) if gains is not None and len(gains) else None
        return lib.coolmic_group_set_master_gain(self.ptr, slot, channels, scale, arr)

    def set_channel_map(self, slot, cmap):
        if cmap is None:
            return lib.coolmic_group_set_channel_map(self.ptr, slot, None)
        m = np.asarray(cmap, dtype=np.uint8)
        return lib.coolmic_group_set_channel_map(self.ptr, slot, m.ctypes.data)

    def set_eq(self, slot, coef):
        if coef is None or len(coef) == 0:
            return lib.coolmic_group_set_eq(self.ptr, slot, 0, None)
        c = np.ascontiguousarray(coef, dtype=np.float32)
        return lib.coolmic_group_set_eq(self.ptr, slot, c.size // 5, c.ctypes.data)

    def get_iohandle(self, slot):
        return IoHandle(lib.coolmic_group_get_iohandle(self.ptr, slot))

    def pump(self):
        return lib.coolmic_group_pump(self.ptr)

    def set_pull_threads(self, threads):
        return lib.coolmic_group_set_pull_threads(self.ptr, threads)

    def vumeter_result(self, slot):
        r = VuResult()
        rc = lib.coolmic_group_vumeter_result(self.ptr, slot, C.byref(r))
        return rc, r

    def streams(self):
        return lib.coolmic_group_streams(self.ptr)

    def unref(self):
        if self.ptr:
            lib.coolmic_ro_unref(self.ptr)
            self.ptr = None


_log_keep = []


def set_log_callback(fn):
    """fn(level:int, msg:str) or None"""
    if fn is None:
        lib.coolmic_logging_set_cb_simple(LOG_FN())
        return
    cb = LOG_FN(lambda lvl, msg: fn(lvl, msg.decode(errors="replace")) or 0)
    _log_keep.append(cb)
    lib.coolmic_logging_set_cb_simple(cb)
