// cmhip_batch.hip -- the batch engine behind include/coolmic_hip.h: the object, its parameters, transfers and
// the run.  (The placement search: cmhip_place.hip; VU windows: cmhip_vu.hip; timing and ceilings:
// cmhip_measure.hip; what they share: cmhip_engine.h.)
//
// Host side of the MI355X path: owns the HBM slots of S streams, the per-stream
// parameter table and the VU windows, launches the kernels of k_block.hip / k_eq.hip / k_misc.hip on
// one HIP stream and finishes VU windows on the host in double, exactly as the
// reference does (ref: src/vumeter.c:189-218).  There is no CPU fallback: without a
// device cmhip_batch_new() fails.
#include "cmhip_engine.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include <vector>

extern "C" int coolmic_sine_period(uint_least32_t rate, int16_t *table, size_t *samples);

// ---------------------------------------------------------------------------
// errors

static thread_local char g_err[512] = "";

int cmhip_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
extern "C" const char *cmhip_last_error(void) { return g_err; }
extern "C" const char *cmhip_version(void) { return "coolmic-dsp-hip 0.1 (gfx950)"; }

extern "C" int cmhip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

extern "C" int cmhip_device_synchronize(int device)
{
    if (device < 0 || device >= cmhip_device_count())
        return fail(COOLMIC_ERROR_INVAL, "device_synchronize: no HIP device %d", device);
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipDeviceSynchronize());
    return COOLMIC_ERROR_NONE;
}

extern "C" int cmhip_device_mem_info(int device, size_t *free_bytes, size_t *total_bytes)
{
    if (device < 0 || device >= cmhip_device_count())
        return fail(COOLMIC_ERROR_INVAL, "device_mem_info: no HIP device %d", device);
    size_t f = 0, t = 0;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMemGetInfo(&f, &t));
    if (free_bytes)
        *free_bytes = f;
    if (total_bytes)
        *total_bytes = t;
    return COOLMIC_ERROR_NONE;
}

extern "C" void *cmhip_device_alloc(int device, size_t bytes)
{
    void *p = nullptr;
    if (device < 0 || device >= cmhip_device_count() || bytes == 0 || hipSetDevice(device) != hipSuccess ||
        hipMalloc(&p, bytes) != hipSuccess) {
        (void)hipGetLastError();
        fail(COOLMIC_ERROR_NOMEM, "cmhip_device_alloc: %zu bytes on device %d", bytes, device);
        return nullptr;
    }
    if (hipMemset(p, 0, bytes) != hipSuccess) {
        (void)hipFree(p);
        fail(COOLMIC_ERROR_GENERIC, "cmhip_device_alloc: clearing failed");
        return nullptr;
    }
    return p;
}

extern "C" void cmhip_device_free(int device, void *p)
{
    if (p && hipSetDevice(device) == hipSuccess)
        (void)hipFree(p);
}

extern "C" int cmhip_device_read(int device, void *dst_host, const void *src_device, size_t bytes)
{
    if (!dst_host || !src_device)
        return fail(COOLMIC_ERROR_FAULT, "device_read: NULL argument");
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(dst_host, src_device, bytes, hipMemcpyDeviceToHost));
    return COOLMIC_ERROR_NONE;
}

// ---------------------------------------------------------------------------
// the batch object

static RunTune read_tune()
{
    RunTune t{};
    if (const char *e = getenv("CMHIP_VU_TILE")) {
        const int v = atoi(e);
        if (v == 4 || v == 8 || v == 16)
            t.vu_tile = (uint32_t)v;
    }
    if (getenv("CMHIP_WIDE4_F32"))
        t.wide4_f32 = 1;
    if (const char *e = getenv("CMHIP_ROWS_RPT")) {
        const int v = atoi(e);
        if (v == 8 || v == 16 || v == 32 || v == 64)      // the tile sizes the kernels are tested with
            t.rows_rpt = (uint32_t)v;
    }
    if (const char *e = getenv("CMHIP_FAST_NW")) {
        const int v = atoi(e);
        if (v == 1 || v == 4 || v == 8)
            t.fast_nw = (uint32_t)v;
    }
    t.place_env = -1;
    if (const char *e = getenv("CMHIP_PLACE")) {
        const int v = atoi(e);
        t.place_env = v <= 0 ? 0 : (v >= 2 ? 2 : 1);
    }
    if (getenv("CMHIP_PLACE_DEBUG"))
        t.place_debug = 1;
    if (getenv("CMHIP_NO_DONE_FLAG"))
        t.no_done_flag = 1;
    t.done_spin_us = 200;
    if (const char *e = getenv("CMHIP_DONE_SPIN_US")) {
        const int v = atoi(e);
        if (v >= 0 && v <= 20000)
            t.done_spin_us = (uint32_t)v;
    }
    return t;
}

// The division constants of one gain (StreamParam): gain = mi * scale + r, mf = ceil(r * 2^32 / scale).
// mf < 2^32 because r <= scale - 1; mi <= 65535.
static void host_gain_consts(uint16_t gain, uint16_t scale, uint16_t *mi, uint32_t *mf)
{
    const uint32_t r = (uint32_t)gain % scale;
    *mi = (uint16_t)((uint32_t)gain / scale);
    *mf = (uint32_t)((((uint64_t)r << 32) + scale - 1u) / scale);
}

static void rebuild_param(cmhip_batch_t *b, unsigned int s)
{
    StreamParam &p = b->h_param[s];
    const uint16_t scale = b->h_scale[s];
    bool unity = scale != 0;                // trunc(x * g / g) == x: same as disabled
    for (unsigned c = 0; unity && c < b->d.channels; c++)
        unity = b->h_gain[(size_t)s * MAX_CH + c] == scale;
    if (scale == 0 || unity) {              // disabled: identity through the same arithmetic
        for (unsigned c = 0; c < MAX_CH; c++) {
            p.mi[c] = 1;
            p.mf[c] = 0;
        }
        p.mode = GAIN_IDENTITY;
    } else {
        bool below = true;
        for (unsigned c = 0; c < MAX_CH; c++) {
            const uint16_t g = b->h_gain[(size_t)s * MAX_CH + c];
            host_gain_consts(g, scale, &p.mi[c], &p.mf[c]);
            if (c < b->d.channels && g >= scale)
                below = false;
        }
        p.mode = below ? GAIN_BELOW_SCALE : GAIN_GENERAL;
    }
    p.mi01 = (uint32_t)p.mi[0] | ((uint32_t)p.mi[1] << 16);
    bool ident = true;
    for (unsigned c = 0; c < b->d.channels; c++)
        ident = ident && p.chmap[c] == c;
    p.map_identity = ident ? 1u : 0u;
    // stereo map as byte selector: output half h takes input half chmap[h]
    const uint32_t lo = p.chmap[0] & 1u, hi = p.chmap[1] & 1u;
    p.perm2 = (2u * lo) | ((2u * lo + 1u) << 8) | ((2u * hi) << 16) | ((2u * hi + 1u) << 24);
    b->param_dirty = true;
}

extern "C" void cmhip_batch_free(cmhip_batch_t *b)
{
    if (!b)
        return;
    (void)hipSetDevice(b->d.device);
    if (b->collecting) {
        // a collect that was begun and never ended: the helpers may still be writing into the caller's arrays
        // (wait for them), but nothing is finished into them HERE -- at free time those arrays may be gone
        // (an error path of a C host, a Python binding that collected them first)
        if (b->pool && b->d.streams >= 512)
            b->pool->finish();
        b->collecting = false;
    }
    if (b->stream)
        (void)hipStreamSynchronize(b->stream);
    for (auto &e : b->ev_used) {
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    for (auto &e : b->ev_free) {
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    if (b->h_done)
        (void)hipHostFree(b->h_done);
    if (b->d.flags & CMHIP_HOSTPCM) {
        if (b->h_out && b->h_out != b->h_in)
            (void)hipHostFree(b->h_out);
        if (b->h_in)
            (void)hipHostFree(b->h_in);
    } else {
        if (b->d_out && b->d_out != b->d_in)
            (void)hipFree(b->d_out);
        (void)hipFree(b->d_in);
    }
    (void)hipFree(b->d_f32);
    (void)hipFree(b->d_param);
    for (int i = 0; i < 3; i++)
        (void)hipFree(b->d_vu2[i]);
    delete b->pool;
    if (b->ev_main)
        (void)hipEventDestroy(b->ev_main);
    if (b->ev_node)
        (void)hipEventDestroy(b->ev_node);
    for (int i = 0; i < 4; i++)
        if (b->ev_done[i])
            (void)hipEventDestroy(b->ev_done[i]);
    for (int i = 0; i < 3; i++)
        if (b->ev_reset[i])
            (void)hipEventDestroy(b->ev_reset[i]);
    if (b->copy_stream) {
        (void)hipStreamSynchronize(b->copy_stream);
        (void)hipStreamDestroy(b->copy_stream);
    }
    (void)hipFree(b->d_nframes);
    (void)hipFree(b->d_eq);
    (void)hipFree(b->d_eqstate);
    (void)hipFree(b->d_sink);
    (void)hipFree(b->d_node_scratch);
    (void)hipFree(b->d_ring);
    if (b->h_ring)
        (void)hipHostFree(b->h_ring);
    (void)hipFree(b->d_dbg);
    for (int i = 0; i < 3; i++)
        if (b->h_pack[i])
            (void)hipHostFree(b->h_pack[i]);
    if (b->h_stage)
        (void)hipHostFree(b->h_stage);
    for (unsigned i = 0; i < STAGE_SLOTS; i++)
        if (b->stage_ev[i])
            (void)hipEventDestroy(b->stage_ev[i]);
    if (b->own_stream && b->stream)
        (void)hipStreamDestroy(b->stream);
    delete b;
}

static int batch_init(cmhip_batch_t *b)
{
    const cmhip_batch_desc_t &d = b->d;
    const size_t S = d.streams;
    HIP_TRY(hipSetDevice(d.device));
    if (d.hip_stream) {
        b->stream = (hipStream_t)d.hip_stream;
        b->own_stream = false;
    } else {
        HIP_TRY(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
        b->own_stream = true;
    }
    b->stride = (d.max_frames * d.channels + 7) / 8 * 8;
    if (d.flags & CMHIP_EQ)                // rows of whole 8-frame chunks: the EQ kernel loads chunks
        b->stride = (d.max_frames + 7) / 8 * 8 * d.channels;
    b->plane = (d.max_frames + 63) / 64 * 64;

    const size_t pcm_bytes = S * b->stride * sizeof(int16_t);
    if (d.flags & CMHIP_EXTSLOTS) {
        // no PCM arrays of its own: every run names them (cmhip_batch_run_slots)
    } else if (d.flags & CMHIP_HOSTPCM) {
        // zero copy: the kernels read and write pinned, device-mapped host memory; an upload or
        // download is a memcpy on the host (for the 1 KiB blocks of the per-stream stages the
        // two DMA submissions cost more than the block itself)
        HIP_TRY(hipHostMalloc((void **)&b->h_in, pcm_bytes, hipHostMallocMapped));
        memset(b->h_in, 0, pcm_bytes);
        HIP_TRY(hipHostGetDevicePointer((void **)&b->d_in, b->h_in, 0));
        if (!b->tune.no_done_flag) {                     // (A/B knob: CMHIP_NO_DONE_FLAG)
            HIP_TRY(hipHostMalloc((void **)&b->h_done, 64, hipHostMallocMapped));
            memset(b->h_done, 0, 64);
            HIP_TRY(hipHostGetDevicePointer((void **)&b->d_done, b->h_done, 0));
        }
        if ((d.flags & CMHIP_OUT_PCM) && !(d.flags & CMHIP_INPLACE)) {
            HIP_TRY(hipHostMalloc((void **)&b->h_out, pcm_bytes, hipHostMallocMapped));
            memset(b->h_out, 0, pcm_bytes);
            HIP_TRY(hipHostGetDevicePointer((void **)&b->d_out, b->h_out, 0));
        } else if (d.flags & CMHIP_OUT_PCM) {
            b->h_out = b->h_in;
            b->d_out = b->d_in;
        }
    } else {
        HIP_TRY(hipMalloc((void **)&b->d_in, pcm_bytes));
        HIP_TRY(hipMemsetAsync(b->d_in, 0, pcm_bytes, b->stream));
        if ((d.flags & CMHIP_OUT_PCM) && !(d.flags & CMHIP_INPLACE)) {
            HIP_TRY(hipMalloc((void **)&b->d_out, pcm_bytes));
            HIP_TRY(hipMemsetAsync(b->d_out, 0, pcm_bytes, b->stream));
        } else if (d.flags & CMHIP_OUT_PCM) {
            b->d_out = b->d_in;
        }
    }
    if (d.flags & CMHIP_OUT_F32) {
        const size_t fbytes = S * d.channels * b->plane * sizeof(float);
        HIP_TRY(hipMalloc((void **)&b->d_f32, fbytes));
        HIP_TRY(hipMemsetAsync(b->d_f32, 0, fbytes, b->stream));
    }
    HIP_TRY(hipMalloc((void **)&b->d_param, S * sizeof(StreamParam)));
    for (int i = 0; i < 3; i++) {
        HIP_TRY(hipMalloc((void **)&b->d_vu2[i], S * sizeof(VuState)));
        HIP_TRY(hipMemsetAsync(b->d_vu2[i], 0, S * sizeof(VuState), b->stream));
        HIP_TRY(hipEventCreate(&b->ev_reset[i]));       // (stamped by the pack kernel's own dispatch)
    }
    b->d_vu = b->d_vu2[0];
    {
        // The copy stream carries small work beside the main stream's long kernels (snapshots, node
        // records): at the highest priority, or its kernels wait for a slot among a quarter of a million
        // workgroups of the run (config 5: k_node_partial took 30-330 us instead of 9, and every few
        // steps the chain snapshot -> collect -> next launch left the card idle for 150 us)
        int least = 0, greatest = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_TRY(hipStreamCreateWithPriority(&b->copy_stream, hipStreamNonBlocking,
                                            getenv("CMHIP_SIDE_PRIORITY_OFF") ? least : greatest));
    }
    HIP_TRY(hipEventCreateWithFlags(&b->ev_main, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&b->ev_node, hipEventDisableTiming));
    for (int i = 0; i < 4; i++)
        HIP_TRY(hipEventCreate(&b->ev_done[i]));
    HIP_TRY(hipMalloc((void **)&b->d_nframes, S * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void **)&b->d_sink, sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(b->d_sink, 0, sizeof(unsigned long long), b->stream));
    HIP_TRY(hipMalloc((void **)&b->d_dbg, 64 * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(b->d_dbg, 0, 64 * sizeof(unsigned long long), b->stream));
    if (d.flags & CMHIP_EQ) {
        HIP_TRY(prepare_eq(d.device));
        HIP_TRY(hipMalloc((void **)&b->d_eq, S * sizeof(EqParam)));
        HIP_TRY(hipMalloc((void **)&b->d_eqstate, S * d.channels * sizeof(EqState)));   // per channel
        HIP_TRY(hipMemsetAsync(b->d_eq, 0, S * sizeof(EqParam), b->stream));
        HIP_TRY(hipMemsetAsync(b->d_eqstate, 0, S * d.channels * sizeof(EqState), b->stream));
        b->h_eq.assign(S, EqParam{});
    }
    for (int i = 0; i < 3; i++) {
        HIP_TRY(hipHostMalloc((void **)&b->h_pack[i], S * (1u + 2u * d.channels) * sizeof(unsigned long long),
                              hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer((void **)&b->d_pack[i], b->h_pack[i], 0));
    }
    HIP_TRY(hipHostMalloc((void **)&b->h_stage, STAGE_SLOTS * STAGE_BYTES, hipHostMallocDefault));
    for (unsigned i = 0; i < STAGE_SLOTS; i++)
        HIP_TRY(hipEventCreateWithFlags(&b->stage_ev[i], hipEventDisableTiming));

    b->h_param.assign(S, StreamParam{});
    b->h_scale.assign(S, 0);
    b->h_gain.assign(S * MAX_CH, 0);
    for (size_t s = 0; s < S; s++) {
        for (unsigned c = 0; c < MAX_CH; c++)
            b->h_param[s].chmap[c] = (uint8_t)(c < d.channels ? c : 0);
        rebuild_param(b, (unsigned)s);
    }
    // PCM arrays of its own, input and output apart: where they lie against each other (place_arrays_apart)
    if (b->d_out && b->d_out != b->d_in && !(d.flags & (CMHIP_HOSTPCM | CMHIP_EXTSLOTS | CMHIP_EQ))) {
        if (cmhip_engine_flush_params(b) || cmhip_engine_place_arrays_apart(b, pcm_bytes))
            return COOLMIC_ERROR_GENERIC;
    }
    HIP_TRY(hipStreamSynchronize(b->stream));
    return COOLMIC_ERROR_NONE;
}

extern "C" cmhip_batch_t *cmhip_batch_new(const cmhip_batch_desc_t *desc)
{
    if (!desc) {
        fail(COOLMIC_ERROR_FAULT, "cmhip_batch_new: desc is NULL");
        return nullptr;
    }
    if (desc->streams == 0 || desc->channels == 0 || desc->channels > MAX_CH ||
        desc->max_frames == 0 || desc->rate == 0) {
        fail(COOLMIC_ERROR_INVAL, "cmhip_batch_new: streams/channels/max_frames/rate out of range");
        return nullptr;
    }
    if ((uint64_t)desc->max_frames * desc->channels >= (1ull << 31)) {
        fail(COOLMIC_ERROR_INVAL, "cmhip_batch_new: slot larger than 2^31 samples");
        return nullptr;
    }
    if (!(desc->flags & (CMHIP_OUT_PCM | CMHIP_OUT_F32 | CMHIP_VU))) {
        fail(COOLMIC_ERROR_INVAL, "cmhip_batch_new: no output requested");
        return nullptr;
    }
    if (cmhip_device_count() <= desc->device || desc->device < 0) {
        fail(COOLMIC_ERROR_NOSYS, "cmhip_batch_new: no HIP device %d (%d visible); there is no CPU path",
             desc->device, cmhip_device_count());
        return nullptr;
    }
    cmhip_batch_t *b = new cmhip_batch();
    b->d = *desc;
    b->stream = nullptr;
    b->own_stream = false;
    b->d_in = b->d_out = nullptr;
    b->h_in = b->h_out = nullptr;
    b->in_flight = false;
    b->h_done = b->d_done = nullptr;
    b->done_seq = 0;
    b->done_flagged = false;
    b->d_f32 = nullptr;
    b->d_param = nullptr;
    b->d_vu = nullptr;
    b->d_vu2[0] = b->d_vu2[1] = b->d_vu2[2] = nullptr;
    b->cur = 0;
    b->copy_stream = nullptr;
    b->ev_main = nullptr;
    b->ev_node = nullptr;
    b->node_reading = false;
    b->ev_done[0] = b->ev_done[1] = b->ev_done[2] = b->ev_done[3] = nullptr;
    b->last_done = nullptr;
    b->done_next = 0;
    b->ev_reset[0] = b->ev_reset[1] = b->ev_reset[2] = nullptr;
    b->reset_pending[0] = b->reset_pending[1] = b->reset_pending[2] = false;
    b->pool = nullptr;
    b->d_nframes = nullptr;
    b->d_eq = nullptr;
    b->d_eqstate = nullptr;
    b->d_sink = nullptr;
    b->d_node_scratch = nullptr;
    b->d_ring = b->h_ring = nullptr;
    b->ring_slots = 0;
    b->ring_seq = 0;
    b->ring_fetched = 0;
    b->d_dbg = nullptr;
    b->h_pack[0] = b->h_pack[1] = b->h_pack[2] = nullptr;
    b->d_pack[0] = b->d_pack[1] = b->d_pack[2] = nullptr;
    b->snap_set2[0] = b->snap_set2[1] = b->snap_set2[2] = 0;
    b->collecting = false;
    b->job_out = nullptr;
    b->job_rc = nullptr;
    b->job_slot = 0;
    b->snap_head = b->snap_count = 0;
    b->h_stage = nullptr;
    for (unsigned i = 0; i < STAGE_SLOTS; i++) {
        b->stage_ev[i] = nullptr;
        b->stage_busy[i] = false;
    }
    b->stage_next = 0;
    b->parity = 0;
    b->param_dirty = true;
    b->all_identity = true;
    b->all_gain_identity = true;
    b->eq_dirty = false;
    b->nsec = 0;
    b->timing = false;
    b->timing_every = 1;
    b->timing_count = 0;
    b->tune = read_tune();
    memset(&b->place, 0, sizeof(b->place));
    b->place.chosen_out = 1;
    b->place.candidates = 2;
    b->vu_off = false;
    if (batch_init(b) != COOLMIC_ERROR_NONE) {
        cmhip_batch_free(b);
        return nullptr;
    }
    return b;
}

// diagnostic hook: the 64 stamp words a -DCMHIP_EQ_STAMPS build of the kernels writes
extern "C" int cmhip_debug_read(cmhip_batch_t *b, unsigned long long *out)
{
    if (!b || !out)
        return COOLMIC_ERROR_FAULT;
    if (hipSetDevice(b->d.device) != hipSuccess || hipStreamSynchronize(b->stream) != hipSuccess ||
        hipMemcpy(out, b->d_dbg, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess)
        return COOLMIC_ERROR_GENERIC;
    return COOLMIC_ERROR_NONE;
}

// test hook: runs launched by this process so far (tests count launches per pull with it)
static std::atomic<unsigned long long> g_runs{0};
extern "C" unsigned long long cmhip_debug_run_count(void) { return g_runs.load(); }

// test hook: the division constants of a gain (host logic, needs no GPU)
extern "C" void cmhip_test_gain_consts(uint16_t gain, uint16_t scale, uint16_t *mi, uint32_t *mf)
{
    if (scale && mi && mf)
        host_gain_consts(gain, scale, mi, mf);
}

// ---------------------------------------------------------------------------
// parameters

// ref: src/transform.c:195-222, per stream
static int set_gain_one(cmhip_batch_t *b, unsigned int s, unsigned int channels, uint16_t scale,
                        const uint16_t *gain)
{
    uint16_t *g = &b->h_gain[(size_t)s * MAX_CH];
    const unsigned int own = b->d.channels;
    if (!channels || !scale || !gain) {
        b->h_scale[s] = 0;
    } else if (channels == own) {
        for (unsigned c = 0; c < own; c++)
            g[c] = gain[c];
        b->h_scale[s] = scale;
    } else if (channels == 1) {
        for (unsigned c = 0; c < own; c++)
            g[c] = gain[0];
        b->h_scale[s] = scale;
    } else if (channels == 2 && own == 1) {
        g[0] = (uint16_t)(((uint32_t)gain[0] + (uint32_t)gain[1]) / 2u);
        b->h_scale[s] = scale;
    } else {
        return COOLMIC_ERROR_INVAL;
    }
    rebuild_param(b, s);
    return COOLMIC_ERROR_NONE;
}

extern "C" int cmhip_batch_set_gain(cmhip_batch_t *b, long stream, unsigned int channels,
                                    uint16_t scale, const uint16_t *gain)
{
    if (!b)
        return fail(COOLMIC_ERROR_FAULT, "set_gain: batch is NULL");
    if (stream >= (long)b->d.streams || stream < -1)
        return fail(COOLMIC_ERROR_INVAL, "set_gain: stream %ld out of range", stream);
    if (stream >= 0)
        return set_gain_one(b, (unsigned)stream, channels, scale, gain);
    int rc = COOLMIC_ERROR_NONE;
    for (unsigned s = 0; s < b->d.streams && rc == COOLMIC_ERROR_NONE; s++)
        rc = set_gain_one(b, s, channels, scale, gain);
    return rc;
}

extern "C" int cmhip_batch_set_chmap(cmhip_batch_t *b, long stream, const uint8_t *map)
{
    if (!b)
        return fail(COOLMIC_ERROR_FAULT, "set_chmap: batch is NULL");
    if (stream >= (long)b->d.streams || stream < -1)
        return fail(COOLMIC_ERROR_INVAL, "set_chmap: stream %ld out of range", stream);
    if (map)
        for (unsigned c = 0; c < b->d.channels; c++)
            if (map[c] >= b->d.channels)
                return fail(COOLMIC_ERROR_INVAL, "set_chmap: map[%u]=%u >= channels", c, map[c]);
    const unsigned lo = stream < 0 ? 0 : (unsigned)stream;
    const unsigned hi = stream < 0 ? b->d.streams : (unsigned)stream + 1;
    for (unsigned s = lo; s < hi; s++) {
        for (unsigned c = 0; c < b->d.channels; c++)
            b->h_param[s].chmap[c] = map ? map[c] : (uint8_t)c;
        rebuild_param(b, s);
    }
    return COOLMIC_ERROR_NONE;
}

extern "C" int cmhip_batch_set_eq(cmhip_batch_t *b, long stream, unsigned int nsec,
                                  const float *coef)
{
    if (!b)
        return fail(COOLMIC_ERROR_FAULT, "set_eq: batch is NULL");
    if (!(b->d.flags & CMHIP_EQ))
        return fail(COOLMIC_ERROR_INVAL, "set_eq: batch was created without CMHIP_EQ");
    if (nsec > MAX_EQ || (nsec && !coef))
        return fail(COOLMIC_ERROR_INVAL, "set_eq: at most %u sections", MAX_EQ);
    if (stream >= (long)b->d.streams || stream < -1)
        return fail(COOLMIC_ERROR_INVAL, "set_eq: stream %ld out of range", stream);
    if (stream >= 0 && nsec != b->nsec)
        return fail(COOLMIC_ERROR_INVAL,
                    "set_eq: the section count is a batch property (%u); set it with stream -1",
                    b->nsec);
    const unsigned lo = stream < 0 ? 0 : (unsigned)stream;
    const unsigned hi = stream < 0 ? b->d.streams : (unsigned)stream + 1;
    for (unsigned s = lo; s < hi; s++) {
        b->h_eq[s].nsec = nsec;
        for (unsigned i = 0; i < nsec; i++)
            for (unsigned j = 0; j < 5; j++)
                b->h_eq[s].coef[i][j] = coef[i * 5 + j];
    }
    b->nsec = nsec;
    b->eq_dirty = true;
    return COOLMIC_ERROR_NONE;
}

extern "C" int cmhip_batch_eq_reset(cmhip_batch_t *b, long stream)
{
    if (!b || !(b->d.flags & CMHIP_EQ))
        return fail(COOLMIC_ERROR_INVAL, "eq_reset: no EQ in this batch");
    if (stream >= (long)b->d.streams || stream < -1)
        return fail(COOLMIC_ERROR_INVAL, "eq_reset: stream %ld out of range", stream);
    if (use(b))
        return COOLMIC_ERROR_GENERIC;
    const size_t per_stream = b->d.channels * sizeof(EqState);
    if (stream < 0)
        HIP_TRY(hipMemsetAsync(b->d_eqstate, 0, b->d.streams * per_stream, b->stream));
    else
        HIP_TRY(hipMemsetAsync(b->d_eqstate + (size_t)stream * b->d.channels, 0, per_stream, b->stream));
    return COOLMIC_ERROR_NONE;
}

extern "C" void cmhip_design_biquad(int kind, double rate, double freq, double gain_db, double q,
                                    float *coef)
{
    // RBJ audio-EQ-cookbook forms, evaluated in double and rounded to float once
    const double A = pow(10., gain_db / 40.);
    const double w0 = 2. * M_PI * freq / rate;
    const double cw = cos(w0), sw = sin(w0);
    double b0, b1, b2, a0, a1, a2;
    if (kind == 1) {
        const double alpha = sw / (2. * q);
        b0 = 1. + alpha * A;
        b1 = -2. * cw;
        b2 = 1. - alpha * A;
        a0 = 1. + alpha / A;
        a1 = -2. * cw;
        a2 = 1. - alpha / A;
    } else {
        const double alpha = sw / 2. * sqrt(2.);
        const double k = 2. * sqrt(A) * alpha;
        const double ap = A + 1., am = A - 1.;
        if (kind == 0) {
            b0 = A * (ap - am * cw + k);
            b1 = 2. * A * (am - ap * cw);
            b2 = A * (ap - am * cw - k);
            a0 = ap + am * cw + k;
            a1 = -2. * (am + ap * cw);
            a2 = ap + am * cw - k;
        } else {
            b0 = A * (ap + am * cw + k);
            b1 = -2. * A * (am + ap * cw);
            b2 = A * (ap + am * cw - k);
            a0 = ap - am * cw + k;
            a1 = 2. * (am - ap * cw);
            a2 = ap - am * cw - k;
        }
    }
    coef[0] = (float)(b0 / a0);
    coef[1] = (float)(b1 / a0);
    coef[2] = (float)(b2 / a0);
    coef[3] = (float)(a1 / a0);
    coef[4] = (float)(a2 / a0);
}

// ---------------------------------------------------------------------------
// geometry, transfers, generation

extern "C" size_t cmhip_batch_stride(const cmhip_batch_t *b) { return b ? b->stride : 0; }
extern "C" size_t cmhip_batch_max_frames(const cmhip_batch_t *b) { return b ? b->d.max_frames : 0; }
extern "C" void *cmhip_batch_dev_in(cmhip_batch_t *b) { return b ? b->d_in : nullptr; }
extern "C" void *cmhip_batch_dev_out(cmhip_batch_t *b) { return b ? b->d_out : nullptr; }
extern "C" void *cmhip_batch_dev_f32(cmhip_batch_t *b) { return b ? b->d_f32 : nullptr; }
extern "C" void *cmhip_batch_hip_stream(cmhip_batch_t *b) { return b ? (void *)b->stream : nullptr; }

// CMHIP_HOSTPCM: the host may touch the slots only while no launch is using them
static int host_slots_quiet(cmhip_batch_t *b)
{
    if (b->in_flight) {
        bool done = false;
        if (b->done_flagged) {
            // the launch's own last act was to store its sequence number here (done_epilogue).  The spin is
            // bounded by what such a launch can take -- one workgroup on a block of at most a few KiB: tens of
            // microseconds, 16 384 frames through the equaliser a few hundred -- then the stream after all, which
            // sleeps instead of holding a core (a kernel that faulted never stores; a long one is not worth a core)
            const auto t_end = std::chrono::steady_clock::now() + std::chrono::microseconds(b->tune.done_spin_us);
            unsigned spins = 0;
            while (!(done = __atomic_load_n(b->h_done, __ATOMIC_ACQUIRE) == b->done_seq)) {
                if ((++spins & 63u) == 0 && std::chrono::steady_clock::now() > t_end)
                    break;
                cmhip_cpu_relax();
            }
        }
        if (!done)
            HIP_TRY(hipStreamSynchronize(b->stream));
        b->in_flight = false;
        b->done_flagged = false;
    }
    return COOLMIC_ERROR_NONE;
}

extern "C" int cmhip_batch_upload(cmhip_batch_t *b, unsigned int stream, const int16_t *pcm,
                                  size_t frames)
{
    if (!b || !pcm)
        return fail(COOLMIC_ERROR_FAULT, "upload: NULL argument");
    if (stream >= b->d.streams || frames > b->d.max_frames || !b->d_in)
        return fail(COOLMIC_ERROR_INVAL, "upload: stream or frames out of range (or a batch without slots of its own)");
    if (use(b))
        return COOLMIC_ERROR_GENERIC;
    const size_t bytes = frames * b->d.channels * sizeof(int16_t);
    int16_t *dst = b->d_in + (size_t)stream * b->stride;
    if (bytes == 0)
        return COOLMIC_ERROR_NONE;
    if (b->h_in) {
        if (host_slots_quiet(b))
            return COOLMIC_ERROR_GENERIC;
        memcpy(b->h_in + (size_t)stream * b->stride, pcm, bytes);
        return COOLMIC_ERROR_NONE;
    }
    if (bytes <= STAGE_BYTES) {
        // small blocks (the 1 KiB pulls of the per-stream stages): bounce through pinned
        // memory so the caller may reuse its buffer as soon as we return
        const unsigned slot = b->stage_next;
        b->stage_next = (slot + 1) % STAGE_SLOTS;
        if (b->stage_busy[slot])
            HIP_TRY(hipEventSynchronize(b->stage_ev[slot]));
        unsigned char *bounce = b->h_stage + (size_t)slot * STAGE_BYTES;
        memcpy(bounce, pcm, bytes);
        HIP_TRY(hipMemcpyAsync(dst, bounce, bytes, hipMemcpyHostToDevice, b->stream));
        HIP_TRY(hipEventRecord(b->stage_ev[slot], b->stream));
        b->stage_busy[slot] = true;
    } else {
        HIP_TRY(hipMemcpyAsync(dst, pcm, bytes, hipMemcpyHostToDevice, b->stream));
        HIP_TRY(hipStreamSynchronize(b->stream));
    }
    return COOLMIC_ERROR_NONE;
}

// whole-batch transfers: one copy for all slots, asynchronous on the batch's stream.  `host`
// mirrors the device layout [S][stride] (use cmhip_host_alloc for pinned memory, which is
// what makes the copy asynchronous and full speed).
extern "C" int cmhip_batch_upload_all(cmhip_batch_t *b, const int16_t *host, size_t frames)
{
    if (!b || !host)
        return fail(COOLMIC_ERROR_FAULT, "upload_all: NULL argument");
    if (frames == 0 || frames > b->d.max_frames || !b->d_in)
        return fail(COOLMIC_ERROR_INVAL, "upload_all: frames out of range (or a batch without slots of its own)");
    if (use(b))
        return COOLMIC_ERROR_GENERIC;
    const size_t span = ((size_t)(b->d.streams - 1) * b->stride + frames * b->d.channels) * sizeof(int16_t);
    if (b->h_in) {
        if (host_slots_quiet(b))
            return COOLMIC_ERROR_GENERIC;
        memcpy(b->h_in, host, span);
        return COOLMIC_ERROR_NONE;
    }
    HIP_TRY(hipMemcpyAsync(b->d_in, host, span, hipMemcpyHostToDevice, b->stream));
    return COOLMIC_ERROR_NONE;
}

extern "C" int cmhip_batch_download_all(cmhip_batch_t *b, int16_t *host, size_t frames)
{
    if (!b || !host)
        return fail(COOLMIC_ERROR_FAULT, "download_all: NULL argument");
    if (!b->d_out)
        return fail(COOLMIC_ERROR_INVAL, "download_all: batch has no PCM output");
    if (frames == 0 || frames > b->d.max_frames)
        return fail(COOLMIC_ERROR_INVAL, "download_all: frames out of range");
    if (use(b))
        return COOLMIC_ERROR_GENERIC;
    const size_t span = ((size_t)(b->d.streams - 1) * b->stride + frames * b->d.channels) * sizeof(int16_t);
    if (b->h_out) {
        if (host_slots_quiet(b))
            return COOLMIC_ERROR_GENERIC;
        memcpy(host, b->h_out, span);
        return COOLMIC_ERROR_NONE;
    }
    HIP_TRY(hipMemcpyAsync(host, b->d_out, span, hipMemcpyDeviceToHost, b->stream));
    return COOLMIC_ERROR_NONE;
}

extern "C" void *cmhip_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        fail(COOLMIC_ERROR_NOMEM, "cmhip_host_alloc: %zu bytes of pinned memory", bytes);
        return nullptr;
    }
    return p;
}

// pinned AND mapped into the device's address space: *device_ptr is what kernels (and
// cmhip_batch_run_slots) take for the memory the host reaches through the returned pointer
extern "C" void *cmhip_host_alloc_mapped(size_t bytes, void **device_ptr)
{
    void *p = nullptr;
    if (!device_ptr || hipHostMalloc(&p, bytes, hipHostMallocMapped) != hipSuccess) {
        (void)hipGetLastError();
        fail(COOLMIC_ERROR_NOMEM, "cmhip_host_alloc_mapped: %zu bytes of mapped pinned memory", bytes);
        return nullptr;
    }
    if (hipHostGetDevicePointer(device_ptr, p, 0) != hipSuccess) {
        (void)hipHostFree(p);
        fail(COOLMIC_ERROR_GENERIC, "cmhip_host_alloc_mapped: no device pointer");
        return nullptr;
    }
    return p;
}

extern "C" void *cmhip_host_alloc_mapped_on(int device, size_t bytes, void **device_ptr)
{
    if (device < 0 || device >= cmhip_device_count() || hipSetDevice(device) != hipSuccess) {
        (void)hipGetLastError();
        fail(COOLMIC_ERROR_INVAL, "cmhip_host_alloc_mapped_on: no HIP device %d", device);
        return nullptr;
    }
    return cmhip_host_alloc_mapped(bytes, device_ptr);
}

extern "C" void cmhip_host_free(void *p)
{
    if (p)
        (void)hipHostFree(p);
}

extern "C" int cmhip_batch_download(cmhip_batch_t *b, unsigned int stream, int16_t *pcm,
                                    size_t frames)
{
    if (!b || !pcm)
        return fail(COOLMIC_ERROR_FAULT, "download: NULL argument");
    if (!b->d_out)
        return fail(COOLMIC_ERROR_INVAL, "download: batch has no PCM output");
    if (stream >= b->d.streams || frames > b->d.max_frames)
        return fail(COOLMIC_ERROR_INVAL, "download: stream or frames out of range");
    if (use(b))
        return COOLMIC_ERROR_GENERIC;
    if (b->h_out) {
        if (host_slots_quiet(b))
            return COOLMIC_ERROR_GENERIC;
        memcpy(pcm, b->h_out + (size_t)stream * b->stride, frames * b->d.channels * sizeof(int16_t));
        return COOLMIC_ERROR_NONE;
    }
    HIP_TRY(hipMemcpyAsync(pcm, b->d_out + (size_t)stream * b->stride,
                           frames * b->d.channels * sizeof(int16_t), hipMemcpyDeviceToHost,
                           b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return COOLMIC_ERROR_NONE;
}

extern "C" int cmhip_batch_download_input(cmhip_batch_t *b, unsigned int stream, int16_t *pcm,
                                          size_t frames)
{
    if (!b || !pcm)
        return fail(COOLMIC_ERROR_FAULT, "download_input: NULL argument");
    if (stream >= b->d.streams || frames > b->d.max_frames || !b->d_in)
        return fail(COOLMIC_ERROR_INVAL, "download_input: stream or frames out of range (or a batch without slots of its own)");
    if (use(b))
        return COOLMIC_ERROR_GENERIC;
    if (b->h_in) {
        if (host_slots_quiet(b))
            return COOLMIC_ERROR_GENERIC;
        memcpy(pcm, b->h_in + (size_t)stream * b->stride, frames * b->d.channels * sizeof(int16_t));
        return COOLMIC_ERROR_NONE;
    }
    HIP_TRY(hipMemcpyAsync(pcm, b->d_in + (size_t)stream * b->stride,
                           frames * b->d.channels * sizeof(int16_t), hipMemcpyDeviceToHost,
                           b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return COOLMIC_ERROR_NONE;
}

extern "C" int cmhip_batch_download_f32(cmhip_batch_t *b, unsigned int stream, unsigned int channel,
                                        float *dst, size_t frames)
{
    if (!b || !dst)
        return fail(COOLMIC_ERROR_FAULT, "download_f32: NULL argument");
    if (!b->d_f32)
        return fail(COOLMIC_ERROR_INVAL, "download_f32: batch has no float output");
    if (stream >= b->d.streams || channel >= b->d.channels || frames > b->d.max_frames)
        return fail(COOLMIC_ERROR_INVAL, "download_f32: argument out of range");
    if (use(b))
        return COOLMIC_ERROR_GENERIC;
    HIP_TRY(hipMemcpyAsync(dst, b->d_f32 + ((size_t)stream * b->d.channels + channel) * b->plane,
                           frames * sizeof(float), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return COOLMIC_ERROR_NONE;
}

extern "C" int cmhip_batch_generate(cmhip_batch_t *b, int mode, uint32_t seed, size_t frames,
                                    uint64_t first_global, uint64_t global_step,
                                    uint64_t frame_offset)
{
    if (!b)
        return fail(COOLMIC_ERROR_FAULT, "generate: batch is NULL");
    if (frames > b->d.max_frames || mode < 0 || mode > 2 || !b->d_in)
        return fail(COOLMIC_ERROR_INVAL, "generate: frames or mode out of range (or a batch without slots of its own)");
    if (use(b))
        return COOLMIC_ERROR_GENERIC;
    GenArgs g;
    memset(&g, 0, sizeof(g));
    g.dst = b->d_in;
    g.streams = b->d.streams;
    g.channels = b->d.channels;
    g.frames = (uint32_t)frames;
    g.stride = b->stride;
    g.seed = seed;
    g.first_global = first_global;
    g.global_step = global_step;
    g.frame_offset = frame_offset;
    size_t n = 0;
    if (coolmic_sine_period(48000, g.sine, &n) != COOLMIC_ERROR_NONE || n != 48)
        return fail(COOLMIC_ERROR_GENERIC, "generate: sine table unavailable");
    HIP_TRY(launch_generate(g, mode, b->stream));
    b->in_flight = true;
    b->done_flagged = false;               // (whatever run went before: this kernel is behind it and carries no flag)
    return COOLMIC_ERROR_NONE;
}

// ---------------------------------------------------------------------------
// the hot path

// the main stream is about to touch the windows a node partial may still be reading on the copy stream
int cmhip_engine_settle_node(cmhip_batch_t *b)
{
    if (b->node_reading) {
        HIP_TRY(hipEventRecord(b->ev_node, b->copy_stream));
        HIP_TRY(hipStreamWaitEvent(b->stream, b->ev_node, 0));
        b->node_reading = false;
    }
    return COOLMIC_ERROR_NONE;
}

int cmhip_engine_flush_params(cmhip_batch_t *b)
{
    if (b->param_dirty) {
        b->all_identity = true;
        for (const auto &p : b->h_param)
            if (!p.map_identity)
                b->all_identity = false;
        b->all_gain_identity = true;
        for (const auto &p : b->h_param)
            if (p.mode != GAIN_IDENTITY)
                b->all_gain_identity = false;
        HIP_TRY(hipMemcpyAsync(b->d_param, b->h_param.data(), b->h_param.size() * sizeof(StreamParam),
                               hipMemcpyHostToDevice, b->stream));
        b->param_dirty = false;
    }
    if (b->eq_dirty) {
        HIP_TRY(hipMemcpyAsync(b->d_eq, b->h_eq.data(), b->h_eq.size() * sizeof(EqParam),
                               hipMemcpyHostToDevice, b->stream));
        b->eq_dirty = false;
    }
    return COOLMIC_ERROR_NONE;
}

static EventPair take_events(cmhip_batch_t *b)
{
    EventPair e{};
    if (!b->ev_free.empty()) {
        e = b->ev_free.back();
        b->ev_free.pop_back();
    } else {
        (void)hipEventCreate(&e.a);
        (void)hipEventCreate(&e.b);
    }
    return e;
}

static int batch_run(cmhip_batch_t *b, size_t frames, const uint32_t *frames_per_stream, const int16_t *slots_in,
                     int16_t *slots_out);

extern "C" int cmhip_batch_run(cmhip_batch_t *b, size_t frames, const uint32_t *frames_per_stream)
{
    if (!b)
        return fail(COOLMIC_ERROR_FAULT, "run: batch is NULL");
    if (!b->d_in)
        return fail(COOLMIC_ERROR_INVAL, "run: this batch has no slots of its own (cmhip_batch_run_slots)");
    return batch_run(b, frames, frames_per_stream, b->d_in, b->d_out);
}

// The same pass over PCM arrays the caller names for this run: device-accessible memory laid out
// like the batch's own, [S][cmhip_batch_stride()] -- e.g. pinned, device-mapped host memory from
// cmhip_host_alloc_mapped(), which the kernel then reads and writes over PCIe (coolmic_group_t
// rotates several such sets so that sources fill one and readers drain another while a third is
// on the GPU).  Parameters, VU windows and filter state are the batch's, as ever.
extern "C" int cmhip_batch_run_slots(cmhip_batch_t *b, size_t frames, const uint32_t *frames_per_stream,
                                     const void *slots_in, void *slots_out)
{
    if (!b || !slots_in)
        return fail(COOLMIC_ERROR_FAULT, "run_slots: NULL argument");
    if (((b->d.flags & CMHIP_OUT_PCM) != 0) != (slots_out != nullptr))
        return fail(COOLMIC_ERROR_INVAL, "run_slots: an output array exactly when the batch writes PCM");
    if ((b->d.flags & CMHIP_INPLACE) && slots_out != slots_in)
        return fail(COOLMIC_ERROR_INVAL, "run_slots: an in-place batch takes the same array twice");
    return batch_run(b, frames, frames_per_stream, (const int16_t *)slots_in, (int16_t *)slots_out);
}

static int batch_run(cmhip_batch_t *b, size_t frames, const uint32_t *frames_per_stream, const int16_t *slots_in,
                     int16_t *slots_out)
{
    if (frames > b->d.max_frames)
        return fail(COOLMIC_ERROR_INVAL, "run: %zu frames exceed the slot capacity %zu", frames,
                    b->d.max_frames);
    if (frames == 0)
        return COOLMIC_ERROR_NONE;
    if (use(b))
        return COOLMIC_ERROR_GENERIC;
    if (frames_per_stream) {
        for (unsigned s = 0; s < b->d.streams; s++)
            if (frames_per_stream[s] > frames)
                return fail(COOLMIC_ERROR_INVAL, "run: frames_per_stream[%u] above frames", s);
        HIP_TRY(hipMemcpyAsync(b->d_nframes, frames_per_stream, b->d.streams * sizeof(uint32_t),
                               hipMemcpyHostToDevice, b->stream));
    }
    if (cmhip_engine_flush_params(b) || cmhip_engine_settle_node(b))
        return COOLMIC_ERROR_GENERIC;

    const bool vu = (b->d.flags & CMHIP_VU) != 0 && !b->vu_off;
    // ring mode: this run's window is a cleared slot of its own (sample indices start at 0: slot 0 of
    // VuState::samples is read, slot 1 written)
    const bool ring = vu && b->ring_slots != 0;
    // (a slot is cleared when it is fetched: a run that wrapped onto an unfetched window would add to stale sums
    // and keys without anybody noticing -- the owner fetches at least once per `ring_slots` runs, transform.c)
    if (ring && b->ring_seq - b->ring_fetched >= b->ring_slots)
        return fail(COOLMIC_ERROR_BUSY, "run: the window ring is full (%u runs unfetched): cmhip_batch_vu_ring_fetch first",
                    b->ring_slots);
    VuState *const window = ring ? b->d_ring + (size_t)(b->ring_seq % b->ring_slots) * b->d.streams : b->d_vu;
    const uint32_t parity = ring ? 0u : b->parity;
    EventPair ev{};                          // timing: the events take the kernel's own start and end
    const bool timed = b->timing && b->timing_count++ % b->timing_every == 0;
    if (timed) {
        ev = take_events(b);
    } else if (vu && !b->h_in) {             // the end of this run, for the next snapshot
        // (not for slots in host memory: those batches are fed block by block and waited for, and the stop
        // event costs a launch 2 us -- tools/ubench_roundtrip.hip; a snapshot records an event of its own then)
        ev.b = b->ev_done[b->done_next];
        b->done_next = (b->done_next + 1u) & 3u;
    }
    b->last_done = vu ? ev.b : nullptr;
    // completion by flag: only where the host waits for every launch (CMHIP_HOSTPCM), every stream runs
    // its whole count, and -- the launcher decides -- the grid is one workgroup
    uint32_t *const flag = (b->d_done && !frames_per_stream) ? b->d_done : nullptr;
    const uint32_t flag_seq = ++b->done_seq;
    bool flagged = false;
    if ((b->d.flags & CMHIP_EQ) && b->nsec) {        // without sections the plain kernels do the same
        EqArgs a;
        memset(&a, 0, sizeof(a));
        a.in = slots_in;
        a.out = (b->d.flags & CMHIP_OUT_PCM) ? slots_out : nullptr;
        a.f32 = b->d_f32;
        a.param = b->d_param;
        a.eq = b->d_eq;
        a.state = b->d_eqstate;
        a.vu = vu ? window : nullptr;
        a.nframes = frames_per_stream ? b->d_nframes : nullptr;
        a.frames = (uint32_t)frames;
        a.streams = b->d.streams;
        a.channels = b->d.channels;
        a.nsec = b->nsec;
        a.whole_streams = (slots_out == slots_in && !b->all_identity) ? 1u : 0u;
        a.parity = parity;
        a.dbg = b->d_dbg;
        a.stride = b->stride;
        a.plane = b->plane;
        a.done_flag = flag;
        a.done_seq = flag_seq;
        HIP_TRY(launch_eq(a, b->stream, ev.a, ev.b, &flagged));
        b->in_flight = true;
    } else {
        RunArgs a;
        memset(&a, 0, sizeof(a));
        a.in = slots_in;
        a.out = (b->d.flags & CMHIP_OUT_PCM) ? slots_out : nullptr;
        a.f32 = b->d_f32;
        a.param = b->d_param;
            a.vu = vu ? window : nullptr;
        a.nframes = frames_per_stream ? b->d_nframes : nullptr;
        a.frames = (uint32_t)frames;
        a.streams = b->d.streams;
        a.channels = b->d.channels;
        a.stride = b->stride;
        a.plane = b->plane;
        a.chunks = 0;                      // the launcher sizes the tiles per kernel variant
        a.identity_maps = b->all_identity ? 1u : 0u;
        a.identity_gains = b->all_gain_identity ? 1u : 0u;
        a.parity = parity;
        a.done_flag = flag;
        a.done_seq = flag_seq;
        HIP_TRY(launch_run(a, b->tune, b->stream, ev.a, ev.b, &flagged));
        b->in_flight = true;
    }
    b->done_flagged = flagged;
    if (timed)
        b->ev_used.push_back(ev);
    g_runs.fetch_add(1, std::memory_order_relaxed);
    if (ring)
        b->ring_seq++;
    else if (vu)
        b->parity ^= 1u;                   // the kernel wrote the other samples slot
    return COOLMIC_ERROR_NONE;
}

extern "C" int cmhip_batch_sync(cmhip_batch_t *b)
{
    if (!b)
        return fail(COOLMIC_ERROR_FAULT, "sync: batch is NULL");
    if (use(b))
        return COOLMIC_ERROR_GENERIC;
    HIP_TRY(hipStreamSynchronize(b->stream));
    return COOLMIC_ERROR_NONE;
}

