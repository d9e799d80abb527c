// cmhip_batch.hip -- the batch engine behind include/coolmic_hip.h.
//
// Host side of the MI355X path: owns the HBM slots of S streams, the per-stream
// parameter table and the VU windows, launches the kernels of k_block.hip / k_eq.hip / k_misc.hip on
// one HIP stream and finishes VU windows on the host in double, exactly as the
// reference does (ref: src/vumeter.c:189-218).  There is no CPU fallback: without a
// device cmhip_batch_new() fails.
#include "cmhip_internal.h"

#include <coolmic-dsp/coolmic-dsp.h>
#include <coolmic_hip.h>

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

#include "work_pool.h"
#include "host_internal.h"

extern "C" int coolmic_sine_period(uint_least32_t rate, int16_t *table, size_t *samples);

using namespace cmhip;

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
#define fail cmhip_fail

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(COOLMIC_ERROR_GENERIC, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                \
    } while (0)

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

constexpr unsigned STAGE_SLOTS = 4;
constexpr size_t STAGE_BYTES = 64 * 1024;

struct EventPair {
    hipEvent_t a, b;
};

struct cmhip_batch {
    cmhip_batch_desc_t d;
    hipStream_t stream;
    bool own_stream;
    size_t stride;                 // samples between slots
    size_t plane;                  // floats between f32 planes

    int16_t *d_in, *d_out;
    int16_t *h_in, *h_out;         // CMHIP_HOSTPCM: the slots live in pinned host memory (d_* alias them)
    bool in_flight;                // a launch may still be using the slots (CMHIP_HOSTPCM)
    // CMHIP_HOSTPCM: a launch of one workgroup reports its end through a word in pinned, device-mapped host
    // memory (RunArgs::done_flag) and the host spins on it -- 4-5 us less per pull than waiting for the stream
    uint32_t *h_done, *d_done;
    uint32_t done_seq;             // the last sequence number handed to a launch
    bool done_flagged;             // ... and that launch carries the flag
    float *d_f32;
    StreamParam *d_param;
    VuState *d_vu;                         // the window runs accumulate into (= d_vu2[cur])
    VuState *d_vu2[3];                     // three sets in rotation: accumulating / being copied out / cleared
    unsigned int cur;
    hipStream_t copy_stream;               // snapshots travel here, beside the next run
    hipEvent_t ev_main, ev_reset[3];
    // A node partial (cmhip_node_partial) reads the current windows on the copy stream, beside the next
    // run; node_reading is set until the window set has rotated (snapshot) or the main stream has been
    // made to wait for the copy stream (settle_node: before anything on the main stream touches them).
    hipEvent_t ev_node;
    bool node_reading;
    // The end of the last run as its own dispatch stamped it (hipExtLaunchKernelGGL): what a snapshot
    // makes the copy stream wait for instead of an event recorded behind the kernel -- one packet less
    // on the main stream per step.  nullptr once anything else on the main stream touched the windows.
    hipEvent_t ev_done[4], last_done;
    unsigned done_next;
    bool reset_pending[3];
    struct WorkPool *pool;
    uint32_t *d_nframes;
    EqParam *d_eq;
    EqState *d_eqstate;
    unsigned long long *d_sink;
    long long *d_node_scratch;             // one node record, for cmhip_batch_vu_node_record (made on first use)
    // ring mode (cmhip_batch_vu_ring): every run accumulates into a window of its own
    VuState *d_ring, *h_ring;              // ring_slots x S windows on the device / pinned staging for a fetch
    unsigned int ring_slots;
    uint64_t ring_seq;                     // sequence number of the next run
    uint64_t ring_fetched;                 // runs below this sequence number have been fetched: their slots are clear
    unsigned long long *d_dbg;             // 64 words, written only by diagnostic builds

    std::vector<StreamParam> h_param;
    std::vector<uint16_t> h_scale;         // the reference's master_gain_scale per stream
    std::vector<uint16_t> h_gain;          // [S][16]
    bool param_dirty;
    bool all_identity;                     // no stream has a channel map (recomputed on upload)
    bool all_gain_identity;                // no stream has a gain (same)
    std::vector<EqParam> h_eq;
    unsigned int nsec;
    bool eq_dirty;

    // three snapshots may be pending (one being finished by the helper threads, one waiting, one on its way):
    // each a packed copy of a window set, [1 + 2C][S] words in pinned, device-mapped host memory that
    // k_vu_pack writes itself (h_pack / d_pack: host / device view)
    unsigned long long *h_pack[3], *d_pack[3];
    unsigned int snap_set2[3];             // which of the three window sets the snapshot closed (its event: ev_reset)
    bool collecting;                       // between cmhip_batch_vu_collect_begin and _end
    coolmic_vumeter_result_t *job_out;
    int *job_rc;
    unsigned int job_slot;
    unsigned int snap_head, snap_count;    // ring of pending snapshots (oldest = head)
    unsigned char *h_stage;                // pinned upload ring, STAGE_SLOTS x STAGE_BYTES
    hipEvent_t stage_ev[4];
    bool stage_busy[4];
    unsigned int stage_next;
    unsigned int parity;                   // current slot of VuState::samples

    bool timing;
    unsigned int timing_every, timing_count;     // every n-th run carries the events (cmhip_batch_timing)
    std::vector<EventPair> ev_used, ev_free;
    RunTune tune;                          // launcher knobs, read once at creation
    cmhip_placement_t place;               // what the placement search did (cmhip_batch_placement)
    bool vu_off;                           // runs leave the windows alone for now (cmhip_batch_vu_pause)
};

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

static inline int use(cmhip_batch_t *b)
{
    HIP_TRY(hipSetDevice(b->d.device));
    return COOLMIC_ERROR_NONE;
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

// Placement of a batch's two PCM arrays.  On MI355X a kernel that streams one large array in and another
// out runs 3-5 % faster when the two lie in different stretches of the card's memory (measured:
// tools/placement_*.py, DESIGN 4.1 -- physical memory falls into stretches of up to 32 GiB of
// three kinds; reads and writes that go to the same kind get in each other's way, and of the pairs
// of different kinds one is better than the others).  Nothing but the virtual address is visible from
// here, so the arrays are chosen by probing (place_arrays_apart, below): more candidates behind spacer
// allocations, the batch's own run on every pair of them.
// The search is the CALLER's decision (CMHIP_PLACE_SEARCH in the batch's flags): a library must not, by
// default, take tens of GiB for a moment and seconds of a constructor.  Only for arrays of 256 MiB and
// more, never more than PLACE_BUDGET_FRAC of the memory reported free, allocations stop after 0.3 s;
// what it did is in cmhip_batch_placement().
constexpr size_t PLACE_MIN_BYTES = 256ull << 20;
// Spacers before candidates 2, 3, ...: 68 GiB in all reach past two whole stretches.  (Larger ones reach further
// -- 4 ... 32 GiB, 124 in all, found the best kind of pair more often -- but allocating from memory that this or
// an earlier process has freed is slow on this driver, which hands out cleared pages: single allocations of
// 16-32 GiB were seen to take 3-6 s.)
constexpr size_t PLACE_SPACER_GIB[] = {0, 4, 8, 16, 16, 24};
constexpr int PLACE_TRIES = 6;
constexpr double PLACE_BUDGET_S = 0.3;
constexpr double PLACE_BUDGET_FRAC = 0.5;

// a probe: the batch's own run (as created: no gain, no maps), full slots, from one candidate into another
static double place_probe_ms(cmhip_batch_t *b, const void *src, void *dst, hipEvent_t e0, hipEvent_t e1)
{
    RunArgs a;
    memset(&a, 0, sizeof(a));
    a.in = (const int16_t *)src;
    a.out = (int16_t *)dst;
    a.f32 = b->d_f32;
    a.param = b->d_param;
    a.vu = (b->d.flags & CMHIP_VU) ? b->d_vu : nullptr;
    a.frames = (uint32_t)b->d.max_frames;
    a.streams = b->d.streams;
    a.channels = b->d.channels;
    a.stride = b->stride;
    a.plane = b->plane;
    a.identity_maps = 1;
    a.identity_gains = 1;
    const int n = 6;
    for (int i = 0; i < 2; i++)
        if (launch_run(a, b->tune, b->stream) != hipSuccess)
            return -1.;
    if (hipEventRecord(e0, b->stream) != hipSuccess)
        return -1.;
    for (int i = 0; i < n; i++)
        if (launch_run(a, b->tune, b->stream) != hipSuccess)
            return -1.;
    float ms = 0.f;
    if (hipEventRecord(e1, b->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
        hipEventElapsedTime(&ms, e0, e1) != hipSuccess)
        return -1.;
    b->place.probe_launches += n + 2;
    return (double)ms / n;
}

// Who searches: a batch created with CMHIP_PLACE_SEARCH, always (the caller asked).  $CMHIP_PLACE, for
// experiments: 0 nobody, 1 also the first large batch of a device in this process without the flag, 2 every
// large batch.
static bool place_search_allowed(const cmhip_batch_t *b)
{
    static std::mutex mu;
    static bool searched[64];
    if (b->tune.place_env == 0)
        return false;
    if ((b->d.flags & CMHIP_PLACE_SEARCH) || b->tune.place_env == 2)
        return true;
    if (b->tune.place_env != 1)
        return false;
    std::lock_guard<std::mutex> g(mu);
    const int d = b->d.device;
    if (d < 0 || d >= 64 || searched[d])
        return false;
    searched[d] = true;
    return true;
}

static int flush_params(cmhip_batch_t *b);

// The batch has its two PCM arrays where hipMalloc first put them (candidates 0 and 1).  More candidates
// follow behind spacers; every pair of candidates is a possible (input, output) -- nothing is in the arrays
// yet -- and the pair the batch's own run is fastest on is kept if it beats the first by more than 2 %.
static int place_arrays_apart(cmhip_batch_t *b, size_t bytes)
{
    bool probed = false;
    void *cand[PLACE_TRIES + 1] = {nullptr}, *spacer[PLACE_TRIES + 1] = {nullptr};
    cand[0] = b->d_in;
    cand[1] = b->d_out;
    int n = 2, in = 0, out = 1;
    size_t free_b = 0, total_b = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    cmhip_placement_t &rec = b->place;
    rec.chosen_in = 0;
    rec.chosen_out = 1;
    rec.candidates = 2;
    if (bytes >= PLACE_MIN_BYTES && place_search_allowed(b) && hipMemGetInfo(&free_b, &total_b) == hipSuccess &&
        hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
        probed = true;
        rec.searched = 1;
        rec.bytes_free_before = free_b;
        // never more than a stated share of what the card reports free, spacers and candidates together:
        // on a fuller card the search reaches less far (fewer candidates), it does not crowd a neighbour out
        const size_t budget = (size_t)((double)free_b * PLACE_BUDGET_FRAC);
        size_t asked = 0;
        // (allocations of this size are normally a few milliseconds; from memory that has been used and freed
        // the driver has been seen to take seconds: then what there is by then decides)
        const auto t_begin = std::chrono::steady_clock::now();
        auto elapsed = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); };
        for (int k = 1; k < PLACE_TRIES && elapsed() <= PLACE_BUDGET_S; k++) {
            const size_t sp = PLACE_SPACER_GIB[k] << 30;
            if (asked + sp + bytes > budget)
                break;
            if (hipMalloc(&spacer[n], sp) != hipSuccess || hipMalloc(&cand[n], bytes) != hipSuccess) {
                (void)hipGetLastError();                  // no room after all
                if (spacer[n])
                    asked += sp;
                break;
            }
            asked += sp + bytes;
            n++;
        }
        rec.bytes_requested = asked;
        rec.candidates = n;
        // Samples that are not zero: a tile of silence adds nothing to its window and skips its atomics, and
        // without them the kinds of pairs lie closer together (3.5 % instead of 5 %: probes on cleared arrays
        // took a pair of one kind for a good one).
        for (int k = 0; k < n; k++)
            if (hipMemsetAsync(cand[k], 0x5a, bytes, b->stream) != hipSuccess)
                break;
        // (the card may come from idle: the probes compare like with like only at settled clocks)
        for (int i = 0; i < 12; i++)
            if (place_probe_ms(b, cand[0], cand[1], e0, e1) < 0.)
                break;
        // (only the best kind of pair is worth taking: 3-7 % faster than a pair of one kind; differences of
        // 1-2 % between pairs do not last)
        double refs[PLACE_TRIES + 1], tbest = 0.;
        int nref = 0, bi = 0, bj = 1;
        for (int i = 0; i < n; i++) {
            const double ref = place_probe_ms(b, cand[0], cand[1], e0, e1);     // (again per row: clocks drift)
            if (ref > 0.)
                refs[nref++] = ref;
            for (int j = i + 1; j < n && ref > 0.; j++) {
                if (i == 0 && j == 1)
                    continue;
                const double t = place_probe_ms(b, cand[i], cand[j], e0, e1);
                if (b->tune.place_debug)
                    fprintf(stderr, "cmhip place: %d -> %d: %.4f ms (0 -> 1: %.4f ms)\n", i, j, t, ref);
                if (t > 0. && (tbest == 0. || t < tbest)) {
                    tbest = t;
                    bi = i;
                    bj = j;
                }
            }
        }
        // the first pair's time: the median of its samples (they scatter by 1 %); the fastest other pair is
        // taken if it is 2 % faster than that (a pair of the best kind is 5-8 % faster than one of one kind, a
        // middling one 3 %; moving the arrays for nothing costs nothing)
        std::sort(refs, refs + nref);
        if (nref)
            rec.first_pair_ms = refs[nref / 2];
        rec.best_pair_ms = tbest;
        if (nref && tbest > 0. && tbest < 0.98 * refs[nref / 2]) {
            in = bi;
            out = bj;
        }
        rec.chosen_in = in;
        rec.chosen_out = out;
        rec.search_ms = 1e3 * elapsed();
        if (b->tune.place_debug)
            fprintf(stderr, "cmhip place: input = candidate %d, output = candidate %d, %.0f ms, %.1f GiB asked of %.1f free\n",
                    in, out, rec.search_ms, (double)asked / (1ull << 30), (double)free_b / (1ull << 30));
    }
    if (e0)
        (void)hipEventDestroy(e0);
    if (e1)
        (void)hipEventDestroy(e1);
    for (int k = 0; k <= PLACE_TRIES; k++) {              // (all of them: a spacer may be there without its candidate)
        if (spacer[k])
            (void)hipFree(spacer[k]);
        if (cand[k] && k != in && k != out)
            (void)hipFree(cand[k]);
    }
    if (!probed)
        return COOLMIC_ERROR_NONE;
    // the probes ran the batch's kernel: whatever they left in the arrays, the windows and the float planes goes
    b->d_in = (int16_t *)cand[in];
    b->d_out = (int16_t *)cand[out];
    HIP_TRY(hipMemsetAsync(b->d_in, 0, bytes, b->stream));
    HIP_TRY(hipMemsetAsync(b->d_out, 0, bytes, b->stream));
    if (b->d.flags & CMHIP_VU)
        for (int i = 0; i < 3; i++)
            HIP_TRY(hipMemsetAsync(b->d_vu2[i], 0, b->d.streams * sizeof(VuState), b->stream));
    if (b->d_f32)
        HIP_TRY(hipMemsetAsync(b->d_f32, 0, b->d.streams * b->d.channels * b->plane * sizeof(float), b->stream));
    return COOLMIC_ERROR_NONE;
}

extern "C" int cmhip_batch_placement(const cmhip_batch_t *b, cmhip_placement_t *out)
{
    if (!b || !out)
        return fail(COOLMIC_ERROR_FAULT, "placement: NULL argument");
    *out = b->place;
    return COOLMIC_ERROR_NONE;
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
        if (flush_params(b) || place_arrays_apart(b, pcm_bytes))
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
static int settle_node(cmhip_batch_t *b)
{
    if (b->node_reading) {
        HIP_TRY(hipEventRecord(b->ev_node, b->copy_stream));
        HIP_TRY(hipStreamWaitEvent(b->stream, b->ev_node, 0));
        b->node_reading = false;
    }
    return COOLMIC_ERROR_NONE;
}

static int flush_params(cmhip_batch_t *b)
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
    if (flush_params(b) || settle_node(b))
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

// ---------------------------------------------------------------------------
// VU windows

static int16_t key_peak(unsigned long long key)
{
    const int mag = (int)(key >> KEY_ABS_SHIFT);
    return (int16_t)((key & 1ull) ? -mag : mag);
}

// ref: src/vumeter.c:203-205 -- integer mean first, then dB in double, capped at 0
static double power_db(unsigned long long sum, unsigned long long count)
{
    double p = (double)(sum / count);
    p = 20. * log10(sqrt(p) / 32768.);
    return fmin(p, 0.);
}

static int finish_window(const cmhip_batch_t *b, const VuState &v, unsigned parity,
                         coolmic_vumeter_result_t *out)
{
    const unsigned C = b->d.channels;
    const unsigned long long frames = v.samples[parity] / C;
    if (frames == 0)
        return COOLMIC_ERROR_INVAL;                      // ref: src/vumeter.c:198-199
    memset(out, 0, sizeof(*out));
    out->rate = b->d.rate;
    out->channels = C;
    out->frames = (size_t)frames;
    unsigned long long all = 0, best = 0;
    for (unsigned c = 0; c < C; c++) {
        all += v.power[c];
        out->channel_power[c] = power_db(v.power[c], frames);
        out->channel_peak[c] = key_peak(v.key[c]);
        if (v.key[c] > best)
            best = v.key[c];
    }
    out->global_power = power_db(all, frames * C);
    out->global_peak = key_peak(best);       // first max-|x| over all channels
    return COOLMIC_ERROR_NONE;
}

// the same from a packed snapshot ([word][stream]: samples, C sums, C keys)
static int finish_packed(const cmhip_batch_t *b, const unsigned long long *pack, unsigned s,
                         coolmic_vumeter_result_t *out)
{
    const unsigned C = b->d.channels;
    const size_t S = b->d.streams;
    const unsigned long long frames = pack[s] / C;
    if (frames == 0)
        return COOLMIC_ERROR_INVAL;                      // ref: src/vumeter.c:198-199
    memset(out, 0, sizeof(*out));
    out->rate = b->d.rate;
    out->channels = C;
    out->frames = (size_t)frames;
    unsigned long long all = 0, best = 0;
    for (unsigned c = 0; c < C; c++) {
        const unsigned long long power = pack[(size_t)(1u + c) * S + s], key = pack[(size_t)(1u + C + c) * S + s];
        all += power;
        out->channel_power[c] = power_db(power, frames);
        out->channel_peak[c] = key_peak(key);
        if (key > best)
            best = key;
    }
    out->global_power = power_db(all, frames * C);
    out->global_peak = key_peak(best);
    return COOLMIC_ERROR_NONE;
}

extern "C" int cmhip_batch_vu_result(cmhip_batch_t *b, unsigned int stream,
                                     coolmic_vumeter_result_t *out)
{
    if (!b || !out)
        return fail(COOLMIC_ERROR_FAULT, "vu_result: NULL argument");
    if (stream >= b->d.streams || !(b->d.flags & CMHIP_VU))
        return fail(COOLMIC_ERROR_INVAL, "vu_result: stream out of range or batch without VU");
    if (use(b))
        return COOLMIC_ERROR_GENERIC;
    b->last_done = nullptr;                  // main-stream work on the windows follows the last run
    if (settle_node(b))
        return COOLMIC_ERROR_GENERIC;
    VuState v;
    HIP_TRY(hipMemcpyAsync(&v, b->d_vu + stream, sizeof(v), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    const int rc = finish_window(b, v, b->parity, out);
    if (rc == COOLMIC_ERROR_NONE)
        HIP_TRY(hipMemsetAsync(b->d_vu + stream, 0, sizeof(VuState), b->stream));
    return rc;
}

extern "C" int cmhip_batch_vu_snapshot(cmhip_batch_t *b)
{
    if (!b)
        return fail(COOLMIC_ERROR_FAULT, "vu_snapshot: batch is NULL");
    if (!(b->d.flags & CMHIP_VU))
        return fail(COOLMIC_ERROR_INVAL, "vu_snapshot: batch without VU");
    if (b->snap_count == 3)
        return fail(COOLMIC_ERROR_BUSY, "vu_snapshot: three snapshots are waiting to be collected");
    if (use(b))
        return COOLMIC_ERROR_GENERIC;
    // The closed windows travel to the host on the copy stream and are cleared there, while
    // the main stream goes straight on with the next block into the next set.  Three sets
    // rotate so that the set a launch switches to was cleared a whole launch earlier: with
    // two, every launch waited for the copy + clear that ran beside its predecessor.
    // One kernel does both (k_vu_pack): it writes what the host needs of every window -- 1 + 2C words,
    // 40 bytes for stereo instead of the 264 of a VuState -- straight into pinned host memory and
    // clears the set; its own dispatch stamps the set's event.
    const unsigned i = b->cur;
    const unsigned slot = (b->snap_head + b->snap_count) % 3u;
    if (b->last_done) {
        HIP_TRY(hipStreamWaitEvent(b->copy_stream, b->last_done, 0));
        b->last_done = nullptr;
    } else {
        HIP_TRY(hipEventRecord(b->ev_main, b->stream));
        HIP_TRY(hipStreamWaitEvent(b->copy_stream, b->ev_main, 0));
    }
    HIP_TRY(launch_vu_pack(b->d_vu2[i], b->d.streams, b->d.channels, b->parity, b->d_pack[slot], b->copy_stream,
                           b->ev_reset[i]));
    b->snap_set2[slot] = i;
    b->reset_pending[i] = true;
    b->cur = (i + 1u) % 3u;
    b->d_vu = b->d_vu2[b->cur];
    b->node_reading = false;                 // (a node partial of the closed set runs ahead of this copy, same stream)
    if (b->reset_pending[b->cur]) {          // the set we switch to must have been cleared
        // (it was, a launch ago, in the steady state: then the main stream needs no packet for it)
        if (hipEventQuery(b->ev_reset[b->cur]) != hipSuccess)
            HIP_TRY(hipStreamWaitEvent(b->stream, b->ev_reset[b->cur], 0));
        b->reset_pending[b->cur] = false;
    }
    b->snap_count++;
    return COOLMIC_ERROR_NONE;
}

// Helpers beside the calling thread: half the hardware threads, at most 12 -- and, inside a container, no more
// than its CPU-time quota leaves beside the launching thread and the runtime's own (measured under a quota of
// 16 CPUs with 256 hardware threads visible: 12 helpers finish 4096 windows in 25 us, 14 take the CPU from the
// thread that launches and the step gets longer, NOTES_r03).  $CMHIP_POOL_THREADS overrides.
static unsigned pool_threads()
{
    if (const char *e = getenv("CMHIP_POOL_THREADS"))
        if (atoi(e) > 0)
            return (unsigned)atoi(e);
    unsigned n = std::thread::hardware_concurrency() / 2;
    n = n < 1 ? 1 : (n > 12 ? 12 : n);
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {          // cgroup v2: "<quota> <period>" or "max <period>"
        char q[32] = "";
        long period = 0;
        if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
            const long cpus = atol(q) / period;
            if (cpus >= 1 && (unsigned)cpus < n + 4u)
                n = cpus > 4 ? (unsigned)(cpus - 4) : 1u;
        }
        fclose(f);
    }
    return n;
}

static void collect_body(void *p, unsigned lo, unsigned hi)
{
    cmhip_batch_t *b = (cmhip_batch_t *)p;
    for (unsigned s = lo; s < hi; s++) {
        const int r = finish_packed(b, b->h_pack[b->job_slot], s, &b->job_out[s]);
        if (b->job_rc)
            b->job_rc[s] = r;
    }
}

// The dB finish of the oldest snapshot, in two halves: begin() waits for the snapshot's data and hands the
// windows to the helper pool, end() takes what is left itself and returns when out[] / rc[] are complete.
// Between the two the caller queues the next block -- with a window per block of a few thousand frames the
// host's finish (thousands of log10 per step) is as long as the kernel, and only beside the next launch does
// it stop counting.  cmhip_batch_vu_collect() is the two in one.
extern "C" int cmhip_batch_vu_collect_begin(cmhip_batch_t *b, coolmic_vumeter_result_t *out, int *rc)
{
    if (!b || !out)
        return fail(COOLMIC_ERROR_FAULT, "vu_collect: NULL argument");
    if (b->collecting)
        return fail(COOLMIC_ERROR_BUSY, "vu_collect_begin: the collect before has not been ended");
    if (b->snap_count == 0)
        return fail(COOLMIC_ERROR_INVAL, "vu_collect: no snapshot pending");
    if (use(b))
        return COOLMIC_ERROR_GENERIC;
    const unsigned slot = b->snap_head;
    HIP_TRY(hipEventSynchronize(b->ev_reset[b->snap_set2[slot]]));
    b->job_out = out;
    b->job_rc = rc;
    b->job_slot = slot;
    b->collecting = true;                    // (the snapshot keeps its place in the ring until end())
    if (b->d.streams >= 512) {
        if (!b->pool) {
            // (helpers beside the calling thread; $CMHIP_POOL_THREADS for hosts with a CPU quota below their
            // core count)
            b->pool = new WorkPool(pool_threads());
        }
        b->pool->start(collect_body, b, b->d.streams);
    }
    return COOLMIC_ERROR_NONE;
}

extern "C" int cmhip_batch_vu_collect_end(cmhip_batch_t *b)
{
    if (!b)
        return fail(COOLMIC_ERROR_FAULT, "vu_collect_end: batch is NULL");
    if (!b->collecting)
        return fail(COOLMIC_ERROR_INVAL, "vu_collect_end: no collect under way");
    if (b->d.streams >= 512)
        b->pool->finish();
    else
        collect_body(b, 0, b->d.streams);
    b->collecting = false;
    b->snap_head = (b->snap_head + 1u) % 3u;
    b->snap_count--;
    return COOLMIC_ERROR_NONE;
}

extern "C" int cmhip_batch_vu_collect(cmhip_batch_t *b, coolmic_vumeter_result_t *out, int *rc)
{
    const int r = cmhip_batch_vu_collect_begin(b, out, rc);
    return r != COOLMIC_ERROR_NONE ? r : cmhip_batch_vu_collect_end(b);
}

extern "C" int cmhip_batch_vu_results(cmhip_batch_t *b, coolmic_vumeter_result_t *out, int *rc)
{
    const int r = cmhip_batch_vu_snapshot(b);
    if (r != COOLMIC_ERROR_NONE)
        return r;
    // NB: unlike the per-stream call this resets every window, also those with no frames
    return cmhip_batch_vu_collect(b, out, rc);
}

extern "C" int cmhip_batch_vu_reset(cmhip_batch_t *b, long stream)
{
    if (!b)
        return fail(COOLMIC_ERROR_FAULT, "vu_reset: batch is NULL");
    if (stream >= (long)b->d.streams || stream < -1)
        return fail(COOLMIC_ERROR_INVAL, "vu_reset: stream %ld out of range", stream);
    if (use(b))
        return COOLMIC_ERROR_GENERIC;
    b->last_done = nullptr;                  // main-stream work on the windows follows the last run
    if (settle_node(b))
        return COOLMIC_ERROR_GENERIC;
    if (stream < 0)
        HIP_TRY(hipMemsetAsync(b->d_vu, 0, b->d.streams * sizeof(VuState), b->stream));
    else
        HIP_TRY(hipMemsetAsync(b->d_vu + stream, 0, sizeof(VuState), b->stream));
    return COOLMIC_ERROR_NONE;
}

extern "C" int cmhip_batch_vu_raw(cmhip_batch_t *b, unsigned int stream, int64_t *power,
                                  int16_t *peak, uint64_t *frames)
{
    if (!b)
        return fail(COOLMIC_ERROR_FAULT, "vu_raw: batch is NULL");
    if (stream >= b->d.streams)
        return fail(COOLMIC_ERROR_INVAL, "vu_raw: stream out of range");
    if (use(b))
        return COOLMIC_ERROR_GENERIC;
    if (settle_node(b))
        return COOLMIC_ERROR_GENERIC;
    VuState v;
    HIP_TRY(hipMemcpyAsync(&v, b->d_vu + stream, sizeof(v), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    for (unsigned c = 0; c < MAX_CH; c++) {
        if (power)
            power[c] = (int64_t)v.power[c];
        if (peak)
            peak[c] = key_peak(v.key[c]);
    }
    if (frames)
        *frames = v.samples[b->parity] / b->d.channels;
    return COOLMIC_ERROR_NONE;
}

// ---------------------------------------------------------------------------
// per-launch window records (engine internal, host_internal.h): transform.c / vumeter.c

static void raw_from_state(const VuState &v, unsigned parity, cmhip_vu_raw_t *out)
{
    static_assert(MAX_CH == 16, "cmhip_vu_raw_t holds sixteen channels");
    for (unsigned c = 0; c < MAX_CH; c++) {
        out->power[c] = v.power[c];
        out->key[c] = v.key[c];
    }
    out->samples = v.samples[parity];
}

extern "C" CMHIP_INTERNAL int cmhip_batch_vu_ring(cmhip_batch_t *b, unsigned int slots)
{
    if (!b)
        return fail(COOLMIC_ERROR_FAULT, "vu_ring: batch is NULL");
    if (!(b->d.flags & CMHIP_VU) || slots > 65536)
        return fail(COOLMIC_ERROR_INVAL, "vu_ring: batch without VU, or too many slots");
    if (use(b))
        return COOLMIC_ERROR_GENERIC;
    HIP_TRY(hipStreamSynchronize(b->stream));
    if (slots != b->ring_slots) {
        (void)hipFree(b->d_ring);
        if (b->h_ring)
            (void)hipHostFree(b->h_ring);
        b->d_ring = b->h_ring = nullptr;
        b->ring_slots = 0;
        if (slots) {
            const size_t bytes = (size_t)slots * b->d.streams * sizeof(VuState);
            HIP_TRY(hipMalloc((void **)&b->d_ring, bytes));
            HIP_TRY(hipHostMalloc((void **)&b->h_ring, bytes, hipHostMallocDefault));
            b->ring_slots = slots;
        }
    }
    if (b->ring_slots)
        HIP_TRY(hipMemsetAsync(b->d_ring, 0, (size_t)b->ring_slots * b->d.streams * sizeof(VuState), b->stream));
    b->ring_fetched = b->ring_seq;           // every slot is clear
    return COOLMIC_ERROR_NONE;
}

extern "C" CMHIP_INTERNAL uint64_t cmhip_batch_vu_ring_seq(const cmhip_batch_t *b) { return b ? b->ring_seq : 0; }

extern "C" CMHIP_INTERNAL int cmhip_batch_vu_ring_fetch(cmhip_batch_t *b, uint64_t first_seq, unsigned int count,
                                                        cmhip_vu_raw_t *out)
{
    if (!b || !out)
        return fail(COOLMIC_ERROR_FAULT, "vu_ring_fetch: NULL argument");
    if (!b->ring_slots || count == 0 || count > b->ring_slots || first_seq + count > b->ring_seq ||
        b->ring_seq - first_seq > b->ring_slots || first_seq != b->ring_fetched)
        return fail(COOLMIC_ERROR_INVAL, "vu_ring_fetch: runs %llu..+%u are not the oldest unfetched ones of the ring (%llu)",
                    (unsigned long long)first_seq, count, (unsigned long long)b->ring_fetched);
    if (use(b))
        return COOLMIC_ERROR_GENERIC;
    const size_t S = b->d.streams;
    const unsigned first = (unsigned)(first_seq % b->ring_slots);
    const unsigned n1 = count < b->ring_slots - first ? count : b->ring_slots - first;     // up to the wrap
    HIP_TRY(hipMemcpyAsync(b->h_ring + (size_t)first * S, b->d_ring + (size_t)first * S, n1 * S * sizeof(VuState),
                           hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipMemsetAsync(b->d_ring + (size_t)first * S, 0, n1 * S * sizeof(VuState), b->stream));
    if (n1 < count) {
        HIP_TRY(hipMemcpyAsync(b->h_ring, b->d_ring, (count - n1) * S * sizeof(VuState), hipMemcpyDeviceToHost, b->stream));
        HIP_TRY(hipMemsetAsync(b->d_ring, 0, (count - n1) * S * sizeof(VuState), b->stream));
    }
    HIP_TRY(hipStreamSynchronize(b->stream));
    for (unsigned i = 0; i < count; i++)
        raw_from_state(b->h_ring[(size_t)((first + i) % b->ring_slots) * S], 1u, &out[i]);
    b->ring_fetched = first_seq + count;
    return COOLMIC_ERROR_NONE;
}

extern "C" CMHIP_INTERNAL int cmhip_batch_vu_raw_state(cmhip_batch_t *b, unsigned int stream, cmhip_vu_raw_t *out)
{
    if (!b || !out)
        return fail(COOLMIC_ERROR_FAULT, "vu_raw_state: NULL argument");
    if (stream >= b->d.streams || !(b->d.flags & CMHIP_VU))
        return fail(COOLMIC_ERROR_INVAL, "vu_raw_state: stream out of range or batch without VU");
    if (use(b) || settle_node(b))
        return COOLMIC_ERROR_GENERIC;
    VuState v;
    HIP_TRY(hipMemcpyAsync(&v, b->d_vu + stream, sizeof(v), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    raw_from_state(v, b->parity, out);
    return COOLMIC_ERROR_NONE;
}

extern "C" CMHIP_INTERNAL void cmhip_vu_raw_merge(cmhip_vu_raw_t *acc, const cmhip_vu_raw_t *piece, unsigned int channels)
{
    for (unsigned c = 0; c < channels && c < MAX_CH; c++) {
        acc->power[c] += piece->power[c];
        uint64_t k = piece->key[c];
        if (k) {                             // the piece's sample indices continue the window's
            const uint64_t idx = (~(k >> 1) & KEY_IDX_MASK) + acc->samples;
            k = (k & ~(KEY_IDX_MASK << 1)) | ((~idx & KEY_IDX_MASK) << 1);
            if (k > acc->key[c])
                acc->key[c] = k;
        }
    }
    acc->samples += piece->samples;
}

extern "C" CMHIP_INTERNAL int cmhip_vu_raw_finish(const cmhip_vu_raw_t *w, unsigned int channels, unsigned int rate,
                                                  coolmic_vumeter_result_t *out)
{
    if (!w || !out || channels == 0 || channels > MAX_CH)
        return COOLMIC_ERROR_FAULT;
    const unsigned long long frames = w->samples / channels;
    if (frames == 0)
        return COOLMIC_ERROR_INVAL;                      // ref: src/vumeter.c:198-199
    memset(out, 0, sizeof(*out));
    out->rate = rate;
    out->channels = channels;
    out->frames = (size_t)frames;
    unsigned long long all = 0, best = 0;
    for (unsigned c = 0; c < channels; c++) {
        all += w->power[c];
        out->channel_power[c] = power_db(w->power[c], frames);
        out->channel_peak[c] = key_peak(w->key[c]);
        if (w->key[c] > best)
            best = w->key[c];
    }
    out->global_power = power_db(all, frames * channels);
    out->global_peak = key_peak(best);
    return COOLMIC_ERROR_NONE;
}

// test hook (host logic, needs no GPU): `count` raw windows of 33 words each (16 sums, 16 keys, samples), one after
// the other in stream order, merged as a meter behind a tee merges the records of the launches it has consumed
// (csrc/vumeter.c), and finished
extern "C" int cmhip_test_merge_windows(const uint64_t *windows, unsigned int count, unsigned int channels,
                                        unsigned int rate, coolmic_vumeter_result_t *out)
{
    if (!windows || !out)
        return COOLMIC_ERROR_FAULT;
    cmhip_vu_raw_t acc;
    memset(&acc, 0, sizeof(acc));
    for (unsigned int i = 0; i < count; i++) {
        cmhip_vu_raw_t w;
        memcpy(&w, windows + (size_t)i * 33u, sizeof(w));
        cmhip_vu_raw_merge(&acc, &w, channels);
    }
    return cmhip_vu_raw_finish(&acc, channels, rate, out);
}

// ---------------------------------------------------------------------------
// node-global VU

extern "C" int cmhip_batch_vu_node_partial(cmhip_batch_t *b, void *dst_device,
                                           uint64_t first_global, uint64_t global_step)
{
    if (!b || !dst_device)
        return fail(COOLMIC_ERROR_FAULT, "vu_node_partial: NULL argument");
    if (!(b->d.flags & CMHIP_VU))
        return fail(COOLMIC_ERROR_INVAL, "vu_node_partial: batch without VU");
    if (use(b))
        return COOLMIC_ERROR_GENERIC;
    long long *dst = (long long *)dst_device;
    return cmhip_batch_node_partial_split(b, dst, dst + CMHIP_NODE_SUM_WORDS, first_global, global_step, 1);
}

extern "C" int cmhip_batch_vu_node_record(cmhip_batch_t *b, int64_t *words_host, uint64_t first_global,
                                          uint64_t global_step)
{
    if (!b || !words_host)
        return fail(COOLMIC_ERROR_FAULT, "vu_node_record: NULL argument");
    if (!(b->d.flags & CMHIP_VU))
        return fail(COOLMIC_ERROR_INVAL, "vu_node_record: batch without VU");
    if (use(b))
        return COOLMIC_ERROR_GENERIC;
    if (!b->d_node_scratch)
        HIP_TRY(hipMalloc((void **)&b->d_node_scratch, CMHIP_NODE_WORDS * sizeof(long long)));
    const int rc = cmhip_batch_node_partial_split(b, b->d_node_scratch, b->d_node_scratch + CMHIP_NODE_SUM_WORDS,
                                                  first_global, global_step, 1);
    if (rc != COOLMIC_ERROR_NONE)
        return rc;
    HIP_TRY(hipMemcpyAsync(words_host, b->d_node_scratch, CMHIP_NODE_WORDS * sizeof(long long),
                           hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return COOLMIC_ERROR_NONE;
}

// internal (node.hip): the same record with its sums and its keys in two places
int cmhip_batch_node_partial_split(cmhip_batch_t *b, long long *dst_sum, long long *dst_key,
                                   uint64_t first_global, uint64_t global_step, int clear)
{
    if (use(b))
        return COOLMIC_ERROR_GENERIC;
    // This kernel reads the windows after the last run, on the same stream: its own end, stamped by
    // its dispatch, is what the next snapshot has to wait for -- no event packet on the main stream.
    if (settle_node(b))
        return COOLMIC_ERROR_GENERIC;
    hipEvent_t done = b->ev_done[b->done_next];
    b->done_next = (b->done_next + 1u) & 3u;
    b->last_done = nullptr;
    HIP_TRY(launch_node_partial(b->d_vu, b->d.streams, b->d.channels, b->parity, first_global, global_step,
                                dst_sum, dst_key, clear != 0, b->stream, done));
    b->last_done = done;
    return COOLMIC_ERROR_NONE;
}

// internal (node.hip): the record of a cmhip_node_t set, built on the COPY stream -- beside the batch's
// next run instead of between two runs (the main stream carries nothing for it: 8 us per block of
// config 5).  The copy stream waits for the last run's own end; a snapshot that follows is behind the
// kernel on the same stream; anything else that touches the windows goes through settle_node().
int cmhip_batch_node_partial_side(cmhip_batch_t *b, long long *dst_sum, long long *dst_key,
                                  uint64_t first_global, uint64_t global_step)
{
    if (use(b))
        return COOLMIC_ERROR_GENERIC;
    if (b->last_done) {
        HIP_TRY(hipStreamWaitEvent(b->copy_stream, b->last_done, 0));
    } else {
        HIP_TRY(hipEventRecord(b->ev_main, b->stream));
        HIP_TRY(hipStreamWaitEvent(b->copy_stream, b->ev_main, 0));
    }
    HIP_TRY(launch_node_partial(b->d_vu, b->d.streams, b->d.channels, b->parity, first_global, global_step,
                                dst_sum, dst_key, false, b->copy_stream, nullptr));
    b->node_reading = true;
    return COOLMIC_ERROR_NONE;
}

void *cmhip_batch_side_stream(cmhip_batch_t *b) { return (void *)b->copy_stream; }

// internal (transform.c): a batch with windows runs without touching them while paused -- the
// transform accumulates only the blocks its fused meter has asked for
extern "C" __attribute__((visibility("hidden"))) void cmhip_batch_vu_pause(cmhip_batch_t *b, int paused)
{
    if (b)
        b->vu_off = paused != 0;
}

int cmhip_batch_device(const cmhip_batch_t *b) { return b->d.device; }
unsigned int cmhip_batch_flags(const cmhip_batch_t *b) { return b->d.flags; }

extern "C" int cmhip_node_finish(const int64_t *w, unsigned int channels, unsigned int rate,
                                 coolmic_vumeter_result_t *out)
{
    if (!w || !out)
        return fail(COOLMIC_ERROR_FAULT, "node_finish: NULL argument");
    if (channels == 0 || channels > MAX_CH)
        return fail(COOLMIC_ERROR_INVAL, "node_finish: channels out of range");
    const unsigned long long frames = (unsigned long long)w[MAX_CH];
    if (frames == 0)
        return COOLMIC_ERROR_INVAL;
    memset(out, 0, sizeof(*out));
    out->rate = rate;
    out->channels = channels;
    out->frames = (size_t)frames;
    unsigned long long all = 0;
    for (unsigned c = 0; c < channels; c++) {
        const unsigned long long k = (unsigned long long)w[MAX_CH + 1 + c];
        all += (unsigned long long)w[c];
        out->channel_power[c] = power_db((unsigned long long)w[c], frames);
        const int mag = (int)(k >> 46);
        out->channel_peak[c] = (int16_t)((k & 1ull) ? -mag : mag);
    }
    const unsigned long long g = (unsigned long long)w[2 * MAX_CH + 1];
    const int gm = (int)(g >> 46);
    out->global_peak = (int16_t)((g & 1ull) ? -gm : gm);
    out->global_power = power_db(all, frames * channels);
    return COOLMIC_ERROR_NONE;
}

// ---------------------------------------------------------------------------
// measurement

extern "C" int cmhip_batch_timing(cmhip_batch_t *b, int enable)
{
    if (!b)
        return fail(COOLMIC_ERROR_FAULT, "timing: batch is NULL");
    b->timing = enable != 0;
    b->timing_every = enable > 1 ? (unsigned)enable : 1u;
    b->timing_count = 0;
    return COOLMIC_ERROR_NONE;
}

extern "C" int cmhip_batch_timing_read(cmhip_batch_t *b, double *kernel_ms, unsigned int *launches)
{
    if (!b)
        return fail(COOLMIC_ERROR_FAULT, "timing_read: batch is NULL");
    if (use(b))
        return COOLMIC_ERROR_GENERIC;
    HIP_TRY(hipStreamSynchronize(b->stream));
    double ms = 0.;
    for (auto &e : b->ev_used) {
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, e.a, e.b));
        ms += t;
    }
    if (kernel_ms)
        *kernel_ms = ms;
    if (launches)
        *launches = (unsigned)b->ev_used.size();
    b->ev_free.insert(b->ev_free.end(), b->ev_used.begin(), b->ev_used.end());
    b->ev_used.clear();
    return COOLMIC_ERROR_NONE;
}

extern "C" double cmhip_batch_ceiling(cmhip_batch_t *b, int mode, size_t frames, int iters)
{
    if (!b || iters <= 0 || frames > b->d.max_frames || (mode == 1 && (!b->d_out || b->d_out == b->d_in))) {
        fail(COOLMIC_ERROR_INVAL, "ceiling: bad arguments (copy needs a separate PCM output)");
        return -1.;
    }
    if (hipSetDevice(b->d.device) != hipSuccess)
        return -1.;
    // whole slots, so that the byte count is exact and contiguous
    (void)frames;
    const size_t bytes = (size_t)b->d.streams * b->stride * sizeof(int16_t);
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
        return -1.;
    for (int i = 0; i < 2; i++)
        (void)launch_ceiling(mode, b->d_in, b->d_out, bytes, b->d_sink, b->stream);
    (void)hipEventRecord(e0, b->stream);
    for (int i = 0; i < iters; i++)
        (void)launch_ceiling(mode, b->d_in, b->d_out, bytes, b->d_sink, b->stream);
    (void)hipEventRecord(e1, b->stream);
    float ms = 0.f;
    if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) {
        fail(COOLMIC_ERROR_GENERIC, "ceiling: event timing failed");
        return -1.;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    const double moved = (double)bytes * (mode == 1 ? 2. : 1.) * iters;
    return moved / (ms * 1e-3) / 1e9;
}
