// cmhip_vu.hip -- VU windows of a batch on the host side: per-stream results, packed snapshots and their collect
// (the dB finish in double, as ref: src/vumeter.c:189-218), per-launch window records for the meters behind a
// tee, node records for the node-global VU.
#include "cmhip_engine.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <thread>

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
    if (cmhip_engine_settle_node(b))
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
    if (cmhip_engine_settle_node(b))
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
    if (cmhip_engine_settle_node(b))
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
    if (use(b) || cmhip_engine_settle_node(b))
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
    if (cmhip_engine_settle_node(b))
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

