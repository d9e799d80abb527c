// group.hip -- coolmic_group_t: many per-stream pipelines on one cmhip batch
// (contract: <coolmic-dsp/group.h>).  Host code only; the arithmetic is the batch's.
#include <hip/hip_runtime.h>

#include <stdlib.h>
#include <string.h>

#include <deque>
#include <vector>

#define COOLMIC_COMPONENT "libcoolmic-dsp/group"
#include "host_internal.h"
#include <coolmic-dsp/group.h>
#include <coolmic_hip.h>
#include "work_pool.h"

// A block's PCM never leaves the pinned set it was produced in until a reader copies it out: the
// batch has no PCM arrays of its own (CMHIP_EXTSLOTS); the group owns two input sets and a ring of
// output sets in pinned, device-mapped host memory, and every launch reads one and writes one over
// PCIe (cmhip_batch_run_slots).  Sources write straight into the set the kernel will read; readers
// read straight from the set the kernel wrote.  (Round 1 copied host -> device -> host with the copy
// engines and then every block once more into a per-stream queue: the host's memory traffic, not the
// GPU, set the rate.)
struct GroupSeg {                            // a stream's share of one block, still in its output set
    unsigned int set;
    uint32_t bytes, pos;
};

struct GroupStream {
    coolmic_iohandle_t *source;
    unsigned char carry[2 * COOLMIC_DSP_VUMETER_MAX_CHANNELS - 1];   // partial input frame
    size_t carry_fill;
    std::deque<GroupSeg> segs;               // processed PCM not yet read, oldest first
    std::vector<unsigned char> spill;        // ... and what had to leave a set that was needed again
    size_t spill_pos;                        //     (older than every segment)
    size_t pending;                          // bytes in spill and segments together
    bool shared_source;                      // another stream of the group reads the same handle or the same backend
                                             // state: always pulled on the pumping thread, in slot order
};

struct coolmic_group {
    coolmic_ro_base_t base;
    uint_least32_t rate;
    unsigned int channels, max_streams, queue_blocks;
    size_t block_frames;
    cmhip_batch_t *batch;
    int device;                              // the GPU the batch lives on (coolmic_group_new_on)
    int16_t *h_in[2];                        // input sets, host view
    void *d_in[2];                           // ... device view
    std::vector<int16_t *> *h_out;           // ring of output sets (queue_blocks + 2), host view
    std::vector<void *> *d_out;
    std::vector<unsigned int> *out_users;    // streams with unread bytes in a set
    unsigned int out_next;                   // set the next block is written to
    unsigned int flight_set;                 // set of the block on the GPU
    unsigned int cur;                        // input set the next pull fills
    size_t stride;                           // samples between slots
    std::vector<GroupStream> *streams;
    std::vector<uint32_t> *nframes;          // frames per stream of the block being pulled
    std::vector<uint32_t> *flight;           // ... of the block on the GPU
    bool in_flight;                          // a block's launch is queued
    WorkPool *pullers;                       // helpers for the pull of a pump (coolmic_group_set_pull_threads)
};

struct GroupHandle {
    coolmic_group_t *group;
    unsigned int slot;
};

static void group_destroy(void *self)
{
    coolmic_group_t *g = (coolmic_group_t *)self;
    delete g->pullers;
    if (g->streams) {
        for (auto &s : *g->streams)
            coolmic_ro_unref(s.source);
        delete g->streams;
    }
    delete g->nframes;
    delete g->flight;
    if (g->in_flight && g->batch)            // the last block's kernel still uses the sets
        (void)cmhip_batch_sync(g->batch);
    cmhip_batch_free(g->batch);
    for (int i = 0; i < 2; i++)
        cmhip_host_free(g->h_in[i]);
    if (g->h_out)
        for (int16_t *p : *g->h_out)
            cmhip_host_free(p);
    delete g->h_out;
    delete g->d_out;
    delete g->out_users;
}

static const coolmic_ro_type_t group_type = {"coolmic_group_t", sizeof(coolmic_group_t), group_destroy};

extern "C" coolmic_group_t *coolmic_group_new(const char *name, igloo_ro_t associated,
                                              uint_least32_t rate, unsigned int channels,
                                              unsigned int max_streams, size_t block_frames,
                                              unsigned int queue_blocks)
{
    return coolmic_group_new_on(coolmic_hip_default_device(), name, associated, rate, channels, max_streams,
                                block_frames, queue_blocks);
}

extern "C" int coolmic_group_device(coolmic_group_t *self) { return self ? self->device : -1; }
extern "C" struct cmhip_batch *coolmic_group_engine(coolmic_group_t *self) { return self ? self->batch : NULL; }

extern "C" coolmic_group_t *coolmic_group_new_on(int device, const char *name, igloo_ro_t associated,
                                                 uint_least32_t rate, unsigned int channels,
                                                 unsigned int max_streams, size_t block_frames,
                                                 unsigned int queue_blocks)
{
    if (!rate || !channels || channels > COOLMIC_DSP_VUMETER_MAX_CHANNELS || !max_streams ||
        !block_frames)
        return NULL;
    if (coolmic_hip_check_device(device) != COOLMIC_ERROR_NONE) {
        coolmic_logging_log(COOLMIC_LOGGING_LEVEL_ERROR, COOLMIC_ERROR_NOSYS,
                            "no HIP device %d for the group (%d visible; there is no CPU path)", device,
                            cmhip_device_count());
        return NULL;
    }
    coolmic_group_t *g = (coolmic_group_t *)coolmic_ro_new_raw(&group_type, name, associated);
    if (!g)
        return NULL;
    g->rate = rate;
    g->channels = channels;
    g->max_streams = max_streams;
    g->block_frames = block_frames;
    g->queue_blocks = queue_blocks ? queue_blocks : 1;
    g->device = device;

    cmhip_batch_desc_t d;
    memset(&d, 0, sizeof(d));
    d.device = device;
    d.streams = max_streams;
    d.channels = channels;
    d.rate = (unsigned int)rate;
    d.max_frames = block_frames;
    d.flags = CMHIP_OUT_PCM | CMHIP_VU | CMHIP_EQ | CMHIP_EXTSLOTS;    // the equaliser is off until coolmic_group_set_eq()
    g->batch = cmhip_batch_new(&d);
    if (!g->batch) {
        coolmic_logging_log(COOLMIC_LOGGING_LEVEL_ERROR, COOLMIC_ERROR_NOSYS,
                            "no HIP engine for the group (there is no CPU path): %s",
                            cmhip_last_error());
        coolmic_ro_unref(g);
        return NULL;
    }
    g->stride = cmhip_batch_stride(g->batch);
    const size_t bytes = (size_t)max_streams * g->stride * sizeof(int16_t);
    const unsigned nout = g->queue_blocks + 2u;          // one on the GPU, queue_blocks being read, one spare
    g->h_out = new std::vector<int16_t *>(nout, nullptr);
    g->d_out = new std::vector<void *>(nout, nullptr);
    g->out_users = new std::vector<unsigned int>(nout, 0);
    bool ok = true;
    for (int i = 0; i < 2 && ok; i++) {
        g->h_in[i] = (int16_t *)cmhip_host_alloc_mapped_on(device, bytes, &g->d_in[i]);
        ok = g->h_in[i] != nullptr;
        if (ok)
            memset(g->h_in[i], 0, bytes);
    }
    for (unsigned i = 0; i < nout && ok; i++) {
        (*g->h_out)[i] = (int16_t *)cmhip_host_alloc_mapped_on(device, bytes, &(*g->d_out)[i]);
        ok = (*g->h_out)[i] != nullptr;
    }
    if (!ok) {
        coolmic_logging_log(COOLMIC_LOGGING_LEVEL_ERROR, COOLMIC_ERROR_NOMEM,
                            "pinned, device-mapped PCM sets of %zu bytes x %u failed", bytes, nout + 2u);
        coolmic_ro_unref(g);
        return NULL;
    }
    g->streams = new std::vector<GroupStream>();
    g->streams->reserve(max_streams);
    g->nframes = new std::vector<uint32_t>(max_streams, 0);
    g->flight = new std::vector<uint32_t>(max_streams, 0);
    return g;
}

extern "C" unsigned int coolmic_group_streams(coolmic_group_t *self)
{
    return self && self->streams ? (unsigned int)self->streams->size() : 0;
}

extern "C" int coolmic_group_add_stream(coolmic_group_t *self, coolmic_iohandle_t *source)
{
    if (!self || !source)
        return COOLMIC_ERROR_FAULT;
    if (self->streams->size() >= self->max_streams)
        return COOLMIC_ERROR_BUSY;
    GroupStream s;
    s.source = source;
    s.carry_fill = 0;
    s.spill_pos = 0;
    s.pending = 0;
    s.shared_source = false;
    // Two streams over one upstream object (the same handle twice, or two handles whose backend is one device,
    // one file, one tee) would race on its state once the pump's reads run on several threads
    // (coolmic_group_set_pull_threads): such streams are marked and keep being read by the pumping thread
    // alone, one after the other, as a single-threaded pump reads everything.
    const void *backend = coolmic_iohandle_backend(source);
    for (auto &o : *self->streams) {
        if (o.source == source || (backend != nullptr && coolmic_iohandle_backend(o.source) == backend)) {
            o.shared_source = true;
            s.shared_source = true;
        }
    }
    coolmic_ro_ref(source);
    self->streams->push_back(std::move(s));
    return (int)self->streams->size() - 1;
}

extern "C" int coolmic_group_set_master_gain(coolmic_group_t *self, unsigned int slot,
                                             unsigned int channels, uint16_t scale,
                                             const uint16_t *gain)
{
    if (!self)
        return COOLMIC_ERROR_FAULT;
    if (slot >= self->streams->size())
        return COOLMIC_ERROR_INVAL;
    return cmhip_batch_set_gain(self->batch, (long)slot, channels, scale, gain);
}

extern "C" int coolmic_group_set_channel_map(coolmic_group_t *self, unsigned int slot,
                                             const uint8_t *map)
{
    if (!self)
        return COOLMIC_ERROR_FAULT;
    if (slot >= self->streams->size())
        return COOLMIC_ERROR_INVAL;
    return cmhip_batch_set_chmap(self->batch, (long)slot, map);
}

extern "C" int coolmic_group_set_eq(coolmic_group_t *self, int slot, unsigned int sections,
                                    const float *coef)
{
    if (!self)
        return COOLMIC_ERROR_FAULT;
    if (slot < -1 || (slot >= 0 && (size_t)slot >= self->streams->size()))
        return COOLMIC_ERROR_INVAL;
    const int rc = cmhip_batch_set_eq(self->batch, (long)slot, sections, coef);
    if (rc == COOLMIC_ERROR_NONE && sections == 0)
        return cmhip_batch_eq_reset(self->batch, -1);
    return rc;
}

// The block on the GPU comes home: wait for its launch; every stream gets a segment in the set the
// kernel wrote (nothing is copied).
static int group_complete(coolmic_group_t *self)
{
    if (!self->in_flight)
        return COOLMIC_ERROR_NONE;
    self->in_flight = false;
    if (cmhip_batch_sync(self->batch) != COOLMIC_ERROR_NONE) {
        coolmic_logging_log(COOLMIC_LOGGING_LEVEL_ERROR, COOLMIC_ERROR_GENERIC,
                            "HIP group block failed: %s", cmhip_last_error());
        return COOLMIC_ERROR_GENERIC;
    }
    const size_t n = self->streams->size();
    const uint32_t framesize = 2u * self->channels;
    const unsigned set = self->flight_set;
    for (size_t i = 0; i < n; i++) {
        const uint32_t fr = (*self->flight)[i];
        if (!fr)
            continue;
        (*self->flight)[i] = 0;
        GroupStream &s = (*self->streams)[i];
        s.segs.push_back(GroupSeg{set, fr * framesize, 0});
        s.pending += (size_t)fr * framesize;
        (*self->out_users)[set]++;
    }
    return COOLMIC_ERROR_NONE;
}

// An output set is needed again while a slow reader still has bytes in it: they move to the
// stream's own spill buffer (sets are used in ring order, so it is the stream's oldest segment).
static void group_vacate(coolmic_group_t *self, unsigned set)
{
    if ((*self->out_users)[set] == 0)
        return;
    for (size_t i = 0; i < self->streams->size(); i++) {
        GroupStream &s = (*self->streams)[i];
        if (s.segs.empty() || s.segs.front().set != set)
            continue;
        const GroupSeg g = s.segs.front();
        s.segs.pop_front();
        if (s.spill_pos == s.spill.size()) {
            s.spill.clear();
            s.spill_pos = 0;
        }
        const unsigned char *src = (const unsigned char *)((*self->h_out)[set] + i * self->stride) + g.pos;
        s.spill.insert(s.spill.end(), src, src + (g.bytes - g.pos));
    }
    (*self->out_users)[set] = 0;
}

// the pull of streams [lo, hi) of one pump: one iohandle read per stream straight into its slot of the
// input set, framed like ref: src/transform.c:126-165.  Streams share nothing here (slot, carry and
// frame count are the stream's own), so ranges may run on different threads.
struct GroupPull {
    coolmic_group_t *g;
    int16_t *h_in;
    size_t framesize, block_bytes, queue_cap;
    int shared;                              // -1: every stream of the range; 0 / 1: only those without / with a shared source
};

static void group_pull_range(void *arg, unsigned lo, unsigned hi)
{
    const GroupPull *p = (const GroupPull *)arg;
    coolmic_group_t *self = p->g;
    for (size_t i = lo; i < hi; i++) {
        GroupStream &s = (*self->streams)[i];
        if (p->shared >= 0 && (int)s.shared_source != p->shared)
            continue;
        (*self->nframes)[i] = 0;
        const size_t coming = self->in_flight ? (size_t)(*self->flight)[i] * p->framesize : 0;
        if (s.pending + coming + p->block_bytes > p->queue_cap)
            continue;                          // this stream's reader is behind: no read-ahead
        unsigned char *dst = (unsigned char *)(p->h_in + i * self->stride);
        size_t have = 0;
        if (s.carry_fill) {
            memcpy(dst, s.carry, s.carry_fill);
            have = s.carry_fill;
            s.carry_fill = 0;
        }
        const ssize_t got = coolmic_iohandle_read(s.source, dst + have, p->block_bytes - have);
        if (got > 0)
            have += (size_t)got;
        const size_t tail = have % p->framesize;
        if (tail) {
            memcpy(s.carry, dst + have - tail, tail);
            s.carry_fill = tail;
            have -= tail;
        }
        (*self->nframes)[i] = (uint32_t)(have / p->framesize);
    }
}

extern "C" int coolmic_group_set_pull_threads(coolmic_group_t *self, unsigned int threads)
{
    if (!self)
        return COOLMIC_ERROR_FAULT;
    if (threads > 64)
        return COOLMIC_ERROR_INVAL;
    delete self->pullers;                    // (no pump is under way: the group is driven by one thread)
    self->pullers = threads > 1 ? new WorkPool(threads - 1) : nullptr;     // the pumping thread pulls too
    return COOLMIC_ERROR_NONE;
}

// One block: pull from every source straight into the input set the kernel will read, then (the
// previous block being home) launch this one on that set and the next output set, and return --
// the GPU works while the caller reads and the next pump pulls.  The block's PCM can be read with
// the next pump, or with the first read that finds a stream empty.
extern "C" int coolmic_group_pump(coolmic_group_t *self)
{
    if (!self)
        return COOLMIC_ERROR_FAULT;
    const size_t n = self->streams->size();
    if (n == 0)
        return 0;
    const size_t framesize = 2u * self->channels;
    const size_t block_bytes = self->block_frames * framesize;
    const size_t queue_cap = block_bytes * self->queue_blocks;
    int16_t *h_in = self->h_in[self->cur];
    uint32_t most = 0;
    int delivered = 0;

    // 1. pull: one iohandle read per stream (group_pull_range), on this thread or spread over the helpers
    GroupPull pull = {self, h_in, framesize, block_bytes, queue_cap, -1};
    if (self->pullers && n >= 16) {
        pull.shared = 0;                     // streams with a source of their own: spread over the helpers
        self->pullers->run(group_pull_range, (void *)&pull, (unsigned)n, 8);
        pull.shared = 1;                     // the rest here, in slot order
        group_pull_range((void *)&pull, 0, (unsigned)n);
    } else {
        group_pull_range((void *)&pull, 0, (unsigned)n);
    }
    for (size_t i = 0; i < n; i++) {
        const uint32_t fr = (*self->nframes)[i];
        if (fr > most)
            most = fr;
        if (fr)
            delivered++;
    }

    // 2. the block before this one
    if (group_complete(self) != COOLMIC_ERROR_NONE)
        return COOLMIC_ERROR_GENERIC;
    if (most == 0)
        return 0;

    // 3. one launch for the whole group: the kernel reads this input set and writes the next
    //    output set, both in host memory, over PCIe
    const unsigned set = self->out_next;
    group_vacate(self, set);
    if (cmhip_batch_run_slots(self->batch, most, self->nframes->data(), self->d_in[self->cur],
                              (*self->d_out)[set]) != COOLMIC_ERROR_NONE) {
        coolmic_logging_log(COOLMIC_LOGGING_LEVEL_ERROR, COOLMIC_ERROR_GENERIC,
                            "HIP group block failed: %s", cmhip_last_error());
        (void)cmhip_batch_sync(self->batch);
        return COOLMIC_ERROR_GENERIC;
    }
    std::swap(self->nframes, self->flight);
    self->in_flight = true;
    self->flight_set = set;
    self->out_next = (set + 1u) % (unsigned)self->h_out->size();
    self->cur ^= 1u;
    return delivered;
}

static ssize_t group_handle_read(void *userdata, void *buffer, size_t len)
{
    GroupHandle *h = (GroupHandle *)userdata;
    coolmic_group_t *g = h->group;
    GroupStream &s = (*g->streams)[h->slot];
    const size_t framesize = 2u * g->channels;
    unsigned char *dst = (unsigned char *)buffer;

    len -= len % framesize;
    if (len == 0)
        return 0;
    if (s.pending == 0) {                        // nothing waiting: the block on the GPU, else a new one
        if (group_complete(g) != COOLMIC_ERROR_NONE)
            return -1;
        if (s.pending == 0) {
            if (coolmic_group_pump(g) < 0 || group_complete(g) != COOLMIC_ERROR_NONE)
                return -1;
        }
    }
    size_t done = 0;
    if (s.spill_pos < s.spill.size()) {          // what left its set early comes first
        size_t k = s.spill.size() - s.spill_pos;
        if (k > len)
            k = len;
        memcpy(dst, s.spill.data() + s.spill_pos, k);
        s.spill_pos += k;
        done = k;
    }
    while (done < len && !s.segs.empty()) {
        GroupSeg &seg = s.segs.front();
        size_t k = seg.bytes - seg.pos;
        if (k > len - done)
            k = len - done;
        memcpy(dst + done, (const unsigned char *)((*g->h_out)[seg.set] + (size_t)h->slot * g->stride) + seg.pos, k);
        seg.pos += (uint32_t)k;
        done += k;
        if (seg.pos == seg.bytes) {
            (*g->out_users)[seg.set]--;
            s.segs.pop_front();
        }
    }
    s.pending -= done;
    return (ssize_t)done;
}

static int group_handle_eof(void *userdata)
{
    GroupHandle *h = (GroupHandle *)userdata;
    GroupStream &s = (*h->group->streams)[h->slot];
    if (s.pending)
        return 0;
    if (h->group->in_flight && (*h->group->flight)[h->slot])
        return 0;                              // frames of this stream are on their way
    if (!s.source)
        return 1;
    return coolmic_iohandle_eof(s.source);
}

static int group_handle_free(void *userdata)
{
    GroupHandle *h = (GroupHandle *)userdata;
    coolmic_ro_unref(h->group);
    free(h);
    return 0;
}

extern "C" coolmic_iohandle_t *coolmic_group_get_iohandle(coolmic_group_t *self, unsigned int slot)
{
    if (!self || slot >= self->streams->size())
        return NULL;
    GroupHandle *h = (GroupHandle *)calloc(1, sizeof(*h));
    if (!h)
        return NULL;
    h->group = self;
    h->slot = slot;
    coolmic_ro_ref(self);
    coolmic_iohandle_t *io = coolmic_iohandle_new(NULL, igloo_RO_NULL, h, group_handle_free,
                                                  group_handle_read, group_handle_eof);
    if (!io) {
        coolmic_ro_unref(self);
        free(h);
    }
    return io;
}

extern "C" int coolmic_group_vumeter_result(coolmic_group_t *self, unsigned int slot,
                                            coolmic_vumeter_result_t *result)
{
    if (!self || !result)
        return COOLMIC_ERROR_FAULT;
    if (slot >= self->streams->size())
        return COOLMIC_ERROR_INVAL;
    return cmhip_batch_vu_result(self->batch, slot, result);
}
