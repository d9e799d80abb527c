// group.hip -- coolmic_group_t: many per-stream pipelines on one cmhip batch
// (contract: <coolmic-dsp/group.h>).  Host code only; the arithmetic is the batch's.
#include <hip/hip_runtime.h>

#include <stdlib.h>
#include <string.h>

#include <vector>

#include "work_pool.h"

#define COOLMIC_COMPONENT "libcoolmic-dsp/group"
#include "host_internal.h"
#include <coolmic-dsp/group.h>
#include <coolmic_hip.h>

struct GroupStream {
    coolmic_iohandle_t *source;
    unsigned char carry[2 * COOLMIC_DSP_VUMETER_MAX_CHANNELS - 1];   // partial input frame
    size_t carry_fill;
    std::vector<unsigned char> queue;        // processed PCM not yet read, whole frames
    size_t queue_pos;                        // bytes already handed out
};

struct coolmic_group {
    coolmic_ro_base_t base;
    uint_least32_t rate;
    unsigned int channels, max_streams, queue_blocks;
    size_t block_frames;
    cmhip_batch_t *batch;
    int16_t *h_in[2], *h_out;                // pinned mirrors of the batch's PCM slots (input: two sets)
    unsigned int cur;                        // input set the next pull fills
    size_t stride;                           // samples between slots
    std::vector<GroupStream> *streams;
    std::vector<uint32_t> *nframes;          // frames per stream of the block being pulled
    std::vector<uint32_t> *flight;           // ... of the block on the GPU
    bool in_flight;                          // upload + launch + download of a block are queued
    WorkPool *pool;                          // helpers for the queue copies of large groups
};

struct GroupHandle {
    coolmic_group_t *group;
    unsigned int slot;
};

static void group_destroy(void *self)
{
    coolmic_group_t *g = (coolmic_group_t *)self;
    if (g->streams) {
        for (auto &s : *g->streams)
            coolmic_ro_unref(s.source);
        delete g->streams;
    }
    delete g->nframes;
    delete g->flight;
    delete g->pool;
    if (g->in_flight)                        // the copies of the last block still use the staging
        (void)hipStreamSynchronize((hipStream_t)cmhip_batch_hip_stream(g->batch));
    for (int i = 0; i < 2; i++)
        if (g->h_in[i])
            (void)hipHostFree(g->h_in[i]);
    if (g->h_out)
        (void)hipHostFree(g->h_out);
    cmhip_batch_free(g->batch);
}

static const coolmic_ro_type_t group_type = {"coolmic_group_t", sizeof(coolmic_group_t), group_destroy};

extern "C" coolmic_group_t *coolmic_group_new(const char *name, igloo_ro_t associated,
                                              uint_least32_t rate, unsigned int channels,
                                              unsigned int max_streams, size_t block_frames,
                                              unsigned int queue_blocks)
{
    if (!rate || !channels || channels > COOLMIC_DSP_VUMETER_MAX_CHANNELS || !max_streams ||
        !block_frames)
        return NULL;
    coolmic_group_t *g = (coolmic_group_t *)coolmic_ro_new_raw(&group_type, name, associated);
    if (!g)
        return NULL;
    g->rate = rate;
    g->channels = channels;
    g->max_streams = max_streams;
    g->block_frames = block_frames;
    g->queue_blocks = queue_blocks ? queue_blocks : 1;

    cmhip_batch_desc_t d;
    memset(&d, 0, sizeof(d));
    d.device = coolmic_hip_default_device();
    d.streams = max_streams;
    d.channels = channels;
    d.rate = (unsigned int)rate;
    d.max_frames = block_frames;
    d.flags = CMHIP_OUT_PCM | CMHIP_VU | CMHIP_EQ;     // the equaliser is off until coolmic_group_set_eq()
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
    if (hipHostMalloc((void **)&g->h_in[0], bytes, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void **)&g->h_in[1], bytes, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void **)&g->h_out, bytes, hipHostMallocDefault) != hipSuccess) {
        coolmic_logging_log(COOLMIC_LOGGING_LEVEL_ERROR, COOLMIC_ERROR_NOMEM,
                            "pinned staging of %zu bytes x 3 failed", bytes);
        coolmic_ro_unref(g);
        return NULL;
    }
    memset(g->h_in[0], 0, bytes);
    memset(g->h_in[1], 0, bytes);
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
    s.queue_pos = 0;
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

// The block on the GPU comes home: wait for its download and hand the PCM to the streams' queues.
// With many streams the copies are shared out over a few helper threads (streams are independent).
static int group_complete(coolmic_group_t *self)
{
    if (!self->in_flight)
        return COOLMIC_ERROR_NONE;
    self->in_flight = false;
    if (hipStreamSynchronize((hipStream_t)cmhip_batch_hip_stream(self->batch)) != hipSuccess) {
        coolmic_logging_log(COOLMIC_LOGGING_LEVEL_ERROR, COOLMIC_ERROR_GENERIC,
                            "HIP group block failed: %s", cmhip_last_error());
        return COOLMIC_ERROR_GENERIC;
    }
    const size_t n = self->streams->size();
    const size_t framesize = 2u * self->channels;
    const size_t queue_cap = self->block_frames * framesize * self->queue_blocks;
    struct Job {
        coolmic_group_t *g;
        size_t framesize, queue_cap;
    } job = {self, framesize, queue_cap};
    auto body = [](void *p, unsigned lo, unsigned hi) {
        Job *j = (Job *)p;
        coolmic_group_t *g = j->g;
        for (unsigned i = lo; i < hi; i++) {
            const uint32_t fr = (*g->flight)[i];
            if (!fr)
                continue;
            (*g->flight)[i] = 0;
            GroupStream &s = (*g->streams)[i];
            if (s.queue_pos == s.queue.size()) {
                s.queue.clear();
                s.queue_pos = 0;
            } else if (s.queue_pos > j->queue_cap) {
                s.queue.erase(s.queue.begin(), s.queue.begin() + (ptrdiff_t)s.queue_pos);
                s.queue_pos = 0;
            }
            const unsigned char *src = (const unsigned char *)(g->h_out + i * g->stride);
            s.queue.insert(s.queue.end(), src, src + (size_t)fr * j->framesize);
        }
    };
    size_t bytes = 0;
    for (size_t i = 0; i < n; i++)
        bytes += (size_t)(*self->flight)[i] * framesize;
    if (bytes >= (4u << 20) && n >= 64) {            // worth waking helpers for
        if (!self->pool) {
            unsigned t = std::thread::hardware_concurrency() / 2;
            self->pool = new WorkPool(t < 1 ? 1 : (t > 6 ? 6 : t));
        }
        self->pool->run(body, &job, (unsigned)n, (unsigned)(n / 32 ? n / 32 : 1));
    } else {
        body(&job, 0, (unsigned)n);
    }
    return COOLMIC_ERROR_NONE;
}

// One block: pull from every source into pinned staging, then (the previous block being home)
// queue upload, launch and download of this one and return -- the GPU and the copies work while
// the caller reads the queues and the next pump pulls.  The block's PCM reaches the queues with
// the next pump, or with the first read that finds a queue empty.
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

    // 1. pull: one iohandle read per stream, framed like ref: src/transform.c:126-165
    for (size_t i = 0; i < n; i++) {
        GroupStream &s = (*self->streams)[i];
        (*self->nframes)[i] = 0;
        const size_t coming = self->in_flight ? (size_t)(*self->flight)[i] * framesize : 0;
        if (s.queue.size() - s.queue_pos + coming + block_bytes > queue_cap)
            continue;                          // this stream's reader is behind: no read-ahead
        unsigned char *dst = (unsigned char *)(h_in + i * self->stride);
        size_t have = 0;
        if (s.carry_fill) {
            memcpy(dst, s.carry, s.carry_fill);
            have = s.carry_fill;
            s.carry_fill = 0;
        }
        const ssize_t got = coolmic_iohandle_read(s.source, dst + have, block_bytes - have);
        if (got > 0)
            have += (size_t)got;
        const size_t tail = have % framesize;
        if (tail) {
            memcpy(s.carry, dst + have - tail, tail);
            s.carry_fill = tail;
            have -= tail;
        }
        const uint32_t fr = (uint32_t)(have / framesize);
        (*self->nframes)[i] = fr;
        if (fr > most)
            most = fr;
        if (fr)
            delivered++;
    }

    // 2. the block before this one: its download was queued a pump ago
    if (group_complete(self) != COOLMIC_ERROR_NONE)
        return COOLMIC_ERROR_GENERIC;
    if (most == 0)
        return 0;

    // 3. one upload, one launch, one download for the whole group, all queued on the batch's stream
    hipStream_t st = (hipStream_t)cmhip_batch_hip_stream(self->batch);
    const size_t span = ((n - 1) * self->stride + (size_t)most * self->channels) * sizeof(int16_t);
    if (hipMemcpyAsync(cmhip_batch_dev_in(self->batch), h_in, span, hipMemcpyHostToDevice, st) !=
            hipSuccess ||
        cmhip_batch_run(self->batch, most, self->nframes->data()) != COOLMIC_ERROR_NONE ||
        hipMemcpyAsync(self->h_out, cmhip_batch_dev_out(self->batch), span, hipMemcpyDeviceToHost, st) !=
            hipSuccess) {
        coolmic_logging_log(COOLMIC_LOGGING_LEVEL_ERROR, COOLMIC_ERROR_GENERIC,
                            "HIP group block failed: %s", cmhip_last_error());
        (void)hipStreamSynchronize(st);
        return COOLMIC_ERROR_GENERIC;
    }
    std::swap(self->nframes, self->flight);
    self->in_flight = true;
    self->cur ^= 1u;
    return delivered;
}

static ssize_t group_handle_read(void *userdata, void *buffer, size_t len)
{
    GroupHandle *h = (GroupHandle *)userdata;
    coolmic_group_t *g = h->group;
    GroupStream &s = (*g->streams)[h->slot];
    const size_t framesize = 2u * g->channels;

    len -= len % framesize;
    if (len == 0)
        return 0;
    if (s.queue_pos == s.queue.size()) {         // nothing buffered: the block on the GPU, else a new one
        if (group_complete(g) != COOLMIC_ERROR_NONE)
            return -1;
        if (s.queue_pos == s.queue.size()) {
            if (coolmic_group_pump(g) < 0 || group_complete(g) != COOLMIC_ERROR_NONE)
                return -1;
        }
    }
    size_t avail = s.queue.size() - s.queue_pos;
    if (avail > len)
        avail = len;
    if (avail) {
        memcpy(buffer, s.queue.data() + s.queue_pos, avail);
        s.queue_pos += avail;
    }
    return (ssize_t)avail;
}

static int group_handle_eof(void *userdata)
{
    GroupHandle *h = (GroupHandle *)userdata;
    GroupStream &s = (*h->group->streams)[h->slot];
    if (s.queue_pos != s.queue.size())
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
