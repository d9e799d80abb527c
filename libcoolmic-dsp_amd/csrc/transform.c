/* transform.c -- per-stream PCM operator in front of the HIP engine
 * (contract: <coolmic-dsp/transform.h>; ref: src/transform.c).
 *
 * The host keeps exactly the reference's framing: a read is cut to whole frames,
 * a carried partial frame goes first, upstream is asked once, the new partial
 * frame is kept for next time (ref: src/transform.c:126-165).  The arithmetic of
 * the whole frames -- channel map, gain, saturation, equaliser -- runs on the GPU
 * through a one-stream cmhip batch; nothing is computed on the CPU.  Only the reference's
 * own early-out (gain disabled, ref: src/transform.c:107-108) skips the device.
 *
 * Parameter setters may be called from another thread while the worker reads
 * (ref: src/simple.c:759-766); they publish under a mutex and take effect at the
 * next read.
 */
#define COOLMIC_COMPONENT "libcoolmic-dsp/transform"
#include "host_internal.h"
#include <coolmic-dsp/transform.h>
#include <coolmic-dsp/vumeter.h>
#include <coolmic_hip.h>

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define TRANSFORM_SLICE_FRAMES 16384u
#define TRANSFORM_RING_SLOTS   256u     /* windows of the last launches that stay on the device (records mode) */

enum { VU_NONE = 0, VU_DIRECT = 1, VU_RECORDS = 2 };

/* records mode: what one launch covered, and its window once it has been fetched */
typedef struct {
    uint64_t off;                      /* first byte, in bytes this transform's handle has returned */
    uint32_t bytes;
    uint64_t seq;                      /* the launch's sequence number in the batch's window ring */
    int have;                          /* raw is valid (fetched from the device) */
    cmhip_vu_raw_t raw;
} transform_record_t;

struct coolmic_transform {
    igloo_ro_base_t __base;
    coolmic_iohandle_t *io;
    unsigned char carry[2 * COOLMIC_DSP_TRANSFORM_MAX_CHANNELS - 1];
    size_t carry_fill;
    uint_least32_t rate;
    unsigned int channels;

    pthread_mutex_t lock;              /* guards the published parameters */
    uint16_t scale;                    /* 0: gain disabled */
    uint16_t gain[COOLMIC_DSP_TRANSFORM_MAX_CHANNELS];
    uint8_t chmap[COOLMIC_DSP_TRANSFORM_MAX_CHANNELS];
    int map_identity;
    unsigned int eq_sections;          /* 0: no equaliser */
    float eq_coef[5 * COOLMIC_DSP_TRANSFORM_MAX_EQ_SECTIONS];
    int eq_clear;                      /* filter state is to be zeroed before the next block */
    int dirty;                         /* device copy is stale */
    unsigned long gen;                 /* bumped by every setter: dirty / eq_clear are only cleared
                                        * once the values they were set with have reached the device */

    cmhip_batch_t *dev;                /* created at the first read that needs it */
    int device_plus1;                  /* coolmic_transform_set_device(): GPU + 1, 0 = the process's default */
    /* A VU meter downstream shares this transform's launch (one launch per pull instead of two).  All three
     * fields are published under `lock` and read once per block (transform_process):
     *   VU_DIRECT   the meter sits directly on this transform's handle: the launch accumulates the meter's
     *               window beside its own arithmetic, for the reads the meter arms (coolmic_transform_fuse_vu);
     *   VU_RECORDS  the meter sits behind a coolmic_tee_t (ref: src/simple.c:217-229) and may lag behind what
     *               this transform has produced: every launch leaves a window record of its own, and the meter
     *               merges the records of exactly the bytes it has consumed (coolmic_transform_records). */
    int vu_mode;
    int vu_armed;                      /* VU_DIRECT: the read that is under way is the meter's own */
    int vu_reset_pending;              /* the window is to be cleared before the next block touches it */

    uint64_t out_bytes;                /* bytes this transform's handle has returned so far */
    /* VU_RECORDS (worker thread only): one descriptor per launch, oldest first, in a circular array */
    transform_record_t *rec;
    size_t rec_cap, rec_head, rec_count;
    uint64_t rec_dropped;              /* descriptors dropped so far (absolute index of rec_head) */
    uint64_t rec_cursor;               /* absolute index where the last lookup ended */
    uint64_t rec_start;                /* out_bytes when the records were switched on */
    uint64_t rec_fetched_seq;          /* ring windows below this sequence number have been fetched */
    int ring_on;                       /* the device batch is in ring mode */
};

static void transform_destroy(void *self)
{
    coolmic_transform_t *t = self;
    coolmic_ro_unref(t->io);
    cmhip_batch_free(t->dev);
    free(t->rec);
    pthread_mutex_destroy(&t->lock);
}

COOLMIC_RO_TYPE(coolmic_transform_t, transform_destroy);

coolmic_transform_t *coolmic_transform_new(const char *name, igloo_ro_t associated,
                                           uint_least32_t rate, unsigned int channels)
{
    coolmic_transform_t *t;
    unsigned int c;

    if (!rate || !channels || channels > COOLMIC_DSP_TRANSFORM_MAX_CHANNELS)
        return NULL;
    t = COOLMIC_RO_NEW(coolmic_transform_t, name, associated);
    if (t == NULL)
        return NULL;
    pthread_mutex_init(&t->lock, NULL);
    t->rate = rate;
    t->channels = channels;
    for (c = 0; c < COOLMIC_DSP_TRANSFORM_MAX_CHANNELS; c++)
        t->chmap[c] = (uint8_t)(c < channels ? c : 0);
    t->map_identity = 1;
    return t;
}

int coolmic_transform_set_device(coolmic_transform_t *self, int device)
{
    int rc;

    if (self == NULL)
        return COOLMIC_ERROR_FAULT;
    rc = coolmic_hip_check_device(device);
    if (rc != COOLMIC_ERROR_NONE)
        return rc;
    pthread_mutex_lock(&self->lock);
    if (self->dev != NULL)
        rc = COOLMIC_ERROR_BUSY;       /* parameters, filter state and window live on the GPU it started on */
    else
        self->device_plus1 = device + 1;
    pthread_mutex_unlock(&self->lock);
    return rc;
}

int coolmic_transform_attach_iohandle(coolmic_transform_t *self, coolmic_iohandle_t *handle)
{
    if (self == NULL)
        return COOLMIC_ERROR_FAULT;
    coolmic_ro_unref(self->io);        /* NULL is fine */
    self->io = handle;
    coolmic_ro_ref(handle);            /* so is detaching with NULL */
    return COOLMIC_ERROR_NONE;
}

/* the parameters read at generation `gen` are on the device (or no device exists yet) */
static void transform_settled(coolmic_transform_t *t, unsigned long gen)
{
    pthread_mutex_lock(&t->lock);
    if (t->gen == gen) {               /* no setter ran in between */
        t->dirty = 0;
        t->eq_clear = 0;
    }
    pthread_mutex_unlock(&t->lock);
}

/* ---- records mode: the descriptor FIFO ---------------------------------------------------------- */

static transform_record_t *rec_at_index(coolmic_transform_t *t, size_t i)
{
    return &t->rec[(t->rec_head + i) % t->rec_cap];
}

static void rec_drop_front(coolmic_transform_t *t, size_t n)
{
    t->rec_head = (t->rec_head + n) % (t->rec_cap ? t->rec_cap : 1);
    t->rec_count -= n;
    t->rec_dropped += n;
    if (t->rec_cursor < t->rec_dropped)
        t->rec_cursor = t->rec_dropped;
}

static int rec_push(coolmic_transform_t *t, uint64_t off, uint32_t bytes, uint64_t seq)
{
    transform_record_t *r;

    if (t->rec_count == t->rec_cap) {
        const size_t cap = t->rec_cap ? 2 * t->rec_cap : 64;
        transform_record_t *grown = malloc(cap * sizeof(*grown));
        size_t i;
        if (grown == NULL)
            return -1;
        for (i = 0; i < t->rec_count; i++)
            grown[i] = *rec_at_index(t, i);
        free(t->rec);
        t->rec = grown;
        t->rec_cap = cap;
        t->rec_head = 0;
    }
    if (t->rec_count >= (1u << 16))    /* nobody consumes them (a meter that stopped reading): the oldest half goes */
        rec_drop_front(t, t->rec_count / 2);
    r = rec_at_index(t, t->rec_count++);
    r->off = off;
    r->bytes = bytes;
    r->seq = seq;
    r->have = 0;
    return 0;
}

/* every window that is still only on the device comes to its descriptor (one copy for all of them); the
 * ring slots are cleared for their next turn */
static int rec_fetch_all(coolmic_transform_t *t)
{
    const uint64_t next = cmhip_batch_vu_ring_seq(t->dev);
    const unsigned int count = (unsigned int)(next - t->rec_fetched_seq);
    cmhip_vu_raw_t *tmp;
    size_t i;

    if (count == 0)
        return 0;
    tmp = malloc(count * sizeof(*tmp));
    if (tmp == NULL || cmhip_batch_vu_ring_fetch(t->dev, t->rec_fetched_seq, count, tmp) != COOLMIC_ERROR_NONE) {
        free(tmp);
        return -1;
    }
    for (i = t->rec_count; i-- > 0;) {           /* newest first: the unfetched ones are at the end */
        transform_record_t *r = rec_at_index(t, i);
        if (r->seq < t->rec_fetched_seq)
            break;
        r->raw = tmp[r->seq - t->rec_fetched_seq];
        r->have = 1;
    }
    free(tmp);
    t->rec_fetched_seq = next;
    return 0;
}

/* whole frames through the GPU, in place.  `off`: where these bytes lie in the handle's output.  0 on success. */
static int transform_process(coolmic_transform_t *t, int16_t *pcm, size_t frames, uint64_t off)
{
    uint16_t scale, gain[COOLMIC_DSP_TRANSFORM_MAX_CHANNELS];
    uint8_t chmap[COOLMIC_DSP_TRANSFORM_MAX_CHANNELS];
    float eq_coef[5 * COOLMIC_DSP_TRANSFORM_MAX_EQ_SECTIONS];
    unsigned int eq_sections;
    unsigned long gen;
    int identity, dirty, eq_clear, idle, vu_mode, vu_on, vu_reset;

    pthread_mutex_lock(&t->lock);
    scale = t->scale;
    memcpy(gain, t->gain, sizeof(gain));
    memcpy(chmap, t->chmap, sizeof(chmap));
    identity = t->map_identity;
    eq_sections = t->eq_sections;
    memcpy(eq_coef, t->eq_coef, sizeof(eq_coef));
    eq_clear = t->eq_clear;
    dirty = t->dirty;
    gen = t->gen;
    vu_mode = t->vu_mode;
    vu_on = vu_mode == VU_RECORDS || (vu_mode == VU_DIRECT && t->vu_armed);
    vu_reset = t->vu_reset_pending;
    if (t->dev != NULL)
        t->vu_reset_pending = 0;       /* done below; without a device there is no window yet */
    pthread_mutex_unlock(&t->lock);

    /* nothing to do, exactly as the reference (ref: src/transform.c:107-108) -- but a device
     * that exists must still hear about it: "equaliser off" also clears the filter state, and
     * a later set_eq() must not filter on from what the old one left behind */
    idle = scale == 0 && identity && eq_sections == 0 && !vu_on;
    if (idle && t->dev == NULL) {
        if (dirty || eq_clear)         /* a batch made later starts from zero state and uploads everything */
            transform_settled(t, gen);
        return 0;
    }
    if (idle && !dirty && !eq_clear && !vu_reset)
        return 0;

    if (t->dev == NULL) {
        cmhip_batch_desc_t d;
        memset(&d, 0, sizeof(d));
        pthread_mutex_lock(&t->lock);
        d.device = coolmic_hip_stage_device(t->device_plus1);
        pthread_mutex_unlock(&t->lock);
        d.streams = 1;
        d.channels = t->channels;
        d.rate = (unsigned int)t->rate;
        d.max_frames = TRANSFORM_SLICE_FRAMES;
        /* (always with a VU window: whether a meter reads it is decided later, and a batch made
         * anew would lose the equaliser's state) */
        d.flags = CMHIP_OUT_PCM | CMHIP_INPLACE | CMHIP_EQ | CMHIP_HOSTPCM | CMHIP_VU;
        {
            cmhip_batch_t *dev = cmhip_batch_new(&d);
            if (dev == NULL) {
                coolmic_logging_log(COOLMIC_LOGGING_LEVEL_ERROR, COOLMIC_ERROR_NOSYS,
                                    "no HIP engine for the transform (there is no CPU path): %s",
                                    cmhip_last_error());
                return -1;
            }
            dirty = 1;
            pthread_mutex_lock(&t->lock);
            t->dev = dev;              /* (under the lock: coolmic_transform_set_device looks at it from any thread) */
            t->vu_reset_pending = 0;   /* a new batch's window is empty */
            pthread_mutex_unlock(&t->lock);
        }
        vu_reset = 0;
    }
    if (vu_reset && cmhip_batch_vu_reset(t->dev, 0) != COOLMIC_ERROR_NONE)
        return -1;
    /* the batch follows the mode: a window ring for the records, the one window otherwise */
    if ((vu_mode == VU_RECORDS) != (t->ring_on != 0)) {
        if (cmhip_batch_vu_ring(t->dev, vu_mode == VU_RECORDS ? TRANSFORM_RING_SLOTS : 0) != COOLMIC_ERROR_NONE) {
            coolmic_logging_log(COOLMIC_LOGGING_LEVEL_ERROR, COOLMIC_ERROR_GENERIC, "window ring: %s", cmhip_last_error());
            return -1;
        }
        t->ring_on = vu_mode == VU_RECORDS;
        t->rec_fetched_seq = cmhip_batch_vu_ring_seq(t->dev);
    }
    if (dirty || eq_clear) {
        int rc = cmhip_batch_set_gain(t->dev, 0, scale ? t->channels : 0, scale, gain);
        if (rc == COOLMIC_ERROR_NONE)
            rc = cmhip_batch_set_chmap(t->dev, 0, identity ? NULL : chmap);
        if (rc == COOLMIC_ERROR_NONE)
            rc = cmhip_batch_set_eq(t->dev, -1, eq_sections, eq_sections ? eq_coef : NULL);
        if (rc == COOLMIC_ERROR_NONE && eq_clear)
            rc = cmhip_batch_eq_reset(t->dev, -1);
        if (rc != COOLMIC_ERROR_NONE) {
            coolmic_logging_log(COOLMIC_LOGGING_LEVEL_ERROR, rc, "parameter upload failed: %s",
                                cmhip_last_error());
            return -1;                 /* dirty / eq_clear stay set: the next read tries again */
        }
        transform_settled(t, gen);
    }
    if (idle)
        return 0;
    cmhip_batch_vu_pause(t->dev, !vu_on);
    while (frames) {
        const size_t n = frames < TRANSFORM_SLICE_FRAMES ? frames : TRANSFORM_SLICE_FRAMES;
        uint64_t seq = 0;
        if (vu_mode == VU_RECORDS) {   /* this launch's slot of the ring must have been fetched and cleared */
            seq = cmhip_batch_vu_ring_seq(t->dev);
            if (seq - t->rec_fetched_seq >= TRANSFORM_RING_SLOTS && rec_fetch_all(t) != 0) {
                coolmic_logging_log(COOLMIC_LOGGING_LEVEL_ERROR, COOLMIC_ERROR_GENERIC,
                                    "fetching window records failed: %s", cmhip_last_error());
                return -1;
            }
        }
        if (cmhip_batch_upload(t->dev, 0, pcm, n) != COOLMIC_ERROR_NONE ||
            cmhip_batch_run(t->dev, n, NULL) != COOLMIC_ERROR_NONE ||
            cmhip_batch_download(t->dev, 0, pcm, n) != COOLMIC_ERROR_NONE) {
            coolmic_logging_log(COOLMIC_LOGGING_LEVEL_ERROR, COOLMIC_ERROR_GENERIC,
                                "HIP transform failed: %s", cmhip_last_error());
            return -1;
        }
        if (vu_mode == VU_RECORDS && rec_push(t, off, (uint32_t)(n * 2u * t->channels), seq) != 0)
            return -1;
        off += n * 2u * t->channels;
        pcm += n * t->channels;
        frames -= n;
    }
    return 0;
}

/* exported (not static) so that iohandle.c can recognise handles made here */
ssize_t coolmic_transform_handle_read(void *userdata, void *buffer, size_t len)
{
    coolmic_transform_t *t = userdata;
    unsigned char *dst = buffer;
    const size_t framesize = 2u * t->channels;
    size_t have = 0, tail;
    ssize_t got;

    len -= len % framesize;
    if (len == 0)
        return 0;

    if (t->carry_fill) {               /* < framesize <= len, always fits */
        memcpy(dst, t->carry, t->carry_fill);
        have = t->carry_fill;
        t->carry_fill = 0;
    }

    got = coolmic_iohandle_read(t->io, dst + have, len - have);
    if (got > 0)
        have += (size_t)got;           /* errors end up as a short (or empty) read */

    tail = have % framesize;
    if (tail) {
        memcpy(t->carry, dst + have - tail, tail);
        t->carry_fill = tail;
        have -= tail;
    }

    if (have && transform_process(t, buffer, have / framesize, t->out_bytes) != 0)
        return -1;
    t->out_bytes += have;
    return (ssize_t)have;
}

static int transform_handle_eof(void *userdata)
{
    coolmic_transform_t *t = userdata;
    /* a carried partial frame needs more upstream bytes anyway, so only upstream counts */
    if (t->io == NULL)
        return 1;
    return coolmic_iohandle_eof(t->io);
}

static int transform_handle_free(void *userdata)
{
    return coolmic_ro_unref((coolmic_transform_t *)userdata);
}

/* ---- the fused VU window (internal: vumeter.c; declared in host_internal.h) ------------------ */

/* A meter attached directly to this transform's handle, with the same rate and channel count,
 * sees exactly the frames this transform returns (ref: src/simple.c:212-229 wires them through a
 * tee, config 1 of BASELINE.json directly).  Then one launch does both loops of the reference --
 * __process (ref: src/transform.c:101-124) and the accumulate loop (ref: src/vumeter.c:161-177).
 * One meter at a time; the window starts empty (it is cleared by the next block, on the thread
 * that reads -- the mode may be switched from another thread while a read is under way). */
int coolmic_transform_fuse_vu(coolmic_transform_t *self, int on)
{
    int rc = COOLMIC_ERROR_NONE;

    if (self == NULL)
        return COOLMIC_ERROR_FAULT;
    pthread_mutex_lock(&self->lock);
    if (on && self->vu_mode != VU_NONE) {
        rc = COOLMIC_ERROR_BUSY;
    } else if (on || self->vu_mode == VU_DIRECT) {
        self->vu_mode = on ? VU_DIRECT : VU_NONE;
        self->vu_armed = 0;
        self->vu_reset_pending = 1;
    }
    pthread_mutex_unlock(&self->lock);
    return rc;
}

void coolmic_transform_arm_vu(coolmic_transform_t *self, int armed)
{
    if (self == NULL)
        return;
    pthread_mutex_lock(&self->lock);
    self->vu_armed = armed ? 1 : 0;
    pthread_mutex_unlock(&self->lock);
}

/* a reset that no block has carried out yet is carried out now (reader's thread) */
static int transform_vu_settle(coolmic_transform_t *self)
{
    int pending;

    pthread_mutex_lock(&self->lock);
    pending = self->vu_reset_pending;
    self->vu_reset_pending = 0;
    pthread_mutex_unlock(&self->lock);
    if (pending && self->dev != NULL && cmhip_batch_vu_reset(self->dev, 0) != COOLMIC_ERROR_NONE)
        return COOLMIC_ERROR_GENERIC;
    return COOLMIC_ERROR_NONE;
}

int coolmic_transform_vu_result(coolmic_transform_t *self, coolmic_vumeter_result_t *result)
{
    if (self == NULL || result == NULL)
        return COOLMIC_ERROR_FAULT;
    if (self->dev == NULL)
        return COOLMIC_ERROR_INVAL;    /* no frame has passed yet (ref: src/vumeter.c:198-199) */
    if (transform_vu_settle(self) != COOLMIC_ERROR_NONE)
        return COOLMIC_ERROR_GENERIC;
    return cmhip_batch_vu_result(self->dev, 0, result);
}

/* the window so far, raw, and a fresh one from here (a meter that leaves takes its frames along) */
int coolmic_transform_vu_take_raw(coolmic_transform_t *self, cmhip_vu_raw_t *raw)
{
    if (self == NULL || raw == NULL)
        return COOLMIC_ERROR_FAULT;
    if (self->dev == NULL)
        return COOLMIC_ERROR_INVAL;
    if (transform_vu_settle(self) != COOLMIC_ERROR_NONE ||
        cmhip_batch_vu_raw_state(self->dev, 0, raw) != COOLMIC_ERROR_NONE ||
        cmhip_batch_vu_reset(self->dev, 0) != COOLMIC_ERROR_NONE)
        return COOLMIC_ERROR_GENERIC;
    return COOLMIC_ERROR_NONE;
}

int coolmic_transform_vu_reset(coolmic_transform_t *self)
{
    if (self == NULL)
        return COOLMIC_ERROR_FAULT;
    pthread_mutex_lock(&self->lock);
    self->vu_reset_pending = 1;
    pthread_mutex_unlock(&self->lock);
    return transform_vu_settle(self);
}

/* ---- window records for a meter behind a tee (internal: vumeter.c) ----------------------------- */

/* on: every launch from now on leaves a record {bytes it covered, its own VU window}.  The windows stay
 * on the device, in a ring, until somebody needs their values (coolmic_transform_records_merge, or the
 * ring coming round): a pull costs no more than it does for the direct meter. */
int coolmic_transform_records(coolmic_transform_t *self, int on)
{
    int rc = COOLMIC_ERROR_NONE;

    if (self == NULL)
        return COOLMIC_ERROR_FAULT;
    pthread_mutex_lock(&self->lock);
    if (on && self->vu_mode != VU_NONE)
        rc = COOLMIC_ERROR_BUSY;
    else if (on)
        self->vu_mode = VU_RECORDS;
    else if (self->vu_mode == VU_RECORDS)
        self->vu_mode = VU_NONE;
    pthread_mutex_unlock(&self->lock);
    if (rc == COOLMIC_ERROR_NONE) {
        self->rec_start = self->out_bytes;
        rec_drop_front(self, self->rec_count);
    }
    return rc;
}

uint64_t coolmic_transform_out_bytes(const coolmic_transform_t *self)
{
    return self->out_bytes;
}

uint64_t coolmic_transform_records_start(const coolmic_transform_t *self)
{
    return self->rec_start;
}

/* the record that covers byte `pos` of the output */
int coolmic_transform_record_at(coolmic_transform_t *self, uint64_t pos, uint64_t *off, uint32_t *bytes)
{
    size_t i = self->rec_cursor >= self->rec_dropped ? (size_t)(self->rec_cursor - self->rec_dropped) : 0;

    if (i >= self->rec_count || rec_at_index(self, i)->off > pos)
        i = 0;
    for (; i < self->rec_count; i++) {
        const transform_record_t *r = rec_at_index(self, i);
        if (pos < r->off)
            break;
        if (pos < r->off + r->bytes) {
            self->rec_cursor = self->rec_dropped + i;
            *off = r->off;
            *bytes = r->bytes;
            return COOLMIC_ERROR_NONE;
        }
    }
    return COOLMIC_ERROR_INVAL;
}

/* merges, oldest first, the records that lie wholly inside [from, to) into `acc` and drops them together
 * with everything older.  Records inside the range must be contiguous from `from` (they are, for bytes
 * that came out of this handle one after the other); COOLMIC_ERROR_INVAL if one is missing. */
int coolmic_transform_records_merge(coolmic_transform_t *self, uint64_t from, uint64_t to, cmhip_vu_raw_t *acc)
{
    size_t i, n = 0;
    uint64_t at = from;

    for (i = 0; i < self->rec_count; i++) {
        transform_record_t *r = rec_at_index(self, i);
        if (r->off + r->bytes <= from) {            /* older than the range: consumed in pieces, or skipped */
            n = i + 1;
            continue;
        }
        if (r->off < from || r->off + r->bytes > to)
            break;
        if (r->off != at)
            return COOLMIC_ERROR_INVAL;
        if (!r->have && (self->dev == NULL || rec_fetch_all(self) != 0 || !r->have))
            return COOLMIC_ERROR_GENERIC;
        cmhip_vu_raw_merge(acc, &r->raw, self->channels);
        at = r->off + r->bytes;
        n = i + 1;
    }
    if (at != to && !(from == to))
        return COOLMIC_ERROR_INVAL;                 /* the range does not end on a record boundary */
    rec_drop_front(self, n);
    return COOLMIC_ERROR_NONE;
}

/* drops the records that end at or before `upto` (their bytes will not be asked for) */
void coolmic_transform_records_drop(coolmic_transform_t *self, uint64_t upto)
{
    size_t n = 0;

    while (n < self->rec_count && rec_at_index(self, n)->off + rec_at_index(self, n)->bytes <= upto)
        n++;
    rec_drop_front(self, n);
}

void coolmic_transform_format(const coolmic_transform_t *self, uint_least32_t *rate, unsigned int *channels)
{
    *rate = self->rate;
    *channels = self->channels;
}

coolmic_iohandle_t *coolmic_transform_get_iohandle(coolmic_transform_t *self)
{
    coolmic_iohandle_t *h;

    if (coolmic_ro_ref(self) != COOLMIC_ERROR_NONE)
        return NULL;
    h = coolmic_iohandle_new(NULL, igloo_RO_NULL, self, transform_handle_free,
                             coolmic_transform_handle_read, transform_handle_eof);
    if (h == NULL)
        coolmic_ro_unref(self);
    return h;
}

int coolmic_transform_set_master_gain(coolmic_transform_t *self, unsigned int channels,
                                      uint16_t scale, const uint16_t *gain)
{
    int rc = COOLMIC_ERROR_NONE;
    unsigned int c;

    if (self == NULL)
        return COOLMIC_ERROR_FAULT;

    pthread_mutex_lock(&self->lock);
    if (!channels || !scale || !gain) {
        self->scale = 0;
    } else if (channels == self->channels) {
        memcpy(self->gain, gain, sizeof(*gain) * channels);
        self->scale = scale;
    } else if (channels == 1) {
        for (c = 0; c < self->channels; c++)
            self->gain[c] = gain[0];
        self->scale = scale;
    } else if (channels == 2 && self->channels == 1) {
        self->gain[0] = (uint16_t)(((uint32_t)gain[0] + (uint32_t)gain[1]) / 2u);
        self->scale = scale;
        coolmic_logging_log(COOLMIC_LOGGING_LEVEL_DEBUG, COOLMIC_ERROR_NONE,
                            "gain: scale=%u, gain[0]=%u (in: %u, %u)", (unsigned int)scale,
                            (unsigned int)self->gain[0], (unsigned int)gain[0],
                            (unsigned int)gain[1]);
    } else {
        rc = COOLMIC_ERROR_INVAL;
    }
    if (rc == COOLMIC_ERROR_NONE) {
        self->dirty = 1;
        self->gen++;
    }
    pthread_mutex_unlock(&self->lock);
    return rc;
}

int coolmic_transform_set_channel_map(coolmic_transform_t *self, const uint8_t *map)
{
    unsigned int c;
    int identity = 1;

    if (self == NULL)
        return COOLMIC_ERROR_FAULT;
    if (map != NULL)
        for (c = 0; c < self->channels; c++)
            if (map[c] >= self->channels)
                return COOLMIC_ERROR_INVAL;

    pthread_mutex_lock(&self->lock);
    for (c = 0; c < self->channels; c++) {
        self->chmap[c] = map ? map[c] : (uint8_t)c;
        if (self->chmap[c] != c)
            identity = 0;
    }
    self->map_identity = identity;
    self->dirty = 1;
    self->gen++;
    pthread_mutex_unlock(&self->lock);
    return COOLMIC_ERROR_NONE;
}

int coolmic_transform_set_eq(coolmic_transform_t *self, unsigned int sections, const float *coef)
{
    if (self == NULL)
        return COOLMIC_ERROR_FAULT;
    if (sections > COOLMIC_DSP_TRANSFORM_MAX_EQ_SECTIONS || (sections && coef == NULL))
        return COOLMIC_ERROR_INVAL;

    pthread_mutex_lock(&self->lock);
    if (sections)
        memcpy(self->eq_coef, coef, sizeof(float) * 5u * sections);
    else
        self->eq_clear = 1;            /* off: the next filter starts from silence */
    self->eq_sections = sections;
    self->dirty = 1;
    self->gen++;
    pthread_mutex_unlock(&self->lock);
    return COOLMIC_ERROR_NONE;
}
