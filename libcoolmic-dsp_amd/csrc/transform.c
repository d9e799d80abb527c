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
#include <string.h>

#define TRANSFORM_SLICE_FRAMES 16384u

struct coolmic_transform {
    coolmic_ro_base_t base;
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
    int fused_vu;                      /* a VU meter sits directly on this transform's handle: the launch
                                        * that transforms a block also accumulates its window (one launch
                                        * per pull instead of two; coolmic_transform_fuse_vu) */
    int vu_armed;                      /* ... for the read that is under way: it is the meter's own */
};

static void transform_destroy(void *self)
{
    coolmic_transform_t *t = self;
    coolmic_ro_unref(t->io);
    cmhip_batch_free(t->dev);
    pthread_mutex_destroy(&t->lock);
}

static const coolmic_ro_type_t transform_type = {
    "coolmic_transform_t", sizeof(coolmic_transform_t), transform_destroy
};

coolmic_transform_t *coolmic_transform_new(const char *name, igloo_ro_t associated,
                                           uint_least32_t rate, unsigned int channels)
{
    coolmic_transform_t *t;
    unsigned int c;

    if (!rate || !channels || channels > COOLMIC_DSP_TRANSFORM_MAX_CHANNELS)
        return NULL;
    t = coolmic_ro_new_raw(&transform_type, name, associated);
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

/* whole frames through the GPU, in place.  0 on success. */
static int transform_process(coolmic_transform_t *t, int16_t *pcm, size_t frames)
{
    uint16_t scale, gain[COOLMIC_DSP_TRANSFORM_MAX_CHANNELS];
    uint8_t chmap[COOLMIC_DSP_TRANSFORM_MAX_CHANNELS];
    float eq_coef[5 * COOLMIC_DSP_TRANSFORM_MAX_EQ_SECTIONS];
    unsigned int eq_sections;
    unsigned long gen;
    int identity, dirty, eq_clear, idle;

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
    pthread_mutex_unlock(&t->lock);

    /* nothing to do, exactly as the reference (ref: src/transform.c:107-108) -- but a device
     * that exists must still hear about it: "equaliser off" also clears the filter state, and
     * a later set_eq() must not filter on from what the old one left behind */
    idle = scale == 0 && identity && eq_sections == 0 && !(t->fused_vu && t->vu_armed);
    if (idle && t->dev == NULL) {
        if (dirty || eq_clear)         /* a batch made later starts from zero state and uploads everything */
            transform_settled(t, gen);
        return 0;
    }
    if (idle && !dirty && !eq_clear)
        return 0;

    if (t->dev == NULL) {
        cmhip_batch_desc_t d;
        memset(&d, 0, sizeof(d));
        d.device = coolmic_hip_default_device();
        d.streams = 1;
        d.channels = t->channels;
        d.rate = (unsigned int)t->rate;
        d.max_frames = TRANSFORM_SLICE_FRAMES;
        /* (always with a VU window: whether a meter reads it is decided later, and a batch made
         * anew would lose the equaliser's state) */
        d.flags = CMHIP_OUT_PCM | CMHIP_INPLACE | CMHIP_EQ | CMHIP_HOSTPCM | CMHIP_VU;
        t->dev = cmhip_batch_new(&d);
        if (t->dev == NULL) {
            coolmic_logging_log(COOLMIC_LOGGING_LEVEL_ERROR, COOLMIC_ERROR_NOSYS,
                                "no HIP engine for the transform (there is no CPU path): %s",
                                cmhip_last_error());
            return -1;
        }
        dirty = 1;
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
    cmhip_batch_vu_pause(t->dev, !(t->fused_vu && t->vu_armed));
    while (frames) {
        const size_t n = frames < TRANSFORM_SLICE_FRAMES ? frames : TRANSFORM_SLICE_FRAMES;
        if (cmhip_batch_upload(t->dev, 0, pcm, n) != COOLMIC_ERROR_NONE ||
            cmhip_batch_run(t->dev, n, NULL) != COOLMIC_ERROR_NONE ||
            cmhip_batch_download(t->dev, 0, pcm, n) != COOLMIC_ERROR_NONE) {
            coolmic_logging_log(COOLMIC_LOGGING_LEVEL_ERROR, COOLMIC_ERROR_GENERIC,
                                "HIP transform failed: %s", cmhip_last_error());
            return -1;
        }
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

    if (have && transform_process(t, buffer, have / framesize) != 0)
        return -1;
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
    return coolmic_ro_unref(userdata);
}

/* ---- the fused VU window (internal: vumeter.c; declared in host_internal.h) ------------------ */

/* A meter attached directly to this transform's handle, with the same rate and channel count,
 * sees exactly the frames this transform returns (ref: src/simple.c:212-229 wires them through a
 * tee, config 1 of BASELINE.json directly).  Then one launch does both loops of the reference --
 * __process (ref: src/transform.c:101-124) and the accumulate loop (ref: src/vumeter.c:161-177).
 * One meter at a time; the window starts empty. */
int coolmic_transform_fuse_vu(coolmic_transform_t *self, int on)
{
    if (self == NULL)
        return COOLMIC_ERROR_FAULT;
    if (on && self->fused_vu)
        return COOLMIC_ERROR_BUSY;
    self->fused_vu = on ? 1 : 0;
    self->vu_armed = 0;
    if (self->dev != NULL && cmhip_batch_vu_reset(self->dev, 0) != COOLMIC_ERROR_NONE)
        return COOLMIC_ERROR_GENERIC;
    return COOLMIC_ERROR_NONE;
}

void coolmic_transform_arm_vu(coolmic_transform_t *self, int armed)
{
    if (self != NULL)
        self->vu_armed = armed ? 1 : 0;
}

int coolmic_transform_vu_result(coolmic_transform_t *self, coolmic_vumeter_result_t *result)
{
    if (self == NULL || result == NULL)
        return COOLMIC_ERROR_FAULT;
    if (self->dev == NULL)
        return COOLMIC_ERROR_INVAL;    /* no frame has passed yet (ref: src/vumeter.c:198-199) */
    return cmhip_batch_vu_result(self->dev, 0, result);
}

int coolmic_transform_vu_reset(coolmic_transform_t *self)
{
    if (self == NULL)
        return COOLMIC_ERROR_FAULT;
    if (self->dev != NULL && cmhip_batch_vu_reset(self->dev, 0) != COOLMIC_ERROR_NONE)
        return COOLMIC_ERROR_GENERIC;
    return COOLMIC_ERROR_NONE;
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
