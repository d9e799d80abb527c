/* vumeter.c -- per-stream VU meter in front of the HIP engine
 * (contract: <coolmic-dsp/vumeter.h>; ref: src/vumeter.c).
 *
 * The host does what the reference does around its loop: pull up to 1024 bytes
 * into a staging buffer, hand the whole frames on, keep the partial frame
 * (ref: src/vumeter.c:112-136,179-184).  The loop itself -- first-max peak and sum
 * of squares per channel (ref: src/vumeter.c:161-177) -- runs on the GPU; the
 * window lives in device memory until coolmic_vumeter_result() fetches it and
 * finishes the dB values in double.  Nothing is accumulated on the CPU.
 *
 * Attached DIRECTLY to a transform's handle (same rate, same channel count, nothing buffered
 * here) the meter sees exactly the frames that transform returns, and the transform's launch
 * accumulates the window beside its own arithmetic (transform.c, coolmic_transform_fuse_vu):
 * one launch per pull instead of two, 24.7 -> ~12 us per 1 KiB.  Through a tee the meter may
 * lag behind what the transform has produced, so its window is not the transform's: there it
 * keeps its own batch.
 */
#define COOLMIC_COMPONENT "libcoolmic-dsp/vumeter"
#include "host_internal.h"
#include <coolmic-dsp/vumeter.h>
#include <coolmic_hip.h>

#include <stdlib.h>
#include <string.h>

#define VUMETER_BUFFER (2 * COOLMIC_DSP_VUMETER_MAX_CHANNELS * 32)    /* 1024 bytes */

struct coolmic_vumeter {
    coolmic_ro_base_t base;
    coolmic_iohandle_t *in;
    uint_least32_t rate;
    unsigned int channels;
    unsigned char buffer[VUMETER_BUFFER];
    size_t fill;
    cmhip_batch_t *dev;                /* one stream, VU only */
    struct coolmic_transform *fused;   /* upstream transform that keeps the window for us, or NULL
                                        * (kept alive by the handle `in`, which holds a reference) */
};

static void vumeter_destroy(void *self)
{
    coolmic_vumeter_t *v = self;
    if (v->fused != NULL)
        coolmic_transform_fuse_vu(v->fused, 0);
    coolmic_ro_unref(v->in);
    cmhip_batch_free(v->dev);
}

static const coolmic_ro_type_t vumeter_type = {
    "coolmic_vumeter_t", sizeof(coolmic_vumeter_t), vumeter_destroy
};

coolmic_vumeter_t *coolmic_vumeter_new(const char *name, igloo_ro_t associated,
                                       uint_least32_t rate, unsigned int channels)
{
    coolmic_vumeter_t *v;

    if (!rate || !channels || channels > COOLMIC_DSP_VUMETER_MAX_CHANNELS)
        return NULL;
    v = coolmic_ro_new_raw(&vumeter_type, name, associated);
    if (v == NULL)
        return NULL;
    v->rate = rate;
    v->channels = channels;
    return v;
}

int coolmic_vumeter_reset(coolmic_vumeter_t *self)
{
    if (self == NULL)
        return COOLMIC_ERROR_FAULT;
    /* like the reference this leaves a buffered partial frame alone */
    if (self->fused != NULL)
        return coolmic_transform_vu_reset(self->fused);
    if (self->dev != NULL && cmhip_batch_vu_reset(self->dev, 0) != COOLMIC_ERROR_NONE)
        return COOLMIC_ERROR_GENERIC;
    return COOLMIC_ERROR_NONE;
}

int coolmic_vumeter_attach_iohandle(coolmic_vumeter_t *self, coolmic_iohandle_t *handle)
{
    struct coolmic_transform *t;

    if (self == NULL)
        return COOLMIC_ERROR_FAULT;
    if (self->fused != NULL) {         /* the window kept upstream ends with the attachment */
        coolmic_transform_fuse_vu(self->fused, 0);
        self->fused = NULL;
    }
    coolmic_ro_unref(self->in);
    self->in = handle;
    coolmic_ro_ref(handle);

    /* a transform right above us, same format, no bytes of another source waiting here, and no
     * frames in a window of our own that the next result would have to merge */
    t = coolmic_iohandle_as_transform(handle);
    if (t != NULL && self->fill == 0 && self->dev == NULL) {
        uint_least32_t rate;
        unsigned int channels;
        coolmic_transform_format(t, &rate, &channels);
        if (rate == self->rate && channels == self->channels &&
            coolmic_transform_fuse_vu(t, 1) == COOLMIC_ERROR_NONE)
            self->fused = t;
    }
    return COOLMIC_ERROR_NONE;
}

static int vumeter_account(coolmic_vumeter_t *v, size_t frames)
{
    if (v->dev == NULL) {
        cmhip_batch_desc_t d;
        memset(&d, 0, sizeof(d));
        d.device = coolmic_hip_default_device();
        d.streams = 1;
        d.channels = v->channels;
        d.rate = (unsigned int)v->rate;
        d.max_frames = VUMETER_BUFFER / 2;
        d.flags = CMHIP_VU;
        v->dev = cmhip_batch_new(&d);
        if (v->dev == NULL) {
            coolmic_logging_log(COOLMIC_LOGGING_LEVEL_ERROR, COOLMIC_ERROR_NOSYS,
                                "no HIP engine for the VU meter (there is no CPU path): %s",
                                cmhip_last_error());
            return -1;
        }
    }
    if (cmhip_batch_upload(v->dev, 0, (const int16_t *)v->buffer, frames) != COOLMIC_ERROR_NONE ||
        cmhip_batch_run(v->dev, frames, NULL) != COOLMIC_ERROR_NONE) {
        coolmic_logging_log(COOLMIC_LOGGING_LEVEL_ERROR, COOLMIC_ERROR_GENERIC,
                            "HIP VU accumulation failed: %s", cmhip_last_error());
        return -1;
    }
    return 0;
}

ssize_t coolmic_vumeter_read(coolmic_vumeter_t *self, ssize_t maxlen)
{
    size_t want, framesize, frames, used;
    ssize_t got, ret;

    coolmic_logging_log(COOLMIC_LOGGING_LEVEL_DEBUG, COOLMIC_ERROR_NONE, "Read request, maxlen=%zi",
                        maxlen);
    if (self == NULL) {
        coolmic_logging_log(COOLMIC_LOGGING_LEVEL_ERROR, COOLMIC_ERROR_FAULT, "Bad state, self=NULL");
        return -1;
    }

    want = sizeof(self->buffer) - self->fill;
    if (maxlen >= 0 && want > (size_t)maxlen)
        want = (size_t)maxlen;
    if (self->fused != NULL)
        coolmic_transform_arm_vu(self->fused, 1);      /* the frames of this read are ours */
    got = coolmic_iohandle_read(self->in, self->buffer + self->fill, want);
    if (self->fused != NULL)
        coolmic_transform_arm_vu(self->fused, 0);
    coolmic_logging_log(COOLMIC_LOGGING_LEVEL_DEBUG, COOLMIC_ERROR_NONE,
                        "Physical read on iohandle returned %zi bytes", got);
    if (got < 0) {
        /* an upstream error only surfaces when nothing is buffered; the reference
         * tests for -1 alone (ref: src/vumeter.c:127-131) and would add any other
         * negative code to its fill counter -- that slip is not reproduced */
        ret = self->fill ? 0 : -1;
    } else {
        self->fill += (size_t)got;
        ret = got;
    }

    framesize = 2u * self->channels;
    frames = self->fill / framesize;
    /* fused: the transform hands over whole frames and has accumulated them in its own launch */
    if (frames && self->fused == NULL && vumeter_account(self, frames) != 0)
        return -1;

    used = frames * framesize;
    if (used < self->fill)
        memmove(self->buffer, self->buffer + used, self->fill - used);
    self->fill -= used;
    return ret;
}

int coolmic_vumeter_result(coolmic_vumeter_t *self, coolmic_vumeter_result_t *result)
{
    int rc;

    if (self == NULL || result == NULL)
        return COOLMIC_ERROR_FAULT;
    if (self->fused != NULL) {
        rc = coolmic_transform_vu_result(self->fused, result);
        if (rc != COOLMIC_ERROR_NONE && rc != COOLMIC_ERROR_INVAL)
            coolmic_logging_log(COOLMIC_LOGGING_LEVEL_ERROR, rc, "HIP VU result failed: %s",
                                cmhip_last_error());
        return rc;
    }
    if (self->dev == NULL)
        return COOLMIC_ERROR_INVAL;    /* no frame was ever accounted */
    rc = cmhip_batch_vu_result(self->dev, 0, result);
    if (rc != COOLMIC_ERROR_NONE && rc != COOLMIC_ERROR_INVAL)
        coolmic_logging_log(COOLMIC_LOGGING_LEVEL_ERROR, rc, "HIP VU result failed: %s",
                            cmhip_last_error());
    return rc;
}
