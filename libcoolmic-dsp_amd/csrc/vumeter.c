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
 * Where the frames are accumulated depends on what the meter is attached to:
 *
 *   DIRECT   the handle of a transform (same rate, same channel count, nothing buffered here): the
 *            meter sees exactly the frames that transform returns, and the transform's launch
 *            accumulates the window beside its own arithmetic (transform.c,
 *            coolmic_transform_fuse_vu): one launch per pull instead of two.
 *   RECORDS  a reader handle of a coolmic_tee_t whose upstream is such a transform -- the product's
 *            wiring (ref: src/simple.c:212-229).  The meter may lag behind what the transform has
 *            produced (the tee buffers up to 8 KiB for its slower reader), so the transform's one
 *            window is not the meter's.  Instead every launch of the transform leaves a window
 *            RECORD of its own (its bytes in the transform's output, its sums and peak keys), the tee
 *            says where this reader is in that output, and the meter merges -- integer adds and
 *            maxima per launch, in stream order, which is the reference's strict-greater update
 *            (ref: src/vumeter.c:163-168) -- the records of exactly the bytes it has consumed.  Only
 *            where a window boundary (a result() or reset()) falls INSIDE a launch's block do the two
 *            parts of that block go through a launch of the meter's own (it holds the bytes: they
 *            came through its buffer).  With 1024-byte pulls on both branches that never happens.
 *   OWN      anything else: a one-stream batch of its own, one upload + launch per read.
 */
#define COOLMIC_COMPONENT "libcoolmic-dsp/vumeter"
#include "host_internal.h"
#include <coolmic-dsp/vumeter.h>
#include <coolmic_hip.h>

#include <stdlib.h>
#include <string.h>

#define VUMETER_BUFFER (2 * COOLMIC_DSP_VUMETER_MAX_CHANNELS * 32)    /* 1024 bytes */
#define VUMETER_FOLD_RECORDS 512      /* whole records waiting in a window before they are folded into it */

enum { METER_OWN = 0, METER_DIRECT = 1, METER_RECORDS = 2 };

struct coolmic_vumeter {
    igloo_ro_base_t __base;
    coolmic_iohandle_t *in;
    uint_least32_t rate;
    unsigned int channels;
    unsigned char buffer[VUMETER_BUFFER];
    size_t fill;
    cmhip_batch_t *dev;                /* one stream, VU only */
    int device_plus1;                  /* coolmic_vumeter_set_device(): GPU + 1, 0 = the process's default */
    int mode;
    struct coolmic_transform *fused;   /* DIRECT / RECORDS: the upstream transform that accumulates for us
                                        * (kept alive by the handle `in`, which holds a reference) */
    /* RECORDS (positions are bytes of the transform's output) */
    void *reader;                      /* the tee's reader behind `in` */
    unsigned int discont;              /* the tee's discontinuity count when the mapping was taken */
    uint64_t apos;                     /* everything before it has been accounted to a window */
    uint64_t merged;                   /* whole records before it have been merged into `acc` */
    uint64_t piece_end;                /* bytes [.., piece_end) go through a launch of our own: the rest of a
                                        * block that a window boundary cut, or what was in the tee before the
                                        * records began (0: none) */
    unsigned int pending;              /* whole records in [merged, ..) not merged yet */
    unsigned char *held;               /* the bytes consumed of the block under way (or of the piece) */
    size_t held_len, held_cap;
    cmhip_vu_raw_t acc;                /* the window so far, as far as it has been merged on the host */
};

static void meter_unfuse(coolmic_vumeter_t *v)
{
    if (v->mode == METER_DIRECT) {
        coolmic_transform_fuse_vu(v->fused, 0);
    } else if (v->mode == METER_RECORDS) {
        coolmic_transform_records(v->fused, 0);
        coolmic_ro_unref(v->fused);    /* (the reference taken in meter_look_upstream) */
    }
    v->mode = METER_OWN;
    v->fused = NULL;
    v->reader = NULL;
}

static void vumeter_destroy(void *self)
{
    coolmic_vumeter_t *v = self;
    meter_unfuse(v);
    coolmic_ro_unref(v->in);
    cmhip_batch_free(v->dev);
    free(v->held);
}

COOLMIC_RO_TYPE(coolmic_vumeter_t, vumeter_destroy);

coolmic_vumeter_t *coolmic_vumeter_new(const char *name, igloo_ro_t associated,
                                       uint_least32_t rate, unsigned int channels)
{
    coolmic_vumeter_t *v;

    if (!rate || !channels || channels > COOLMIC_DSP_VUMETER_MAX_CHANNELS)
        return NULL;
    v = COOLMIC_RO_NEW(coolmic_vumeter_t, name, associated);
    if (v == NULL)
        return NULL;
    v->rate = rate;
    v->channels = channels;
    return v;
}

int coolmic_vumeter_set_device(coolmic_vumeter_t *self, int device)
{
    int rc;

    if (self == NULL)
        return COOLMIC_ERROR_FAULT;
    rc = coolmic_hip_check_device(device);
    if (rc != COOLMIC_ERROR_NONE)
        return rc;
    if (self->dev != NULL)
        return COOLMIC_ERROR_BUSY;     /* the window so far is on the GPU the meter started on */
    self->device_plus1 = device + 1;
    return COOLMIC_ERROR_NONE;
}

/* ---- a launch of our own --------------------------------------------------------------------------- */

static int meter_device(coolmic_vumeter_t *v)
{
    cmhip_batch_desc_t d;

    if (v->dev != NULL)
        return 0;
    memset(&d, 0, sizeof(d));
    d.device = coolmic_hip_stage_device(v->device_plus1);
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
    return 0;
}

/* whole frames into the window of our own batch */
static int vumeter_account(coolmic_vumeter_t *v, const unsigned char *bytes, size_t frames)
{
    const size_t cap = VUMETER_BUFFER / 2, framesize = 2u * v->channels;

    if (meter_device(v) != 0)
        return -1;
    while (frames) {
        const size_t n = frames < cap ? frames : cap;
        if (cmhip_batch_upload(v->dev, 0, (const int16_t *)bytes, n) != COOLMIC_ERROR_NONE ||
            cmhip_batch_run(v->dev, n, NULL) != COOLMIC_ERROR_NONE) {
            coolmic_logging_log(COOLMIC_LOGGING_LEVEL_ERROR, COOLMIC_ERROR_GENERIC,
                                "HIP VU accumulation failed: %s", cmhip_last_error());
            return -1;
        }
        bytes += n * framesize;
        frames -= n;
    }
    return 0;
}

/* RECORDS: the held bytes are a piece of the window that no record covers on its own -- through our own
 * batch (an empty window of it), and its result behind what `acc` holds */
static int meter_piece(coolmic_vumeter_t *v)
{
    cmhip_vu_raw_t raw;

    if (v->held_len == 0)
        return 0;
    if (vumeter_account(v, v->held, v->held_len / (2u * v->channels)) != 0 ||
        cmhip_batch_vu_raw_state(v->dev, 0, &raw) != COOLMIC_ERROR_NONE ||
        cmhip_batch_vu_reset(v->dev, 0) != COOLMIC_ERROR_NONE)
        return -1;
    cmhip_vu_raw_merge(&v->acc, &raw, v->channels);
    v->held_len = 0;
    return 0;
}

static int meter_hold(coolmic_vumeter_t *v, const unsigned char *bytes, size_t n)
{
    if (v->held_len + n > v->held_cap) {
        size_t cap = v->held_cap ? v->held_cap : 4096;
        unsigned char *grown;
        while (cap < v->held_len + n)
            cap *= 2;
        grown = realloc(v->held, cap);
        if (grown == NULL)
            return -1;
        v->held = grown;
        v->held_cap = cap;
    }
    memcpy(v->held + v->held_len, bytes, n);
    v->held_len += n;
    return 0;
}

/* RECORDS: whole records up to `to` (a record boundary) into `acc` */
static int meter_merge_upto(coolmic_vumeter_t *v, uint64_t to)
{
    if (to > v->merged) {
        if (coolmic_transform_records_merge(v->fused, v->merged, to, &v->acc) != COOLMIC_ERROR_NONE)
            return -1;
        v->merged = to;
    }
    v->pending = 0;
    return 0;
}

/* RECORDS: `n` bytes of whole frames at `apos` have been consumed */
static int meter_consume(coolmic_vumeter_t *v, const unsigned char *bytes, size_t n)
{
    while (n) {
        uint64_t off;
        uint32_t len;
        size_t take;

        if (v->apos < v->piece_end) {              /* a piece for our own launch, until piece_end */
            take = v->piece_end - v->apos < n ? (size_t)(v->piece_end - v->apos) : n;
            if (meter_hold(v, bytes, take) != 0)
                return -1;
            v->apos += take;
            if (v->apos == v->piece_end) {         /* complete: in front of every record that follows */
                if (meter_piece(v) != 0)
                    return -1;
                v->piece_end = 0;
                v->merged = v->apos;
            }
        } else {
            if (coolmic_transform_record_at(v->fused, v->apos, &off, &len) != COOLMIC_ERROR_NONE)
                return -1;                         /* bytes no launch of the transform accounts for */
            take = off + len - v->apos < n ? (size_t)(off + len - v->apos) : n;
            if (v->apos == off && take == len) {
                v->pending++;                      /* a whole block at once: nothing to keep */
            } else {
                if (meter_hold(v, bytes, take) != 0)
                    return -1;
                if (v->apos + take == off + len) { /* the block is complete: its record will do */
                    v->held_len = 0;
                    v->pending++;
                }
            }
            v->apos += take;
        }
        bytes += take;
        n -= take;
    }
    if (v->pending >= VUMETER_FOLD_RECORDS)        /* a long window: fold what is complete */
        return meter_merge_upto(v, v->apos - v->held_len);
    return 0;
}

/* RECORDS: everything accounted so far is in `acc` afterwards; a block under way is cut at apos */
static int meter_close_window(coolmic_vumeter_t *v)
{
    if (v->apos < v->piece_end || v->piece_end != 0) {
        /* inside a piece: what is held is all there is of this window since the piece began */
        return meter_piece(v);
    }
    if (meter_merge_upto(v, v->apos - v->held_len) != 0)
        return -1;
    if (v->held_len) {                             /* the boundary falls inside a block: both parts are ours */
        uint64_t off;
        uint32_t len;
        if (coolmic_transform_record_at(v->fused, v->apos - 1, &off, &len) != COOLMIC_ERROR_NONE || meter_piece(v) != 0)
            return -1;
        v->piece_end = off + len;
    }
    v->merged = v->apos;
    return 0;
}

/* RECORDS -> OWN, keeping what the window holds: after a discontinuity (somebody else read the transform's
 * handle, the tee got another upstream) or an error of the shared path */
static int meter_fall_back(coolmic_vumeter_t *v)
{
    int rc = meter_close_window(v);
    coolmic_logging_log(COOLMIC_LOGGING_LEVEL_DEBUG, COOLMIC_ERROR_NONE,
                        "VU meter behind a tee: back to a batch of its own (rc %d)", rc);
    meter_unfuse(v);
    v->piece_end = 0;
    v->held_len = 0;
    v->pending = 0;
    return rc;
}

/* ---- attaching --------------------------------------------------------------------------------------- */

/* what is above us?  (called when a handle is attached and, behind a tee that had no transform above it
 * yet, before a read) */
static void meter_look_upstream(coolmic_vumeter_t *self)
{
    struct coolmic_transform *t;
    uint_least32_t rate;
    unsigned int channels, discont = 0;
    uint64_t next = 0;
    void *reader;

    /* the frames must be ours alone: no bytes of another source waiting here, no frames in a window of
     * our own that the next result would have to merge */
    if (self->mode != METER_OWN || self->fill != 0 || self->dev != NULL || self->acc.samples != 0)
        return;
    t = coolmic_iohandle_as_transform(self->in);
    if (t != NULL) {
        coolmic_transform_format(t, &rate, &channels);
        if (rate == self->rate && channels == self->channels &&
            coolmic_transform_fuse_vu(t, 1) == COOLMIC_ERROR_NONE) {
            self->fused = t;
            self->mode = METER_DIRECT;
        }
        return;
    }
    reader = coolmic_iohandle_as_tee_reader(self->in);
    if (reader == NULL)
        return;
    t = coolmic_tee_reader_upstream(reader, &next, &discont);
    if (t == NULL)
        return;
    coolmic_transform_format(t, &rate, &channels);
    if (rate != self->rate || channels != self->channels || next % (2u * channels) != 0 ||
        coolmic_transform_records(t, 1) != COOLMIC_ERROR_NONE)
        return;
    /* (behind a tee the transform is held by the TEE's handle, which may be exchanged under us: a
     * reference of our own while we point at it) */
    coolmic_ro_ref(t);
    self->fused = t;
    self->mode = METER_RECORDS;
    self->reader = reader;
    self->discont = discont;
    self->apos = self->merged = next;
    /* what the tee holds from before this moment has no records: a piece for our own launch */
    self->piece_end = coolmic_transform_records_start(t) > next ? coolmic_transform_records_start(t) : 0;
    self->held_len = 0;
    self->pending = 0;
    memset(&self->acc, 0, sizeof(self->acc));
}

int coolmic_vumeter_attach_iohandle(coolmic_vumeter_t *self, coolmic_iohandle_t *handle)
{
    if (self == NULL)
        return COOLMIC_ERROR_FAULT;
    /* the frames accounted so far stay in the window, as in the reference (ref: src/vumeter.c:101-110
     * touches nothing but the handle) */
    if (self->mode == METER_RECORDS) {
        meter_fall_back(self);
    } else if (self->mode == METER_DIRECT) {
        cmhip_vu_raw_t raw;
        if (coolmic_transform_vu_take_raw(self->fused, &raw) == COOLMIC_ERROR_NONE)
            cmhip_vu_raw_merge(&self->acc, &raw, self->channels);
        meter_unfuse(self);
    }
    coolmic_ro_unref(self->in);
    self->in = handle;
    coolmic_ro_ref(handle);
    meter_look_upstream(self);
    return COOLMIC_ERROR_NONE;
}

int coolmic_vumeter_reset(coolmic_vumeter_t *self)
{
    if (self == NULL)
        return COOLMIC_ERROR_FAULT;
    /* like the reference this leaves a buffered partial frame alone */
    if (self->mode == METER_DIRECT)
        return coolmic_transform_vu_reset(self->fused);
    if (self->mode == METER_RECORDS) { /* the window closes here, and what it held is dropped */
        if (meter_close_window(self) != 0)
            return COOLMIC_ERROR_GENERIC;
        coolmic_transform_records_drop(self->fused, self->merged);
    }
    memset(&self->acc, 0, sizeof(self->acc));
    if (self->dev != NULL && cmhip_batch_vu_reset(self->dev, 0) != COOLMIC_ERROR_NONE)
        return COOLMIC_ERROR_GENERIC;
    return COOLMIC_ERROR_NONE;
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

    if (self->mode == METER_OWN && self->dev == NULL)
        meter_look_upstream(self);     /* (a tee that got its transform after we got the tee) */

    want = sizeof(self->buffer) - self->fill;
    if (maxlen >= 0 && want > (size_t)maxlen)
        want = (size_t)maxlen;
    if (self->mode == METER_DIRECT)
        coolmic_transform_arm_vu(self->fused, 1);      /* the frames of this read are ours */
    got = coolmic_iohandle_read(self->in, self->buffer + self->fill, want);
    if (self->mode == METER_DIRECT)
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
    used = frames * framesize;
    if (frames && self->mode == METER_RECORDS) {
        /* the bytes are where we think they are only while the tee's mapping has held (a pull inside the
         * read above may just have broken it) */
        const uint64_t before = self->apos;
        if (coolmic_tee_reader_discont(self->reader) != self->discont || meter_consume(self, self->buffer, used) != 0) {
            /* what has been accounted through records stays accounted; the rest of this read, and every
             * read from now on, through our own batch */
            const size_t done = (size_t)(self->apos - before);
            if (meter_fall_back(self) != 0 ||
                (done < used && vumeter_account(self, self->buffer + done, (used - done) / framesize) != 0))
                return -1;
        }
    } else if (frames && self->mode == METER_OWN) {
        if (vumeter_account(self, self->buffer, frames) != 0)
            return -1;
    }
    /* (DIRECT: the transform hands over whole frames and has accumulated them in its own launch) */

    if (used < self->fill)
        memmove(self->buffer, self->buffer + used, self->fill - used);
    self->fill -= used;
    return ret;
}

int coolmic_vumeter_result(coolmic_vumeter_t *self, coolmic_vumeter_result_t *result)
{
    cmhip_vu_raw_t own;
    int rc;

    if (self == NULL || result == NULL)
        return COOLMIC_ERROR_FAULT;
    if (self->mode == METER_DIRECT) {
        rc = coolmic_transform_vu_result(self->fused, result);
        if (rc != COOLMIC_ERROR_NONE && rc != COOLMIC_ERROR_INVAL)
            coolmic_logging_log(COOLMIC_LOGGING_LEVEL_ERROR, rc, "HIP VU result failed: %s",
                                cmhip_last_error());
        return rc;
    }
    if (self->mode == METER_RECORDS && meter_close_window(self) != 0) {
        coolmic_logging_log(COOLMIC_LOGGING_LEVEL_ERROR, COOLMIC_ERROR_GENERIC, "HIP VU result failed: %s",
                            cmhip_last_error());
        return COOLMIC_ERROR_GENERIC;
    }
    if (self->acc.samples == 0) {      /* nothing merged on the host: the window is our own batch's */
        if (self->dev == NULL)
            return COOLMIC_ERROR_INVAL;    /* no frame was ever accounted */
        rc = cmhip_batch_vu_result(self->dev, 0, result);
        if (rc != COOLMIC_ERROR_NONE && rc != COOLMIC_ERROR_INVAL)
            coolmic_logging_log(COOLMIC_LOGGING_LEVEL_ERROR, rc, "HIP VU result failed: %s",
                                cmhip_last_error());
        return rc;
    }
    /* merged records (and pieces) first, then what our own batch's window holds (frames accounted after a
     * fall-back): stream order */
    if (self->dev != NULL) {
        if (cmhip_batch_vu_raw_state(self->dev, 0, &own) != COOLMIC_ERROR_NONE ||
            cmhip_batch_vu_reset(self->dev, 0) != COOLMIC_ERROR_NONE)
            return COOLMIC_ERROR_GENERIC;
        cmhip_vu_raw_merge(&self->acc, &own, self->channels);
    }
    rc = cmhip_vu_raw_finish(&self->acc, self->channels, (unsigned int)self->rate, result);
    memset(&self->acc, 0, sizeof(self->acc));
    return rc;
}

/* test hook (not in a public header): 0 a batch of its own, 1 sharing the launch of a transform right above,
 * 2 sharing it through a tee by window records */
int coolmic_debug_vumeter_mode(const coolmic_vumeter_t *self)
{
    return self ? self->mode : -1;
}
