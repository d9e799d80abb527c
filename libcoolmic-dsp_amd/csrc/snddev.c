/* snddev.c -- the sources "null", "sine" and "stdio" behind the reference's snddev API
 * (contract: <coolmic-dsp/snddev.h>; ref: src/snddev.c:60-215, src/snddev_null.c,
 * src/snddev_sine.c:101-193, src/snddev_stdio.c:50-78): capture handle, and the playback side
 * (attach a handle, iterate) that "null" / "sine" discard and "stdio" writes to its file.
 * Data sources and sinks for the chain; they stay on the CPU.  Hardware drivers are the
 * reference's own. */
#define COOLMIC_COMPONENT "libcoolmic-dsp/snddev"
#include "host_internal.h"
#include <coolmic-dsp/snddev.h>

#include <stdio.h>
#include <string.h>
#include <strings.h>

enum source_kind { SOURCE_NULL, SOURCE_SINE, SOURCE_STDIO };

struct coolmic_snddev {
    igloo_ro_base_t __base;
    enum source_kind kind;
    /* sine: one period and the byte position inside it */
    int16_t period[96];
    size_t period_bytes;
    size_t phase;
    /* stdio: raw PCM file being replayed or written */
    FILE *file;
    /* playback: handle whose PCM goes to the device, and what the device has not taken yet */
    coolmic_iohandle_t *tx;
    unsigned char txbuffer[1024];
    size_t txfill;
};

static void snddev_destroy(void *self)
{
    coolmic_snddev_t *dev = self;
    coolmic_ro_unref(dev->tx);
    if (dev->file != NULL)
        fclose(dev->file);
}

COOLMIC_RO_TYPE(coolmic_snddev_t, snddev_destroy);

static ssize_t snddev_read(void *userdata, void *buffer, size_t len)
{
    coolmic_snddev_t *dev = userdata;

    coolmic_logging_log(COOLMIC_LOGGING_LEVEL_DEBUG, COOLMIC_ERROR_NONE,
                        "Read request, buffer=%p, len=%zu", buffer, len);
    if (dev->kind == SOURCE_NULL) {
        memset(buffer, 0, len);                  /* silence, always the full request */
    } else if (dev->kind == SOURCE_STDIO) {
        return (ssize_t)fread(buffer, 1, len, dev->file);   /* 0 at end of file (or a file opened for writing only) */
    } else {
        /* endless repetition of the period, byte granular: a read may stop in the
         * middle of a sample and the next one continues there */
        const unsigned char *src = (const unsigned char *)dev->period;
        unsigned char *dst = buffer;
        size_t left = len, pos = dev->phase;
        while (left) {
            size_t run = dev->period_bytes - pos;
            if (run > left)
                run = left;
            memcpy(dst, src + pos, run);
            dst += run;
            left -= run;
            pos += run;
            if (pos == dev->period_bytes)
                pos = 0;
        }
        dev->phase = pos;
    }
    return (ssize_t)len;
}

static int snddev_handle_free(void *userdata)
{
    coolmic_ro_unref(userdata);
    return 0;
}

coolmic_snddev_t *coolmic_snddev_new(const char *name, igloo_ro_t associated, const char *driver,
                                     void *device, uint_least32_t rate, unsigned int channels,
                                     int flags, ssize_t buffer)
{
    coolmic_snddev_t *dev;
    enum source_kind kind;
    int16_t period[96];
    size_t n = 0;
    FILE *file = NULL;

    (void)buffer;
    if (!rate || !channels || !flags)
        return NULL;
    if (driver == NULL)                          /* AUTO: no hardware here, so "null" */
        driver = COOLMIC_DSP_SNDDEV_DRIVER_NULL;
    if (strcasecmp(driver, COOLMIC_DSP_SNDDEV_DRIVER_NULL) == 0) {
        kind = SOURCE_NULL;
    } else if (strcasecmp(driver, COOLMIC_DSP_SNDDEV_DRIVER_SINE) == 0) {
        kind = SOURCE_SINE;
        if (channels != 1 || coolmic_sine_period(rate, period, &n) != COOLMIC_ERROR_NONE)
            return NULL;                         /* mono, table rates only */
    } else if (strcasecmp(driver, COOLMIC_DSP_SNDDEV_DRIVER_STDIO) == 0) {
        /* raw PCM file: `device` is its name; read, written, or both by the flags
         * (ref: src/snddev_stdio.c:50-78) */
        const int rxtx = flags & COOLMIC_DSP_SNDDEV_RXTX;
        kind = SOURCE_STDIO;
        if (device == NULL || *(const char *)device == 0 || rxtx == 0)
            return NULL;
        file = fopen(device, rxtx == COOLMIC_DSP_SNDDEV_RXTX ? "w+b" : (rxtx & COOLMIC_DSP_SNDDEV_RX) ? "rb" : "wb");
        if (file == NULL)
            return NULL;
    } else {
        return NULL;
    }

    dev = COOLMIC_RO_NEW(coolmic_snddev_t, name, associated);
    if (dev == NULL) {
        if (file != NULL)
            fclose(file);
        return NULL;
    }
    dev->kind = kind;
    dev->file = file;
    if (kind == SOURCE_SINE) {
        memcpy(dev->period, period, n * sizeof(int16_t));
        dev->period_bytes = n * sizeof(int16_t);
    }
    return dev;
}

coolmic_iohandle_t *coolmic_snddev_get_iohandle(coolmic_snddev_t *self)
{
    coolmic_iohandle_t *h;

    if (self == NULL)
        return NULL;
    coolmic_ro_ref(self);
    h = coolmic_iohandle_new(NULL, igloo_RO_NULL, self, snddev_handle_free, snddev_read, NULL);
    if (h == NULL)
        coolmic_ro_unref(self);
    return h;
}

/* ---- playback side (ref: src/snddev.c:143-152, 171-215) ------------------------------------------ */

int coolmic_snddev_attach_iohandle(coolmic_snddev_t *self, coolmic_iohandle_t *handle)
{
    if (self == NULL)
        return COOLMIC_ERROR_FAULT;
    coolmic_ro_unref(self->tx);
    self->tx = handle;
    coolmic_ro_ref(handle);            /* NULL detaches */
    return COOLMIC_ERROR_NONE;
}

/* what the device takes of `len` bytes: everything for null / sine (discarded), what fwrite takes for stdio */
static ssize_t snddev_write(coolmic_snddev_t *dev, const void *buffer, size_t len)
{
    if (dev->kind == SOURCE_STDIO)
        return (ssize_t)fwrite(buffer, 1, len, dev->file);
    return (ssize_t)len;
}

/* NONE: nothing left in the buffer; BUSY: the device took none or part of it; GENERIC: it failed */
static int snddev_flush(coolmic_snddev_t *dev)
{
    ssize_t took;

    if (dev->txfill == 0)
        return COOLMIC_ERROR_NONE;
    took = snddev_write(dev, dev->txbuffer, dev->txfill);
    if (took < 0)
        return COOLMIC_ERROR_GENERIC;
    if ((size_t)took < dev->txfill) {
        memmove(dev->txbuffer, dev->txbuffer + took, dev->txfill - (size_t)took);
        dev->txfill -= (size_t)took;
        return COOLMIC_ERROR_BUSY;
    }
    dev->txfill = 0;
    return COOLMIC_ERROR_NONE;
}

/* one round of playback: what is still buffered first, then up to 1 KiB more from the attached handle */
int coolmic_snddev_iter(coolmic_snddev_t *self)
{
    ssize_t got;
    int rc;

    if (self == NULL)
        return COOLMIC_ERROR_FAULT;
    rc = snddev_flush(self);
    if (rc != COOLMIC_ERROR_NONE)
        return rc;
    got = coolmic_iohandle_read(self->tx, self->txbuffer, sizeof(self->txbuffer));
    if (got < 0)
        return COOLMIC_ERROR_GENERIC;  /* (also without a handle attached: the read reports FAULT) */
    if (got == 0)
        return COOLMIC_ERROR_NONE;
    self->txfill = (size_t)got;
    return snddev_flush(self);
}
