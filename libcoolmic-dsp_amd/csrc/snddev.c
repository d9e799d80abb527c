/* snddev.c -- capture sources "null" and "sine" behind the reference's snddev API
 * (contract: <coolmic-dsp/snddev.h>; ref: src/snddev.c:60-169, src/snddev_null.c,
 * src/snddev_sine.c:101-193, src/snddev_stdio.c:50-78).  These are data sources for the
 * chain and stay on the CPU; hardware drivers and playback are out of scope. */
#define COOLMIC_COMPONENT "libcoolmic-dsp/snddev"
#include "host_internal.h"
#include <coolmic-dsp/snddev.h>

#include <stdio.h>
#include <string.h>
#include <strings.h>

enum source_kind { SOURCE_NULL, SOURCE_SINE, SOURCE_STDIO };

struct coolmic_snddev {
    coolmic_ro_base_t base;
    enum source_kind kind;
    /* sine: one period and the byte position inside it */
    int16_t period[96];
    size_t period_bytes;
    size_t phase;
    /* stdio: raw PCM file being replayed */
    FILE *file;
};

static void snddev_destroy(void *self)
{
    coolmic_snddev_t *dev = self;
    if (dev->file != NULL)
        fclose(dev->file);
}

static const coolmic_ro_type_t snddev_type = {"coolmic_snddev_t", sizeof(coolmic_snddev_t), snddev_destroy};

static ssize_t snddev_read(void *userdata, void *buffer, size_t len)
{
    coolmic_snddev_t *dev = userdata;

    coolmic_logging_log(COOLMIC_LOGGING_LEVEL_DEBUG, COOLMIC_ERROR_NONE,
                        "Read request, buffer=%p, len=%zu", buffer, len);
    if (dev->kind == SOURCE_NULL) {
        memset(buffer, 0, len);                  /* silence, always the full request */
    } else if (dev->kind == SOURCE_STDIO) {
        return (ssize_t)fread(buffer, 1, len, dev->file);   /* 0 at end of file */
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
        /* raw PCM replay: `device` is the file name (ref: src/snddev_stdio.c:50-78), RX only */
        kind = SOURCE_STDIO;
        if (device == NULL || *(const char *)device == 0 || !(flags & COOLMIC_DSP_SNDDEV_RX) ||
            (flags & COOLMIC_DSP_SNDDEV_TX))
            return NULL;
        file = fopen(device, "rb");
        if (file == NULL)
            return NULL;
    } else {
        return NULL;
    }

    dev = coolmic_ro_new_raw(&snddev_type, name, associated);
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
