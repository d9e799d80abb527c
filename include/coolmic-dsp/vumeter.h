/*
 * vumeter.h -- terminal analyser: per-channel and global peak + power, on the GPU.
 *
 * Drop-in for the reference stage (ref: include/coolmic-dsp/vumeter.h:42-107,
 * src/vumeter.c).  Accumulation (first-max-|x| peak, exact 64-bit sum of squares)
 * runs in HIP kernels; the dB values are finished on the host in double exactly as
 * the reference does (ref: src/vumeter.c:201-212), so results are bit-identical.
 */
#ifndef __COOLMIC_DSP_VUMETER_H__
#define __COOLMIC_DSP_VUMETER_H__

#include <stdint.h>
#include <sys/types.h>
#include "ro-compat.h"
#include "iohandle.h"

#ifdef __cplusplus
extern "C" {
#endif

#define COOLMIC_DSP_VUMETER_MAX_CHANNELS 16

typedef struct coolmic_vumeter coolmic_vumeter_t;

/* Result of one measuring window.  Field order and types are the reference's
 * (ref: vumeter.h:48-83); 192 bytes on LP64.  Peaks are the signed value of the
 * first sample that reached the largest magnitude; powers are in dB relative to
 * full scale, never above 0, -inf for silence. */
typedef struct {
    uint_least32_t rate;
    unsigned int channels;
    size_t frames;
    int16_t global_peak;
    double global_power;
    int16_t channel_peak[COOLMIC_DSP_VUMETER_MAX_CHANNELS];
    double channel_power[COOLMIC_DSP_VUMETER_MAX_CHANNELS];
} coolmic_vumeter_result_t;

coolmic_vumeter_t  *coolmic_vumeter_new(const char *name, igloo_ro_t associated,
                                        uint_least32_t rate, unsigned int channels);
int                 coolmic_vumeter_reset(coolmic_vumeter_t *self);
int                 coolmic_vumeter_attach_iohandle(coolmic_vumeter_t *self,
                                                    coolmic_iohandle_t *handle);
/* pulls at most maxlen bytes (-1: internal default of 1024) and accounts the whole
 * frames; returns the bytes pulled, -1 on error with nothing buffered */
ssize_t             coolmic_vumeter_read(coolmic_vumeter_t *self, ssize_t maxlen);
/* COOLMIC_ERROR_INVAL while the window holds no frame; resets the window on success */
int                 coolmic_vumeter_result(coolmic_vumeter_t *self,
                                           coolmic_vumeter_result_t *result);

/* ---- addition of this implementation (not in the reference) ---- */

/* Which GPU a meter with a launch of its own accounts its frames on (a meter that shares the launch of the
 * transform above it uses that transform's, coolmic_transform_set_device()).  COOLMIC_ERROR_BUSY once the meter
 * has device state of its own, COOLMIC_ERROR_INVAL for a device the process does not see.  Without a call:
 * $COOLMIC_HIP_DEVICE, else 0. */
int                 coolmic_vumeter_set_device(coolmic_vumeter_t *self, int device);

#ifdef __cplusplus
}
#endif
#endif
