/*
 * snddev.h -- PCM sources used to drive the chain: "null" (silence) and "sine"
 * (1 kHz, amplitude 32766, mono).  Same constructor and handle contract as the
 * reference (ref: include/coolmic-dsp/snddev.h:40-83, src/snddev.c:98-169,
 * src/snddev_sine.c:118-193, src/snddev_null.c:33-55).  Only the capture (RX)
 * side is provided; hardware drivers are out of scope.
 */
#ifndef __COOLMIC_DSP_SNDDEV_H__
#define __COOLMIC_DSP_SNDDEV_H__

#include <stdint.h>
#include <sys/types.h>
#include "ro-compat.h"
#include "iohandle.h"

#ifdef __cplusplus
extern "C" {
#endif

#define COOLMIC_DSP_SNDDEV_DRIVER_AUTO   NULL
#define COOLMIC_DSP_SNDDEV_DRIVER_NULL   "null"
#define COOLMIC_DSP_SNDDEV_DRIVER_SINE   "sine"
#define COOLMIC_DSP_SNDDEV_DRIVER_STDIO  "stdio"    /* raw PCM file replay; device = file name */

#define COOLMIC_DSP_SNDDEV_RX    0x0001
#define COOLMIC_DSP_SNDDEV_TX    0x0002
#define COOLMIC_DSP_SNDDEV_RXTX  (COOLMIC_DSP_SNDDEV_RX|COOLMIC_DSP_SNDDEV_TX)

typedef struct coolmic_snddev coolmic_snddev_t;

/* NULL for rate/channels/flags of 0, an unknown driver, or a driver that refuses
 * the format (sine: mono only, rate must be 8/16/24/32/44/44.1/48/96 kHz; stdio: the file
 * must open for reading, capture only) */
coolmic_snddev_t   *coolmic_snddev_new(const char *name, igloo_ro_t associated, const char *driver,
                                       void *device, uint_least32_t rate, unsigned int channels,
                                       int flags, ssize_t buffer);
/* endless capture handle (no eof callback), keeps the device alive while it lives */
coolmic_iohandle_t *coolmic_snddev_get_iohandle(coolmic_snddev_t *self);

#ifdef __cplusplus
}
#endif
#endif
