/*
 * snddev.h -- PCM sources used to drive the chain: "null" (silence), "sine" (1 kHz,
 * amplitude 32766, mono) and "stdio" (a raw PCM file).  Same functions and contracts as
 * the reference (ref: include/coolmic-dsp/snddev.h:40-83, src/snddev.c:98-215,
 * src/snddev_sine.c:118-193, src/snddev_null.c:33-55, src/snddev_stdio.c:50-78), both the
 * capture handle and the playback side.  Hardware drivers (oss, opensl) and the driver
 * vtable they plug into are the reference's own: inside its build its snddev*.c stay
 * (INTEGRATION.md 3), and this header is the stand-alone library's.
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
/* hardware drivers of the reference: known by name, not built into the stand-alone library (coolmic_snddev_new
 * returns NULL for them); inside the reference's build its own snddev*.c serve them */
#define COOLMIC_DSP_SNDDEV_DRIVER_OSS    "oss"
#define COOLMIC_DSP_SNDDEV_DRIVER_OPENSL "opensl"

#define COOLMIC_DSP_SNDDEV_RX    0x0001
#define COOLMIC_DSP_SNDDEV_TX    0x0002
#define COOLMIC_DSP_SNDDEV_RXTX  (COOLMIC_DSP_SNDDEV_RX|COOLMIC_DSP_SNDDEV_TX)

typedef struct coolmic_snddev coolmic_snddev_t;

/* What a driver fills in for the device object (ref: include/coolmic-dsp/snddev.h:55-68): declared so that a
 * host's driver sources compile against this header.  The stand-alone library has its three sources built in and
 * no registry for further drivers; inside the reference's build its own snddev.c, which owns that registry,
 * stays. */
typedef struct coolmic_snddev_driver coolmic_snddev_driver_t;
struct coolmic_snddev_driver {
    int (*free)(coolmic_snddev_driver_t *dev);
    ssize_t (*read)(coolmic_snddev_driver_t *dev, void *buffer, size_t len);         /* capture */
    ssize_t (*write)(coolmic_snddev_driver_t *dev, const void *buffer, size_t len);  /* playback */
    int userdata_i;                    /* the driver's own */
    void *userdata_vp;
};

/* NULL for rate/channels/flags of 0, an unknown driver, or a driver that refuses
 * the format (sine: mono only, rate must be 8/16/24/32/44/44.1/48/96 kHz; stdio: the file
 * must open -- "rb" for RX, "wb" for TX, "w+b" for both) */
coolmic_snddev_t   *coolmic_snddev_new(const char *name, igloo_ro_t associated, const char *driver,
                                       void *device, uint_least32_t rate, unsigned int channels,
                                       int flags, ssize_t buffer);
/* the handle whose PCM is to be played back by coolmic_snddev_iter(); NULL detaches
 * (ref: src/snddev.c:143-152) */
int                 coolmic_snddev_attach_iohandle(coolmic_snddev_t *self, coolmic_iohandle_t *handle);
/* endless capture handle (no eof callback), keeps the device alive while it lives */
coolmic_iohandle_t *coolmic_snddev_get_iohandle(coolmic_snddev_t *self);
/* one round of playback: flushes what the device has not taken yet, then reads up to 1 KiB from
 * the attached handle and hands it to the device.  COOLMIC_ERROR_NONE, _BUSY (the device took
 * only part), _GENERIC (read or write failed).  null and sine discard, stdio writes to its file
 * (ref: src/snddev.c:171-215) */
int                 coolmic_snddev_iter(coolmic_snddev_t *self);

#ifdef __cplusplus
}
#endif
#endif
