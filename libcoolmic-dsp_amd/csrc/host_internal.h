/* host_internal.h -- declarations shared by the host-side C files of
 * libcoolmic-dsp-hip.so (not installed, not part of the public ABI). */
#ifndef COOLMIC_HOST_INTERNAL_H
#define COOLMIC_HOST_INTERNAL_H

#include <stddef.h>
#include <stdint.h>
#include <sys/types.h>

#include <coolmic-dsp/coolmic-dsp.h>
#include <coolmic-dsp/iohandle.h>
#include <coolmic-dsp/logging.h>
#include <coolmic-dsp/ro-compat.h>
#include <coolmic-dsp/vumeter.h>

#ifdef __cplusplus
extern "C" {
#endif

/* one period of the 1 kHz test tone at `rate` (ref: src/snddev_sine.c:36-99,184):
 * rate/1000 samples, amplitude 32766; COOLMIC_ERROR_NOSYS for unsupported rates */
int coolmic_sine_period(uint_least32_t rate, int16_t *table, size_t *samples);

/* HIP device the per-object (non-batch) stages run on: $COOLMIC_HIP_DEVICE or 0 */
int coolmic_hip_default_device(void);

/* A VU meter attached directly to a transform's handle shares the transform's launch (vumeter.c,
 * transform.c): the handle is recognised, the transform accumulates the window, the meter reads it. */
struct coolmic_transform;
struct coolmic_transform *coolmic_iohandle_as_transform(coolmic_iohandle_t *h);
int coolmic_transform_fuse_vu(struct coolmic_transform *self, int on);
/* the meter brackets its own reads with arm(1) / arm(0): only their frames enter the window, whoever
 * else reads the transform's handle meanwhile */
void coolmic_transform_arm_vu(struct coolmic_transform *self, int armed);
struct cmhip_batch;
void cmhip_batch_vu_pause(struct cmhip_batch *b, int paused);
int coolmic_transform_vu_result(struct coolmic_transform *self, coolmic_vumeter_result_t *result);
int coolmic_transform_vu_reset(struct coolmic_transform *self);
void coolmic_transform_format(const struct coolmic_transform *self, uint_least32_t *rate, unsigned int *channels);

#ifdef __cplusplus
}
#endif
#endif
