/*
 * transform.h -- PCM operator: frame alignment + per-channel integer gain, on the GPU.
 *
 * Drop-in for the reference stage (ref: include/coolmic-dsp/transform.h:35-53,
 * src/transform.c).  The first four functions keep the reference's names,
 * argument meaning and return values; the arithmetic of every read runs in the
 * HIP kernels of libcoolmic-dsp-hip.so (there is no CPU path: without a usable
 * device a read that needs arithmetic fails with -1 and an ERROR log line).
 *
 * PCM is native-endian int16, interleaved; a frame is one sample per channel.
 */
#ifndef __COOLMIC_DSP_TRANSFORM_H__
#define __COOLMIC_DSP_TRANSFORM_H__

#include <stdint.h>
#include "ro-compat.h"
#include "iohandle.h"

#ifdef __cplusplus
extern "C" {
#endif

#define COOLMIC_DSP_TRANSFORM_MAX_CHANNELS  16

typedef struct coolmic_transform coolmic_transform_t;

/* NULL when rate or channels is 0 (ref: src/transform.c:65-81); channels above
 * COOLMIC_DSP_TRANSFORM_MAX_CHANNELS are refused here, the reference would overrun. */
coolmic_transform_t   *coolmic_transform_new(const char *name, igloo_ro_t associated,
                                             uint_least32_t rate, unsigned int channels);

/* upstream PCM handle; NULL detaches.  Takes its own reference (ref: :83-92). */
int                    coolmic_transform_attach_iohandle(coolmic_transform_t *self,
                                                         coolmic_iohandle_t *handle);

/* downstream handle: reads return whole frames only, partial frames are carried
 * to the next read, upstream errors surface as a short/zero read (ref: :126-193) */
coolmic_iohandle_t    *coolmic_transform_get_iohandle(coolmic_transform_t *self);

/* gain[c]/scale per channel, unsigned.  channels==stream channels: one each;
 * channels==1: broadcast; channels==2 on a mono stream: truncating mean;
 * otherwise COOLMIC_ERROR_INVAL and nothing changes; channels/scale/gain of 0
 * disables the gain (ref: :195-222).  Result of a sample: trunc(x*gain/scale)
 * saturated to [-32768, 32767] (ref: :101-124). */
int                    coolmic_transform_set_master_gain(coolmic_transform_t *self,
                                                         unsigned int channels, uint16_t scale,
                                                         const uint16_t *gain);

/* ---- additions of this implementation (not in the reference) ---- */

/* out[frame][c] = in[frame][map[c]], applied before the gain; NULL = identity.
 * COOLMIC_ERROR_INVAL if an entry is >= the stream's channels. */
int                    coolmic_transform_set_channel_map(coolmic_transform_t *self,
                                                         const uint8_t *map);

/* IIR equaliser after map and gain: `sections` (0..4) biquads of five floats each,
 * {b0, b1, b2, a1, a2} normalised to a0 = 1, evaluated per channel in Direct Form I on
 * x/32768.f with state of its own per channel; the result goes back to int16 (round to
 * nearest even, saturated).  0 sections switches the filter off and clears its state; new
 * coefficients keep the state.  COOLMIC_ERROR_INVAL for more than 4 sections or a NULL
 * coefficient pointer. */
#define COOLMIC_DSP_TRANSFORM_MAX_EQ_SECTIONS 4
int                    coolmic_transform_set_eq(coolmic_transform_t *self, unsigned int sections,
                                                const float *coef);

/* Which GPU this transform's arithmetic runs on (one process may drive every GPU of a node: one pipeline per
 * capture stream, ref: src/simple.c:198-200, stream s on GPU s % N).  Valid until the first read that needs
 * arithmetic has created the transform's device state: COOLMIC_ERROR_BUSY after that, COOLMIC_ERROR_INVAL for a
 * device the process does not see.  Without a call: $COOLMIC_HIP_DEVICE, else 0. */
int                    coolmic_transform_set_device(coolmic_transform_t *self, int device);

#ifdef __cplusplus
}
#endif
#endif
