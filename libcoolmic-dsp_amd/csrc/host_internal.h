/* host_internal.h -- declarations shared by the host-side C files of
 * libcoolmic-dsp-hip.so (not installed, not part of the public ABI). */
#ifndef COOLMIC_HOST_INTERNAL_H
#define COOLMIC_HOST_INTERNAL_H

#include <stddef.h>
#include <stdint.h>
#include <sys/types.h>

/* (first: inside the reference's build libigloo's headers want the application's types declared before
 * anything else pulls <igloo/ro.h> in) */
#include "ro_glue.h"

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

/* HIP device the per-object (non-batch) stages run on: $COOLMIC_HIP_DEVICE or 0 -- unless the stage was given
 * one of its own (coolmic_transform_set_device, coolmic_vumeter_set_device, coolmic_group_new_on) */
int coolmic_hip_default_device(void);
int coolmic_hip_stage_device(int chosen_plus1);
int coolmic_hip_check_device(int device);

/* Everything from here on is glue between the translation units of this library: hidden, not exported. */
#pragma GCC visibility push(hidden)

ssize_t coolmic_transform_handle_read(void *userdata, void *buffer, size_t len);
const void *coolmic_iohandle_backend(const coolmic_iohandle_t *self);
ssize_t coolmic_tee_reader_read(void *userdata, void *buffer, size_t len);

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

/* ---- per-launch window records (a meter behind a tee shares the transform's launch) ------------
 * A raw VU window as the device keeps it: sums of squares and packed peak keys per channel
 * (key = |peak| << 47 | (~index & (2^46-1)) << 1 | negative, index in interleaved samples since the
 * window opened; 0: no sample that is not zero) and the interleaved samples accounted. */
typedef struct cmhip_vu_raw {
    uint64_t power[16];
    uint64_t key[16];
    uint64_t samples;
} cmhip_vu_raw_t;
/* ring mode of a batch (engine internal): with `slots` > 0 every run accumulates into a window of its own,
 * slot (sequence number % slots) of a ring of cleared windows, instead of the batch's current window;
 * slots == 0 switches back.  The sequence number of the next run: cmhip_batch_vu_ring_seq(). */
int cmhip_batch_vu_ring(struct cmhip_batch *b, unsigned int slots);
uint64_t cmhip_batch_vu_ring_seq(const struct cmhip_batch *b);
/* the windows of runs first_seq .. first_seq+count-1 of stream 0 (none of them older than `slots` runs):
 * one copy, waits for the batch's stream, clears the slots for their next turn */
int cmhip_batch_vu_ring_fetch(struct cmhip_batch *b, uint64_t first_seq, unsigned int count, cmhip_vu_raw_t *out);
/* the batch's current (ordinary) window of a stream, raw; waits for the batch's stream */
int cmhip_batch_vu_raw_state(struct cmhip_batch *b, unsigned int stream, cmhip_vu_raw_t *out);
/* acc <- acc followed by piece (piece later in the stream): sums add, the piece's sample indices continue
 * the window's, the larger key wins -- what the device's atomics do from launch to launch of one window,
 * and the reference's strict-greater update in stream order (ref: src/vumeter.c:163-168) */
void cmhip_vu_raw_merge(cmhip_vu_raw_t *acc, const cmhip_vu_raw_t *piece, unsigned int channels);
/* ref: src/vumeter.c:189-218 on a raw window; COOLMIC_ERROR_INVAL while it holds no frame */
int cmhip_vu_raw_finish(const cmhip_vu_raw_t *w, unsigned int channels, unsigned int rate,
                        coolmic_vumeter_result_t *out);
int coolmic_transform_vu_take_raw(struct coolmic_transform *self, cmhip_vu_raw_t *raw);
int coolmic_transform_records_merge(struct coolmic_transform *self, uint64_t from, uint64_t to, cmhip_vu_raw_t *acc);
void coolmic_transform_records_drop(struct coolmic_transform *self, uint64_t upto);
int coolmic_transform_vu_result(struct coolmic_transform *self, coolmic_vumeter_result_t *result);
int coolmic_transform_vu_reset(struct coolmic_transform *self);
void coolmic_transform_format(const struct coolmic_transform *self, uint_least32_t *rate, unsigned int *channels);

/* A VU meter behind a coolmic_tee_t (ref: src/simple.c:217-229) shares the transform's launch too: the
 * transform leaves one window record per launch, the tee says where its readers are in the transform's
 * output, the meter merges the records of the bytes it has consumed (transform.c, tee.c, vumeter.c). */
void *coolmic_iohandle_as_tee_reader(coolmic_iohandle_t *h);
struct coolmic_transform *coolmic_tee_reader_upstream(void *reader, uint64_t *next_off, unsigned int *discont);
unsigned int coolmic_tee_reader_discont(void *reader);
int coolmic_transform_records(struct coolmic_transform *self, int on);
uint64_t coolmic_transform_out_bytes(const struct coolmic_transform *self);
uint64_t coolmic_transform_records_start(const struct coolmic_transform *self);
int coolmic_transform_record_at(struct coolmic_transform *self, uint64_t pos, uint64_t *off, uint32_t *bytes);

#pragma GCC visibility pop

#ifdef __cplusplus
}
#endif
#endif
