/*
 * group.h -- many transform -> vumeter pipelines served by ONE launch per block.
 *
 * Not in the reference.  A host that runs thousands of capture streams builds one
 * coolmic_group_t instead of thousands of coolmic_transform_t / coolmic_vumeter_t
 * pairs; every stream keeps the reference's operator interface on both sides:
 *
 *   upstream:    any coolmic_iohandle_t per stream (snddev, file, user callback)
 *   downstream:  coolmic_group_get_iohandle(slot) -- a coolmic_iohandle_t that delivers
 *                the stream's transformed PCM in whole frames, like the handle of
 *                coolmic_transform_get_iohandle() (ref: src/transform.c:126-193)
 *   meter:       coolmic_group_vumeter_result(slot) -- the contract of
 *                coolmic_vumeter_result() (ref: src/vumeter.c:189-218)
 *
 * coolmic_group_pump() moves one block: it pulls up to block_frames from every
 * upstream handle into pinned staging (partial frames are carried to the next pump,
 * as ref: src/transform.c:155-160 does per read) and queues upload, the fused
 * channel-map/gain/VU kernel over all streams and the download on the GPU -- then it
 * returns.  The block's PCM reaches the streams' output queues with the next pump or
 * with the first read that finds a queue empty, so a host that pumps and reads in a
 * loop reads block k while the GPU works on block k+1.  A read on an empty downstream
 * handle first brings the block in flight home and only then pumps by itself, so a
 * purely pull-driven chain works unchanged and never pulls more than it did; streams
 * are read ahead by at most `queue_blocks` blocks (the block in flight counts; use
 * >= 2 for the overlap), and parameter changes take effect at the next pump.
 */
#ifndef __COOLMIC_DSP_GROUP_H__
#define __COOLMIC_DSP_GROUP_H__

#include <stdint.h>
#include <sys/types.h>
#include "ro-compat.h"
#include "iohandle.h"
#include "vumeter.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct coolmic_group coolmic_group_t;

/* NULL for rate/channels/max_streams/block_frames of 0, channels > 16, or no usable GPU.
 * queue_blocks >= 1 is how many processed blocks a stream may hold unread. */
coolmic_group_t    *coolmic_group_new(const char *name, igloo_ro_t associated, uint_least32_t rate,
                                      unsigned int channels, unsigned int max_streams,
                                      size_t block_frames, unsigned int queue_blocks);

/* The same group on a GPU of the caller's choice instead of $COOLMIC_HIP_DEVICE (default 0): a host that drives
 * every GPU of a node from one process makes one group per GPU and gives capture stream s to the group of GPU
 * s % N, one pump thread per group (examples/group_server.c --gpus N).  NULL also for a device the
 * process does not see. */
coolmic_group_t    *coolmic_group_new_on(int device, const char *name, igloo_ro_t associated, uint_least32_t rate,
                                         unsigned int channels, unsigned int max_streams,
                                         size_t block_frames, unsigned int queue_blocks);
/* the GPU a group lives on; -1 for NULL */
int                 coolmic_group_device(coolmic_group_t *self);
/* the group's batch engine (cmhip_batch_t of <coolmic_hip.h>), for what the group itself does not wrap: the
 * node-global VU over the groups of several GPUs (cmhip_node_partial(node, engine, ...) after a pump; the record
 * counts the block that pump put in flight).  Owned by the group; not to be run, resized or freed. */
struct cmhip_batch *coolmic_group_engine(coolmic_group_t *self);

/* adds a stream fed by `source` (takes its own reference); returns the slot (>= 0) or
 * COOLMIC_ERROR_FAULT / COOLMIC_ERROR_BUSY when the group is full */
int                 coolmic_group_add_stream(coolmic_group_t *self, coolmic_iohandle_t *source);

/* per-stream parameters, rules of coolmic_transform_set_master_gain / _set_channel_map */
int                 coolmic_group_set_master_gain(coolmic_group_t *self, unsigned int slot,
                                                  unsigned int channels, uint16_t scale,
                                                  const uint16_t *gain);
int                 coolmic_group_set_channel_map(coolmic_group_t *self, unsigned int slot,
                                                  const uint8_t *map);
/* equaliser as coolmic_transform_set_eq().  The number of sections is one for the whole
 * group: slot -1 sets every stream (and may change the count), a slot >= 0 only replaces
 * that stream's coefficients and must keep the count (else COOLMIC_ERROR_INVAL). */
int                 coolmic_group_set_eq(coolmic_group_t *self, int slot, unsigned int sections,
                                         const float *coef);

/* transformed PCM of one stream; keeps the group alive while it lives */
coolmic_iohandle_t *coolmic_group_get_iohandle(coolmic_group_t *self, unsigned int slot);

/* one block for every stream whose queue has room.  Returns the number of streams whose source
 * delivered at least one frame to this block, 0 when no source had anything (the block that was
 * in flight is in the queues then), negative on error. */
int                 coolmic_group_pump(coolmic_group_t *self);

/* The pull of a pump -- one read per upstream handle -- spread over `threads` threads (the pumping thread
 * counts; 0 or 1: the pumping thread alone, which is how a group starts).  The handles of DIFFERENT streams
 * are then read at the same time, each stream's own handle still by one thread at a time and once per pump:
 * fine for the sources of this library and for callbacks that keep their state per stream, as the reference's
 * pipelines do with one thread each (ref: src/simple.c:292-310).  Streams that were given the same handle, or
 * handles over the same backend object (one device, one tee), are recognised when they are added and stay on
 * the pumping thread, in slot order.  What the group cannot see -- read callbacks of different userdata that
 * share state behind it -- is the caller's to keep thread-safe.  COOLMIC_ERROR_INVAL above 64. */
int                 coolmic_group_set_pull_threads(coolmic_group_t *self, unsigned int threads);

/* VU window of one stream since its last result; COOLMIC_ERROR_INVAL while it holds no frame */
int                 coolmic_group_vumeter_result(coolmic_group_t *self, unsigned int slot,
                                                 coolmic_vumeter_result_t *result);

unsigned int        coolmic_group_streams(coolmic_group_t *self);

#ifdef __cplusplus
}
#endif
#endif
