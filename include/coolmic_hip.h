/*
 * coolmic_hip.h -- C ABI of the MI355X batch engine behind the transform -> vumeter path.
 *
 * Plain pointers and sizes only.  One cmhip_batch_t owns, on one GPU, a block of
 * S independent capture streams that share a channel count: their PCM slots in
 * HBM, their gain / channel-map / EQ parameters and their VU accumulators.  One
 * cmhip_batch_run() is one pass of the reference's per-sample loops over every
 * stream of the batch:
 *
 *   replaces, per stream and per block:
 *     __process()            ref: src/transform.c:101-124   (gain, saturate)
 *     the accumulate loop    ref: src/vumeter.c:161-177     (peak, sum of squares)
 *     int16 -> float planar  ref: src/enc_vorbis.c:108-115  (optional output)
 *   and cmhip_batch_vu_result() replaces
 *     coolmic_vumeter_result ref: src/vumeter.c:189-218     (dB on the host, double)
 *
 * The per-stream objects of <coolmic-dsp/transform.h> and <coolmic-dsp/vumeter.h>
 * sit on top of this engine.  Every function returns a COOLMIC_ERROR_* number
 * (<coolmic-dsp/coolmic-dsp.h>) unless stated; cmhip_last_error() has the text.
 *
 * HBM layout: pcm[stream][frame][channel], int16, each stream's slot contiguous
 * and 16-byte aligned (cmhip_batch_stride() samples apart).  Planar float output:
 * f32[stream][channel][frame], planes cmhip_batch_max_frames() apart.
 */
#ifndef COOLMIC_HIP_H
#define COOLMIC_HIP_H

#include <stddef.h>
#include <stdint.h>
#include <coolmic-dsp/vumeter.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cmhip_batch cmhip_batch_t;

/* what a run produces */
#define CMHIP_OUT_PCM      0x0001u   /* materialise the transformed int16 PCM */
#define CMHIP_OUT_F32      0x0002u   /* planar float copy of the transformed PCM */
#define CMHIP_VU           0x0004u   /* accumulate the VU meters */
#define CMHIP_INPLACE      0x0008u   /* PCM output overwrites the input slots (as the reference does) */
#define CMHIP_EQ           0x0010u   /* biquad EQ after map and gain, every channel with state of its own */
#define CMHIP_HOSTPCM      0x0020u   /* PCM slots in pinned host memory the kernels access directly (zero copy):
                                      * for small batches fed block by block, e.g. the per-stream stages */
#define CMHIP_EXTSLOTS     0x0040u   /* no PCM slots of its own: every run names them (cmhip_batch_run_slots) */
#define CMHIP_PLACE_SEARCH 0x0080u   /* at creation, look for a faster physical placement of the two PCM arrays
                                      * (cmhip_batch_placement below).  OFF unless asked for: the search takes
                                      * memory and time for a few per cent of the kernel */

/* synthetic inputs generated on the device */
#define CMHIP_GEN_NULL     0         /* zeros, as snddev "null" */
#define CMHIP_GEN_SINE     1         /* 48-sample sine period, stream s starts at phase 7*s */
#define CMHIP_GEN_NOISE    2         /* per-stream LCG, seed + global stream id */

#define CMHIP_MAX_EQ_SECTIONS 4

typedef struct cmhip_batch_desc {
    int          device;          /* HIP device ordinal */
    unsigned int streams;         /* S >= 1 */
    unsigned int channels;        /* 1..16, shared by the batch */
    unsigned int rate;            /* Hz, reported in results */
    size_t       max_frames;      /* slot capacity per stream, frames */
    unsigned int flags;           /* CMHIP_* outputs */
    void        *hip_stream;      /* hipStream_t to launch on, NULL: own stream */
} cmhip_batch_desc_t;

/* ---- process level ------------------------------------------------------- */
int          cmhip_device_count(void);            /* 0 without a usable GPU */
int          cmhip_device_synchronize(int device); /* everything queued on that device has finished */
int          cmhip_device_mem_info(int device, size_t *free_bytes, size_t *total_bytes); /* hipMemGetInfo */
/* plain device memory for a host without HIP headers of its own (e.g. the destination of
 * cmhip_batch_vu_node_partial for a host that brings its own collective); zero-filled */
void        *cmhip_device_alloc(int device, size_t bytes);        /* NULL on failure */
void         cmhip_device_free(int device, void *p);
int          cmhip_device_read(int device, void *dst_host, const void *src_device, size_t bytes); /* synchronises the device */
const char  *cmhip_last_error(void);              /* per-thread text of the last failure */
const char  *cmhip_version(void);

/* ---- life cycle ---------------------------------------------------------- */
cmhip_batch_t *cmhip_batch_new(const cmhip_batch_desc_t *desc);   /* NULL on failure */
void           cmhip_batch_free(cmhip_batch_t *b);

/* ---- placement of the PCM arrays (CMHIP_PLACE_SEARCH) ---------------------- */
/* On MI355X a kernel that streams one large array in and another out runs 3-5 % faster when the two
 * lie in different stretches of the card's memory; nothing but probing shows which.  A batch created
 * with CMHIP_PLACE_SEARCH (two PCM arrays of its own, >= 256 MiB each) allocates up to five more
 * candidate arrays behind spacer allocations, times its own run on every pair, keeps the fastest pair
 * if it beats the first by 2 % and frees the rest before cmhip_batch_new() returns.  It never asks
 * for more than HALF of the memory hipMemGetInfo reports free (spacers and candidates together;
 * fewer candidates on a fuller card) and stops allocating after 0.3 s.  Without the flag nothing of
 * this happens: two hipMalloc calls, no probe launches.  $CMHIP_PLACE overrides for experiments:
 * 0 never, 1 the first large batch of a device even without the flag, 2 every large batch. */
typedef struct cmhip_placement {
    int      searched;          /* 1 when the probes ran */
    int      candidates;        /* arrays probed, the first two included (2..7) */
    int      chosen_in;         /* candidate kept as the input array (0 = where hipMalloc first put it) */
    int      chosen_out;        /* candidate kept as the output array (1 = where hipMalloc first put it) */
    int      probe_launches;    /* launches of the batch's kernel the search made */
    double   first_pair_ms;     /* the run on the first pair (median of its probes; no gain, no map) */
    double   best_pair_ms;      /* the run on the fastest other pair */
    double   search_ms;         /* wall time of the whole search */
    uint64_t bytes_requested;   /* spacers + extra candidates the search allocated (all freed again) */
    uint64_t bytes_free_before; /* hipMemGetInfo free when it started */
} cmhip_placement_t;
int cmhip_batch_placement(const cmhip_batch_t *b, cmhip_placement_t *out);

/* ---- parameters (take effect at the next run) ---------------------------- */
/* same meaning and return values as coolmic_transform_set_master_gain
 * (ref: src/transform.c:195-222); stream == -1 addresses every stream */
int cmhip_batch_set_gain(cmhip_batch_t *b, long stream, unsigned int channels, uint16_t scale,
                         const uint16_t *gain);
/* out channel c reads input channel map[c]; NULL = identity */
int cmhip_batch_set_chmap(cmhip_batch_t *b, long stream, const uint8_t *map);
/* nsec biquads, 5 floats each {b0,b1,b2,a1,a2} (a0-normalised); nsec 0 = bypass.
 * Every channel of a stream runs the stream's filter with state of its own, kept across
 * runs; cmhip_batch_eq_reset() zeroes it.  The section count is one for the whole batch:
 * stream -1 sets all streams (and may change the count), a stream >= 0 must keep it. */
int cmhip_batch_set_eq(cmhip_batch_t *b, long stream, unsigned int nsec, const float *coef);
int cmhip_batch_eq_reset(cmhip_batch_t *b, long stream);
/* RBJ designs on the host in double, cast to float: kind 0 low shelf, 1 peaking, 2 high
 * shelf (shelf slope 1).  coef receives 5 floats. */
void cmhip_design_biquad(int kind, double rate, double freq, double gain_db, double q,
                         float *coef);

/* ---- geometry and raw device pointers ------------------------------------ */
size_t  cmhip_batch_stride(const cmhip_batch_t *b);       /* samples between stream slots */
size_t  cmhip_batch_max_frames(const cmhip_batch_t *b);
void   *cmhip_batch_dev_in(cmhip_batch_t *b);             /* int16 [S][stride] */
void   *cmhip_batch_dev_out(cmhip_batch_t *b);            /* int16 [S][stride] (== in when INPLACE) */
void   *cmhip_batch_dev_f32(cmhip_batch_t *b);            /* float [S][C][max_frames] or NULL */
void   *cmhip_batch_hip_stream(cmhip_batch_t *b);         /* the hipStream_t launches go to */

/* ---- moving PCM (asynchronous on the batch's stream) ---------------------- */
int cmhip_batch_upload(cmhip_batch_t *b, unsigned int stream, const int16_t *pcm, size_t frames);
int cmhip_batch_download(cmhip_batch_t *b, unsigned int stream, int16_t *pcm, size_t frames);
/* whole-batch forms: `host` mirrors the device layout, int16 [S][cmhip_batch_stride()], and
 * one copy moves every slot; asynchronous on the batch's stream (cmhip_batch_sync() to wait).
 * Pinned memory from cmhip_host_alloc() gives full PCIe speed and real asynchrony. */
int   cmhip_batch_upload_all(cmhip_batch_t *b, const int16_t *host, size_t frames);
int   cmhip_batch_download_all(cmhip_batch_t *b, int16_t *host, size_t frames);
void *cmhip_host_alloc(size_t bytes);             /* NULL on failure */
/* pinned and mapped into the device: the host uses the returned pointer, kernels *device_ptr.
 * (_on: the device whose kernels will use it -- a process that drives several GPUs; the plain form takes
 * whichever device is current in the calling thread) */
void *cmhip_host_alloc_mapped(size_t bytes, void **device_ptr);
void *cmhip_host_alloc_mapped_on(int device, size_t bytes, void **device_ptr);
void  cmhip_host_free(void *p);
/* reads an input slot back (generated or uploaded PCM); synchronises */
int cmhip_batch_download_input(cmhip_batch_t *b, unsigned int stream, int16_t *pcm, size_t frames);
int cmhip_batch_download_f32(cmhip_batch_t *b, unsigned int stream, unsigned int channel,
                             float *dst, size_t frames);
/* fill every stream's input slot on the device; stream s of this batch is global
 * stream first_global + s*global_step (round-robin shards use first=rank, step=N) */
int cmhip_batch_generate(cmhip_batch_t *b, int mode, uint32_t seed, size_t frames,
                         uint64_t first_global, uint64_t global_step, uint64_t frame_offset);

/* ---- the hot path --------------------------------------------------------- */
/* process `frames` frames of every stream (frames_per_stream, if not NULL, gives each
 * stream its own count <= frames; host array of S entries).  Asynchronous. */
int cmhip_batch_run(cmhip_batch_t *b, size_t frames, const uint32_t *frames_per_stream);
/* the same pass over PCM arrays named for this run: device-accessible memory laid out like the
 * batch's own, int16 [S][cmhip_batch_stride()] (slots_out NULL exactly when the batch writes no
 * PCM, == slots_in for CMHIP_INPLACE).  With pinned, device-mapped host memory
 * (cmhip_host_alloc_mapped) the kernel moves the block over PCIe itself, and a host can rotate
 * several sets: sources fill one, readers drain another, a third is on the GPU. */
int cmhip_batch_run_slots(cmhip_batch_t *b, size_t frames, const uint32_t *frames_per_stream,
                          const void *slots_in, void *slots_out);
int cmhip_batch_sync(cmhip_batch_t *b);

/* ---- VU windows ------------------------------------------------------------ */
/* result of one stream's current window, then reset of that window -- the contract of
 * coolmic_vumeter_result (ref: src/vumeter.c:189-218), INVAL while it holds no frame */
int cmhip_batch_vu_result(cmhip_batch_t *b, unsigned int stream, coolmic_vumeter_result_t *out);
/* all streams at once: out[S], rc[S] (rc may be NULL); one device round trip */
int cmhip_batch_vu_results(cmhip_batch_t *b, coolmic_vumeter_result_t *out, int *rc);
/* two-phase form that overlaps with the next run: snapshot copies the accumulators of
 * every stream to pinned host memory and opens a new window on the device (async);
 * collect waits for that copy and finishes the dB values on the host */
int cmhip_batch_vu_snapshot(cmhip_batch_t *b);
int cmhip_batch_vu_collect(cmhip_batch_t *b, coolmic_vumeter_result_t *out, int *rc);
/* collect in two halves, for hosts that close a window every block of a few thousand frames: begin waits for
 * the oldest snapshot and hands its windows to the helper threads, end returns when out[] and rc[] (which
 * must stay valid until then) are complete; between the two the caller queues its next run.  One at a time;
 * the snapshot keeps its place among the three that may be pending (cmhip_batch_vu_snapshot returns
 * COOLMIC_ERROR_BUSY for a fourth) until end. */
int cmhip_batch_vu_collect_begin(cmhip_batch_t *b, coolmic_vumeter_result_t *out, int *rc);
int cmhip_batch_vu_collect_end(cmhip_batch_t *b);
int cmhip_batch_vu_reset(cmhip_batch_t *b, long stream);
/* raw accumulators of a stream (synchronises): power[16], peak[16], frames */
int cmhip_batch_vu_raw(cmhip_batch_t *b, unsigned int stream, int64_t *power, int16_t *peak,
                       uint64_t *frames);

/* ---- node-global VU ---------------------------------- */
/* Reduces this batch's current windows over its streams into one record of
 * CMHIP_NODE_WORDS int64 words written to device memory `dst` (asynchronous):
 *   [0..15]  sum of squares per channel      -> combine across GPUs with SUM
 *   [16]     frames summed over streams      -> SUM
 *   [17..32] packed peak key per channel     -> combine with MAX
 *   [33]     packed global peak key          -> MAX
 * Keys order by (|peak|, earliest frame, lowest global stream id); decode with
 * cmhip_node_finish().  A host that brings its own collective reduces `dst` itself; one
 * that wants the engine to do it uses cmhip_node_t below. */
#define CMHIP_NODE_WORDS      34
#define CMHIP_NODE_SUM_WORDS  17
int cmhip_batch_vu_node_partial(cmhip_batch_t *b, void *dst_device, uint64_t first_global,
                                uint64_t global_step);
/* the same record straight to host memory, words[CMHIP_NODE_WORDS]; waits for the batch's last run */
int cmhip_batch_vu_node_record(cmhip_batch_t *b, int64_t *words_host, uint64_t first_global,
                               uint64_t global_step);
/* host: turn a combined record into a result (frames = total frames over streams) */
int cmhip_node_finish(const int64_t *words, unsigned int channels, unsigned int rate,
                      coolmic_vumeter_result_t *out);

/* The exchange itself, in C over RCCL (xGMI inside a node) -- no Python, no torch: one
 * cmhip_node_t per GPU (= per rank; one process per GPU, or one thread per GPU of one
 * process).  It owns two sets of `max_records` record slots on its device, so that the
 * records of B blocks travel in ONE pair of collectives (the exchange is latency bound) and
 * a set can be exchanged while the next blocks fill the other.  Per set the sums of all
 * slots are contiguous and so are the keys:
 *     ncclAllReduce(sums, B*17, ncclInt64,  ncclSum)     words 0..16  of every record
 *     ncclAllReduce(keys, B*17, ncclUint64, ncclMax)     words 17..33 of every record
 * issued as one RCCL group on the node's own HIP stream.  librccl is loaded when the first
 * node is created (dlopen of librccl.so.1; a host that never asks for the node-global VU
 * never maps it). */
typedef struct cmhip_node cmhip_node_t;
#define CMHIP_NODE_ID_BYTES 128
/* rank 0 makes the id (ncclGetUniqueId) and hands the 128 bytes to the other ranks by
 * whatever means the host has (a socket, a file, shared memory of one process) */
int           cmhip_node_unique_id(void *id128);
/* collective over all ranks (ncclCommInitRank): every rank calls it with the same id */
cmhip_node_t *cmhip_node_new(int device, int nranks, int rank, const void *id128,
                             unsigned int max_records);
void          cmhip_node_free(cmhip_node_t *n);
int           cmhip_node_ranks(const cmhip_node_t *n);
/* "hip=<path of the HIP runtime the engine is bound to> rccl=<path of the librccl it loaded>": librccl is
 * taken from next to that runtime (a process that also imported a torch wheel holds a second pair) */
const char   *cmhip_node_runtime(void);
/* the batch's current windows -> slot `slot` of set `set` (asynchronous, beside the batch's
 * next run; the batch must live on the node's device).  Waits, on the device, for the last
 * exchange of that set, and clears the set when its first slot after an exchange is filled:
 * fetch a set's results before putting the next block into it. */
int cmhip_node_partial(cmhip_node_t *n, cmhip_batch_t *b, unsigned int set, unsigned int slot,
                       uint64_t first_global, uint64_t global_step);
/* all-reduce slots 0..count-1 of `set` over the ranks, in place, after everything the batch
 * `after` (may be NULL) has queued so far; asynchronous -- the host does not wait */
int cmhip_node_allreduce(cmhip_node_t *n, unsigned int set, unsigned int count, cmhip_batch_t *after);
/* wait for the exchange of `set` and copy its combined records to the host:
 * words[count][CMHIP_NODE_WORDS], ready for cmhip_node_finish() */
int cmhip_node_fetch(cmhip_node_t *n, unsigned int set, unsigned int count, int64_t *words);
/* "replicas only" form of the same combine on the host (no collective): SUM / MAX of `nranks`
 * records into out[CMHIP_NODE_WORDS]; the parity check of the RCCL path */
int cmhip_node_merge_host(const int64_t *records, unsigned int nranks, int64_t *out);

/* ---- measurement ----------------------------------------------------------- */
/* enable = 1: every run is bracketed by hipEvents on the batch's stream (stamped by the kernel's own
 * dispatch); enable = n > 1: every n-th run only -- the events cost a run about 5 us of its stream's time,
 * a sample of the launches leaves the throughput as it is without them; 0: off */
int cmhip_batch_timing(cmhip_batch_t *b, int enable);
/* sums since the last call: milliseconds and launches of the dominant kernel; resets */
int cmhip_batch_timing_read(cmhip_batch_t *b, double *kernel_ms, unsigned int *launches);
/* plain HBM ceilings measured with the same buffers: mode 0 read-only sum, 1 copy.
 * Returns GB/s of algorithmic bytes (read: bytes; copy: 2*bytes) or <0 on error. */
double cmhip_batch_ceiling(cmhip_batch_t *b, int mode, size_t frames, int iters);

#ifdef __cplusplus
}
#endif
#endif
