// cmhip_internal.h -- device-side records and launcher prototypes shared by
// k_block.hip / k_eq.hip / k_misc.hip (the gfx950 kernels) and cmhip_batch.hip (the engine).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

namespace cmhip {

constexpr unsigned MAX_CH = 16;
constexpr unsigned MAX_EQ = 4;

// Per-stream transform parameters, rebuilt on the host whenever a setter runs.
// The reference's q = trunc(x * gain / scale) (ref: src/transform.c:111-119) is done on magnitudes with
// the division split at the integer part of the gain:
//     gain = mi * scale + r   (r < scale),   mf = ceil(r * 2^32 / scale)   (< 2^32 because r <= scale - 1)
//     floor(|x| * gain / scale) = |x| * mi + mulhi(|x|, mf)
// exact for every |x| <= 32768, gain <= 65535, scale in 1..65535: with e = mf * scale - r * 2^32 in
// [0, scale) the product |x| * mf / 2^32 lies |x| * e / (scale * 2^32) < 2^31 / (scale * 2^32) < 1 / scale
// above |x| * r / scale, which is itself at least 1 / scale below the next integer when it is not one
// (cmhip_test_gain_consts / test_division_constants_are_exact prove it per class of gain and scale).
// Two VALU instructions per sample (v_mul_hi_u32, v_mad_u32_u16) instead of the three of a division
// by magic number and shift.  A disabled gain (scale 0 in the reference, ref: src/transform.c:107-108) is
// stored as mi 1 / mf 0, the identity through the same code.
//
// mode names the shorter forms the read-only runs take where the VALU binds (a VU window and no PCM
// result).  GAIN_IDENTITY: the gain is disabled or every gain equals the scale -- the magnitudes are the
// samples' own.  GAIN_BELOW_SCALE: every gain of the stream is below its scale -- mi is 0 everywhere, one
// v_mul_hi_u32 per sample, and the quotient never reaches the saturation limits.  GAIN_GENERAL: the rest.
// 128 bytes, one cache line per stream (round 3 kept the short forms in a table of their own, 96 + 68 bytes).
struct alignas(128) StreamParam {
    uint32_t mode;             // GAIN_GENERAL / GAIN_BELOW_SCALE / GAIN_IDENTITY
    uint32_t perm2;            // stereo channel map as a v_perm_b32 selector
    uint32_t map_identity;     // 1 when chmap is the identity
    uint32_t mi01;             // mi[0] | mi[1] << 16: what the mono / stereo kernels need sits in the first 32 bytes
    uint32_t mf[MAX_CH];       // ceil((gain[c] % scale) * 2^32 / scale)
    uint16_t mi[MAX_CH];       // gain[c] / scale
    uint8_t  chmap[MAX_CH];    // out channel c reads in channel chmap[c]
};
static_assert(sizeof(StreamParam) == 128, "one line per stream");
constexpr uint32_t GAIN_GENERAL = 0, GAIN_BELOW_SCALE = 1, GAIN_IDENTITY = 2;

// Per-stream VU window, all 64-bit so that every update is an integer atomic
// (add / max are associative and commutative: results do not depend on the order
// in which waves arrive).
//   key = |peak| << 47 | (~sample_index & (2^46-1)) << 1 | negative
// sample_index counts interleaved samples since the window opened, so the largest
// key is the largest magnitude and, among equals, the earliest sample: the
// reference's strict-greater update (ref: src/vumeter.c:163-168).
struct VuState {
    unsigned long long power[MAX_CH];
    unsigned long long key[MAX_CH];
    // interleaved samples accounted so far.  Two slots: a run reads slot `parity` and the
    // stream's first tile writes slot `parity^1`, so no wave can see a half-updated value
    // and no extra kernel is needed to advance the window; the host flips parity per run.
    unsigned long long samples[2];
};

constexpr int      KEY_ABS_SHIFT = 47;
constexpr uint64_t KEY_IDX_MASK  = (1ull << 46) - 1;

struct EqParam {
    uint32_t nsec;
    float    coef[MAX_EQ][5];  // b0 b1 b2 a1 a2
};
struct EqState {
    float s[MAX_EQ][4];        // x1 x2 y1 y2 per section
};

struct RunArgs {
    const int16_t *in;
    int16_t       *out;            // may equal in; nullptr: PCM not written
    float         *f32;            // planar float output or nullptr
    const StreamParam *param;
    VuState       *vu;             // nullptr: no VU
    const uint32_t *nframes;       // per-stream frame counts or nullptr
    uint32_t       frames;         // uniform count when nframes == nullptr
    uint32_t       streams;
    uint32_t       channels;
    uint64_t       stride;         // samples between stream slots (multiple of 8)
    uint64_t       plane;          // floats between planes of the f32 output
    uint32_t       chunks;         // 4 KiB tiles (one wave each) per stream slot
    uint32_t       parity;         // which VuState::samples slot is current
    uint32_t       identity_maps;  // 1 when no stream of the batch has a channel map
    uint32_t       identity_gains; // 1 when no stream of the batch has a gain (disabled or unity everywhere)
    // Completion by flag, for launches of ONE workgroup (the 1 KiB pulls of the per-stream stages): when not
    // null the workgroup, at its very end, makes its stores visible to the host and stores done_seq there
    // (pinned, device-mapped host memory).  The host spins on the word instead of waiting for the stream:
    // 4-5 us less per launch-and-wait on MI355X (tools/ubench_roundtrip.hip).  The launcher clears it
    // when the grid has more than one workgroup.
    uint32_t      *done_flag;
    uint32_t       done_seq;
};

// Tuning knobs of the block kernels' launcher, read from the environment ONCE, when a batch is
// created, and validated there (0 = the built-in choice): $CMHIP_VU_TILE in {4, 8, 16},
// $CMHIP_WIDE4_F32 set at all, $CMHIP_ROWS_RPT in {8, 16, 32, 64} (tools/ab_tiles.py, bench_generic.py).
struct RunTune {
    uint32_t vu_tile;
    uint32_t wide4_f32;
    uint32_t rows_rpt;
    uint32_t fast_nw;              // CMHIP_FAST_NW in {1, 4, 8}: waves per workgroup of the mono / stereo forms with a window
    int32_t  place_env;            // CMHIP_PLACE: -1 unset (only batches created with CMHIP_PLACE_SEARCH search), 0 never,
                                   // 1 the first large batch of a device also without the flag, 2 every large batch
    uint32_t place_debug;          // CMHIP_PLACE_DEBUG: the probe times of the placement search on stderr
    uint32_t no_done_flag;         // CMHIP_NO_DONE_FLAG: one-workgroup launches are waited for through the stream (A/B)
    uint32_t done_spin_us;         // CMHIP_DONE_SPIN_US: how long the host spins on the completion word before it
                                   // waits for the stream instead (default 200; 0 ... 20000)
};

struct EqArgs {
    const int16_t *in;
    int16_t       *out;            // int16 result or nullptr
    float         *f32;            // float result or nullptr
    const StreamParam *param;
    const EqParam *eq;
    EqState       *state;
    VuState       *vu;             // VU of the int16 result, or nullptr
    const uint32_t *nframes;
    uint32_t       frames;
    uint32_t       streams;
    uint32_t       channels;       // every channel of a stream runs the stream's filter, with state of its own
    uint32_t       nsec;           // biquad sections, same for every stream of the batch
    uint32_t       whole_streams;  // keep the channels of a stream in one workgroup (in place + channel maps)
    uint32_t       parity;
    uint64_t       stride;
    uint64_t       plane;
    unsigned long long *dbg;       // 64 words for in-kernel stamps (diagnostic builds only)
    uint32_t      *done_flag;      // as RunArgs::done_flag
    uint32_t       done_seq;
};

struct GenArgs {
    int16_t *dst;
    uint32_t streams, channels, frames;
    uint64_t stride;
    uint32_t seed;
    uint64_t first_global, global_step, frame_offset;
    int16_t  sine[48];
};

// launchers (k_block.hip, k_eq.hip, k_misc.hip)
// (ev_start / ev_stop: optional events that take the kernel's own start and end -- hipExtLaunchKernelGGL
// stamps them from the dispatch itself, without the extra packets of hipEventRecord around the launch)
// (*flagged: the launch was one workgroup and carries the completion flag of RunArgs::done_flag)
hipError_t launch_run(const RunArgs &a, const RunTune &tune, hipStream_t st, hipEvent_t ev_start = nullptr,
                      hipEvent_t ev_stop = nullptr, bool *flagged = nullptr);
// (the first launch of an EQ kernel variant on a device raises its dynamic-LDS limit there:
// prepare_eq does that for a batch's device when the batch is created, launch_eq checks it)
hipError_t prepare_eq(int device);
hipError_t launch_eq(const EqArgs &a, hipStream_t st, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr,
                     bool *flagged = nullptr);
hipError_t launch_generate(const GenArgs &a, int mode, hipStream_t st);
hipError_t launch_node_partial(const VuState *vu, uint32_t streams, uint32_t channels,
                               uint32_t parity, uint64_t first_global, uint64_t global_step,
                               long long *dst_sum, long long *dst_key, bool clear, hipStream_t st,
                               hipEvent_t ev_stop = nullptr);
// (a set of windows -> [1 + 2C][streams] words in pinned, device-mapped host memory; clears the set)
hipError_t launch_vu_pack(VuState *vu, uint32_t streams, uint32_t channels, uint32_t parity,
                          unsigned long long *dst_host_mapped, hipStream_t st, hipEvent_t ev_stop);
hipError_t launch_ceiling(int mode, const void *src, void *dst, size_t bytes,
                          unsigned long long *sink, hipStream_t st);

}  // namespace cmhip

// engine internals shared between cmhip_batch.hip and node.hip (not part of the C ABI)
struct cmhip_batch;
#define CMHIP_INTERNAL __attribute__((visibility("hidden")))
CMHIP_INTERNAL int cmhip_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
CMHIP_INTERNAL int cmhip_batch_node_partial_split(struct cmhip_batch *b, long long *dst_sum, long long *dst_key,
                                                  uint64_t first_global, uint64_t global_step, int clear);
CMHIP_INTERNAL int cmhip_batch_node_partial_side(struct cmhip_batch *b, long long *dst_sum, long long *dst_key,
                                                 uint64_t first_global, uint64_t global_step);
CMHIP_INTERNAL void *cmhip_batch_side_stream(struct cmhip_batch *b);
CMHIP_INTERNAL int cmhip_batch_device(const struct cmhip_batch *b);
CMHIP_INTERNAL unsigned int cmhip_batch_flags(const struct cmhip_batch *b);
