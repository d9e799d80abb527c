/*
 * oracle.h -- CPU restatement of libcoolmic-dsp's transform -> vumeter hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under libcoolmic-dsp_amd/ may include,
 * link or call this.  Allowed callers: tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py -- always as the checker, never as the product.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - integer gain, framing, VU accumulate + dB finalise: PINNED to the vectors
 *     of SURVEY.md section 8(c) (G1..G5, K1..K9), which the survey captured from
 *     the compiled reference; they are committed as tests/golden/survey_8c.json.
 *     The reference itself cannot be rebuilt in this image (it needs libigloo
 *     headers that are absent, and writing a stand-in is not allowed), so no
 *     oracle/_ref exists.
 *   - channel map, int16<->float convert, biquad EQ: PARITY UNPINNED.  The
 *     reference has no such code (only the x/32768.f convert in enc_vorbis.c);
 *     this file is their specification.
 *
 * All "ref:" citations are paths below /root/reference.
 */
#ifndef COOLMIC_ORACLE_H
#define COOLMIC_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include <sys/types.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_MAX_CHANNELS 16

/* error numbers, ref: include/coolmic-dsp/coolmic-dsp.h:36-42 */
#define ORACLE_ERROR_NONE     0
#define ORACLE_ERROR_GENERIC (-1)
#define ORACLE_ERROR_NOSYS   (-8)
#define ORACLE_ERROR_FAULT   (-9)
#define ORACLE_ERROR_INVAL   (-10)

/* ---- pull-read handle (ref: src/iohandle.c:74-113) ------------------------ */
typedef ssize_t (*oracle_read_fn)(void *userdata, void *buffer, size_t len);
typedef int (*oracle_eof_fn)(void *userdata);

typedef struct oracle_handle {
    void *userdata;
    oracle_read_fn read;
    oracle_eof_fn eof;
} oracle_handle_t;

ssize_t oracle_handle_read(oracle_handle_t *h, void *buffer, size_t len);
int oracle_handle_eof(oracle_handle_t *h);

/* ---- gain parameters (ref: src/transform.c:195-222) ----------------------- */
typedef struct oracle_gain {
    uint16_t scale;                       /* 0 = gain disabled */
    uint16_t gain[ORACLE_MAX_CHANNELS];
} oracle_gain_t;

int oracle_gain_set(oracle_gain_t *g, unsigned int stream_channels,
                    unsigned int channels, uint16_t scale, const uint16_t *gain);

/* in-place integer gain (ref: src/transform.c:101-124) */
void oracle_gain_apply(const oracle_gain_t *g, int16_t *samples, size_t frames,
                       unsigned int channels);

/* ---- transform stage with framing (ref: src/transform.c:36-52,126-179) ---- */
typedef struct oracle_transform {
    oracle_handle_t *io;
    unsigned char carry[2 * ORACLE_MAX_CHANNELS - 1];
    size_t carry_fill;
    unsigned int channels;
    oracle_gain_t gain;
} oracle_transform_t;

void oracle_transform_init(oracle_transform_t *t, unsigned int channels, oracle_handle_t *io);
ssize_t oracle_transform_read(void *self, void *buffer, size_t len);   /* oracle_read_fn */
int oracle_transform_eof(void *self);                                  /* oracle_eof_fn */

/* ---- VU meter (ref: src/vumeter.c:35-57,112-218; result vumeter.h:48-83) -- */
typedef struct oracle_vu_result {
    uint_least32_t rate;
    unsigned int channels;
    size_t frames;
    int16_t global_peak;
    double global_power;
    int16_t channel_peak[ORACLE_MAX_CHANNELS];
    double channel_power[ORACLE_MAX_CHANNELS];
} oracle_vu_result_t;

typedef struct oracle_vumeter {
    oracle_handle_t *in;
    uint_least32_t rate;
    unsigned int channels;
    unsigned char buffer[2 * ORACLE_MAX_CHANNELS * 32];
    size_t fill;
    int64_t power[ORACLE_MAX_CHANNELS];
    oracle_vu_result_t result;
} oracle_vumeter_t;

void oracle_vumeter_init(oracle_vumeter_t *v, uint_least32_t rate, unsigned int channels,
                         oracle_handle_t *in);
void oracle_vumeter_reset(oracle_vumeter_t *v);
ssize_t oracle_vumeter_read(oracle_vumeter_t *v, ssize_t maxlen);
int oracle_vumeter_result(oracle_vumeter_t *v, oracle_vu_result_t *out);
/* accumulate a block of whole frames directly (the loop at src/vumeter.c:161-177) */
void oracle_vumeter_accumulate(oracle_vumeter_t *v, const int16_t *samples, size_t frames);
/* dB value of an integer mean square (src/vumeter.c:203-205) */
double oracle_power_db(int64_t sum, uint64_t count);

/* ---- synthetic sources (ref: src/snddev_sine.c:118-150, src/snddev_null.c:33-39) */
typedef struct oracle_sine {
    int16_t table[96];
    size_t len;          /* bytes in one period */
    size_t pos;          /* byte phase */
} oracle_sine_t;

int oracle_sine_table(uint_least32_t rate, int16_t *table, size_t *samples);
int oracle_sine_init(oracle_sine_t *s, uint_least32_t rate);
ssize_t oracle_sine_read(void *self, void *buffer, size_t len);        /* oracle_read_fn */
ssize_t oracle_null_read(void *self, void *buffer, size_t len);        /* oracle_read_fn */

/* memory source: serves a byte array in pieces of at most `chunk` bytes; EOF after */
typedef struct oracle_memsrc {
    const unsigned char *data;
    size_t len, pos, chunk;
} oracle_memsrc_t;
ssize_t oracle_memsrc_read(void *self, void *buffer, size_t len);      /* oracle_read_fn */
int oracle_memsrc_eof(void *self);                                     /* oracle_eof_fn */

/* LCG noise of SURVEY 8(c) G4 / 8(d): s = s*1664525 + 1013904223; x = (int16)(s>>16) */
uint32_t oracle_lcg_fill(uint32_t state, int16_t *out, size_t samples);
/* state after `n` draws starting from `state` (jump-ahead used by the GPU generator) */
uint32_t oracle_lcg_skip(uint32_t state, uint64_t n);

/* ---- extensions: this file is their spec (parity unpinned) ---------------- */
/* out[f][c] = in[f][map[c]] for every frame; in and out must not overlap */
void oracle_chmap_apply(const uint8_t *map, const int16_t *in, int16_t *out, size_t frames,
                        unsigned int channels);
/* interleaved int16 -> planar float, value x/32768.f (ref: src/enc_vorbis.c:108-115);
 * plane c starts at out + c*plane_stride */
void oracle_i16_to_f32_planar(const int16_t *in, float *out, size_t plane_stride, size_t frames,
                              unsigned int channels);
/* float -> int16: round-to-nearest-even of y*32768, saturated, NaN -> 0 */
int16_t oracle_f32_to_i16(float y);

/* One biquad section, Direct Form I, evaluated with this exact operation order:
 *   f = fmaf(b2, x2, fmaf(b1, x1, b0*x0));
 *   y = fmaf(-a1, y1, fmaf(-a2, y2, f));
 * coefficients are a0-normalised floats.  state = {x1, x2, y1, y2}.  */
typedef struct oracle_biquad { float b0, b1, b2, a1, a2; } oracle_biquad_t;
void oracle_biquad_run(const oracle_biquad_t *q, float state[4], const float *in, float *out,
                       size_t n);
/* RBJ cookbook designs computed in double, cast to float.  kind: 0 low shelf,
 * 1 peaking, 2 high shelf.  shelves use slope S = 1; peaking uses Q. */
void oracle_biquad_design(oracle_biquad_t *q, int kind, double rate, double freq, double gain_db,
                          double Q);
/* the 3-band EQ of BASELINE config 3: low shelf 200 Hz +3 dB, peaking 1 kHz -2 dB Q=1,
 * high shelf 6 kHz +2 dB */
void oracle_eq3_design(oracle_biquad_t q[3], double rate);
/* mono stream: int16 -> (optional gain) -> x/32768.f -> nsec biquads -> float out
 * (and int16 out if out_i16 != NULL).  state holds 4 floats per section. */
void oracle_eq_run_mono(const oracle_gain_t *g, const oracle_biquad_t *q, unsigned int nsec,
                        float *state, const int16_t *in, float *out_f32, int16_t *out_i16,
                        size_t n);

/* ---- VU presentation helpers (ref: src/util.c:30-138).  PINNED: the reference's util.c compiles from its
 * own source (oracle/Makefile, target _ref) and these are bit-equal to that build over every peak value and
 * dense grids of powers and colours (tests/test_ref_util.py, tests/golden/ref_util.json) ------------------ */
uint32_t oracle_ahsv2argb(double alpha, double hue, double saturation, double value);
double oracle_power2hue(double power);      /* "default" profile */
double oracle_peak2hue(int16_t peak);       /* "default" profile */

/* ---- timing helpers for bench.py's cpu_baseline leg ----------------------- */
/* Runs `streams` independent noise streams of `frames` frames x `channels` through
 * chmap -> gain -> VU with `threads` pthreads (streams split statically).  Returns
 * seconds spent in the processing loops (generation excluded).  checksum receives
 * a sum over all streams of (sum of squares + peaks) so the work cannot be elided. */
double oracle_bench_block(unsigned int threads, unsigned int streams, unsigned int channels,
                          size_t frames, const uint8_t *map, uint16_t scale,
                          const uint16_t *gain, uint32_t seed0, uint64_t *checksum);
/* Same arithmetic through the pull chain with 1024-byte reads (reference
 * granularity, ref: src/vumeter.c:48), one thread, sine source. */
double oracle_bench_chain(size_t total_frames, uint16_t scale, uint16_t gain0,
                          uint64_t *checksum);

#ifdef __cplusplus
}
#endif
#endif
