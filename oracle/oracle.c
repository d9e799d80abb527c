/*
 * oracle.c -- CPU restatement of libcoolmic-dsp's transform -> vumeter hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Scalar C, same arithmetic as the
 * reference: 64-bit multiply + truncating divide + saturate for the gain,
 * strict-greater first-max peak and int64 sum of squares for the VU meter,
 * integer mean then 20*log10(sqrt(p)/32768) clamped to <= 0 in double.
 * Build with -ffp-contract=off (oracle/Makefile does).
 *
 * PARITY UNPINNED for the hot path.  The reference ships no tests, fixtures or golden vectors, and
 * the path's files cannot be built in this image (its transform.c / vumeter.c / iohandle.c need libigloo's
 * headers; a build against written stand-ins is not a build of the reference), so there is no
 * oracle/_ref for them.  (The one exception is at the path's edge: src/util.c, the VU colour helpers, compiles
 * from its own source -- oracle/_ref/libref_util.so -- and the three functions at the end of this file are held
 * bit for bit against that build and the vectors taken from it: tests/test_ref_util.py.)  What the rest IS
 * checked against, bit for bit: the vectors of SURVEY.md 8(c)
 * (tests/golden/survey_8c.json, tests/test_oracle_golden.py) -- outputs the survey stage captured
 * from such a stand-in build, good evidence but not a pin -- and a second restatement in pure
 * Python written from the reference's text (tests/test_second_witness.py).  Channel map, float
 * converts and the biquad EQ have no counterpart in the reference at all: this file is their spec.
 */
#define _GNU_SOURCE
#include "oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------- */
/* pull-read handle.  ref: src/iohandle.c:74-104 (read loop), :106-113 (eof)  */

ssize_t oracle_handle_read(oracle_handle_t *h, void *buffer, size_t len)
{
    unsigned char *dst = buffer;
    ssize_t total = 0;

    if (h == NULL || buffer == NULL)
        return ORACLE_ERROR_FAULT;          /* iohandle.c:79-80 */
    if (len == 0)
        return 0;                           /* iohandle.c:81-82 */
    if (h->read == NULL)
        return ORACLE_ERROR_NOSYS;          /* iohandle.c:83-84 */

    /* keep asking the backend until satisfied; a 0 ends the call with what we
     * have, a negative ends it with what we have or, with nothing, the error */
    while (len > 0) {
        ssize_t got = h->read(h->userdata, dst, len);
        if (got < 0)
            return total ? total : got;
        if (got == 0)
            break;
        dst += got;
        len -= (size_t)got;
        total += got;
    }
    return total;
}

int oracle_handle_eof(oracle_handle_t *h)
{
    if (h == NULL)
        return ORACLE_ERROR_FAULT;
    return h->eof ? h->eof(h->userdata) : 0;
}

/* ------------------------------------------------------------------------- */
/* gain parameters.  ref: src/transform.c:195-222                             */

int oracle_gain_set(oracle_gain_t *g, unsigned int stream_channels, unsigned int channels,
                    uint16_t scale, const uint16_t *gain)
{
    unsigned int c;

    if (g == NULL)
        return ORACLE_ERROR_FAULT;
    if (channels == 0 || scale == 0 || gain == NULL) {   /* :200-203 disable */
        g->scale = 0;
        return ORACLE_ERROR_NONE;
    }
    if (channels == stream_channels) {                   /* :205-208 one per channel */
        for (c = 0; c < channels && c < ORACLE_MAX_CHANNELS; c++)
            g->gain[c] = gain[c];
    } else if (channels == 1) {                          /* :209-213 broadcast */
        for (c = 0; c < stream_channels && c < ORACLE_MAX_CHANNELS; c++)
            g->gain[c] = gain[0];
    } else if (channels == 2 && stream_channels == 1) {  /* :214-218 stereo -> mono mean */
        g->gain[0] = (uint16_t)(((uint32_t)gain[0] + (uint32_t)gain[1]) / 2u);
    } else {
        return ORACLE_ERROR_INVAL;                       /* :219-221 old params kept */
    }
    g->scale = scale;
    return ORACLE_ERROR_NONE;
}

/* ref: src/transform.c:101-124 */
void oracle_gain_apply(const oracle_gain_t *g, int16_t *samples, size_t frames,
                       unsigned int channels)
{
    size_t n = frames * channels, i;
    unsigned int c = 0;

    if (g->scale == 0)                                   /* :107-108 */
        return;
    for (i = 0; i < n; i++) {
        int64_t v = (int64_t)samples[i] * (int64_t)g->gain[c];
        v /= (int64_t)g->scale;                          /* C division: toward zero */
        if (v >= 32767)
            v = 32767;
        else if (v <= -32768)
            v = -32768;
        samples[i] = (int16_t)v;
        if (++c == channels)
            c = 0;
    }
}

/* ------------------------------------------------------------------------- */
/* transform stage.  ref: src/transform.c:126-165 (read), :167-179 (eof)      */

void oracle_transform_init(oracle_transform_t *t, unsigned int channels, oracle_handle_t *io)
{
    memset(t, 0, sizeof(*t));
    t->channels = channels;
    t->io = io;
}

ssize_t oracle_transform_read(void *self, void *buffer, size_t len)
{
    oracle_transform_t *t = self;
    unsigned char *dst = buffer;
    const size_t framesize = 2u * t->channels;
    size_t have = 0, tail;
    ssize_t got;

    len -= len % framesize;                              /* :133-134 */
    if (len == 0)
        return 0;                                        /* :136-137 */

    if (t->carry_fill) {                                 /* :141-148 */
        memcpy(dst, t->carry, t->carry_fill);
        have = t->carry_fill;
        t->carry_fill = 0;
    }

    got = oracle_handle_read(t->io, dst + have, len - have);   /* :150, one call */
    if (got > 0)
        have += (size_t)got;                             /* negatives are swallowed */

    tail = have % framesize;                             /* :155-160 */
    if (tail) {
        memcpy(t->carry, dst + have - tail, tail);
        t->carry_fill = tail;
        have -= tail;
    }

    oracle_gain_apply(&t->gain, (int16_t *)buffer, have / framesize, t->channels);  /* :162 */
    return (ssize_t)have;
}

int oracle_transform_eof(void *self)
{
    oracle_transform_t *t = self;
    if (t->io == NULL)
        return 1;
    return oracle_handle_eof(t->io);
}

/* ------------------------------------------------------------------------- */
/* VU meter.  ref: src/vumeter.c                                              */

void oracle_vumeter_reset(oracle_vumeter_t *v)           /* :88-99 */
{
    memset(v->power, 0, sizeof(v->power));
    memset(&v->result, 0, sizeof(v->result));
    v->result.rate = v->rate;
    v->result.channels = v->channels;
    /* note: the byte buffer fill is NOT reset, as in the reference */
}

void oracle_vumeter_init(oracle_vumeter_t *v, uint_least32_t rate, unsigned int channels,
                         oracle_handle_t *in)
{
    memset(v, 0, sizeof(*v));
    v->rate = rate;
    v->channels = channels;
    v->in = in;
    oracle_vumeter_reset(v);
}

/* the per-sample loop of src/vumeter.c:161-177 */
void oracle_vumeter_accumulate(oracle_vumeter_t *v, const int16_t *samples, size_t frames)
{
    size_t f;
    unsigned int c;

    for (f = 0; f < frames; f++) {
        for (c = 0; c < v->channels; c++) {
            int x = *samples++;
            if (abs(x) > abs((int)v->result.channel_peak[c])) {   /* strict: first max wins */
                v->result.channel_peak[c] = (int16_t)x;
                if (abs(x) > abs((int)v->result.global_peak))
                    v->result.global_peak = (int16_t)x;
            }
            v->power[c] += (int64_t)x * (int64_t)x;
        }
    }
    v->result.frames += frames;
}

ssize_t oracle_vumeter_read(oracle_vumeter_t *v, ssize_t maxlen)
{
    size_t want, framesize, frames, used;
    ssize_t got, ret;

    if (v == NULL)
        return -1;                                       /* :148-151 */

    /* physical read, :112-136 */
    want = sizeof(v->buffer) - v->fill;
    if (maxlen >= 0 && want > (size_t)maxlen)
        want = (size_t)maxlen;
    got = oracle_handle_read(v->in, v->buffer + v->fill, want);
    if (got < 0) {
        /* the reference tests for exactly -1 (:127-131); other negative codes
         * would corrupt its fill counter -- treated the same way here */
        ret = v->fill ? 0 : got;
    } else {
        v->fill += (size_t)got;
        ret = got;
    }

    framesize = 2u * v->channels;
    frames = v->fill / framesize;
    oracle_vumeter_accumulate(v, (const int16_t *)v->buffer, frames);

    used = frames * framesize;                           /* :179-184 */
    if (used < v->fill)
        memmove(v->buffer, v->buffer + used, v->fill - used);
    v->fill -= used;
    return ret;
}

double oracle_power_db(int64_t sum, uint64_t count)
{
    double p = (double)((uint64_t)sum / count);          /* integer mean first */
    p = 20. * log10(sqrt(p) / 32768.);
    return fmin(p, 0.);
}

int oracle_vumeter_result(oracle_vumeter_t *v, oracle_vu_result_t *out)   /* :189-218 */
{
    unsigned int c;
    int64_t all = 0;

    if (v == NULL || out == NULL)
        return ORACLE_ERROR_FAULT;
    if (v->result.frames == 0)
        return ORACLE_ERROR_INVAL;

    for (c = 0; c < v->channels; c++) {
        all += v->power[c];
        v->result.channel_power[c] = oracle_power_db(v->power[c], (uint64_t)v->result.frames);
    }
    v->result.global_power =
        oracle_power_db(all, (uint64_t)(v->result.frames * (size_t)v->channels));

    *out = v->result;
    oracle_vumeter_reset(v);
    return ORACLE_ERROR_NONE;
}

/* ------------------------------------------------------------------------- */
/* synthetic sources                                                          */

/* ref: src/snddev_sine.c:36-99 holds one literal period per rate.  Every one of
 * those tables equals trunc(A*sin(2*pi*k/N)) for any A in [32766.938, 32767);
 * tests/golden/make_fixtures.py re-checks that against the source text. */
#define ORACLE_SINE_AMPLITUDE 32766.97

int oracle_sine_table(uint_least32_t rate, int16_t *table, size_t *samples)
{
    static const uint_least32_t rates[] = {8000, 16000, 24000, 32000, 44000, 44100, 48000, 96000};
    size_t i, n = 0;

    for (i = 0; i < sizeof(rates) / sizeof(rates[0]); i++)
        if (rates[i] == rate)
            n = rate / 1000;                             /* snddev_sine.c:184 */
    if (n == 0)
        return ORACLE_ERROR_NOSYS;                       /* :175-177 */
    for (i = 0; i < n; i++)
        table[i] = (int16_t)trunc(ORACLE_SINE_AMPLITUDE * sin(2. * M_PI * (double)i / (double)n));
    *samples = n;
    return ORACLE_ERROR_NONE;
}

int oracle_sine_init(oracle_sine_t *s, uint_least32_t rate)
{
    size_t n;
    int rc = oracle_sine_table(rate, s->table, &n);
    if (rc != ORACLE_ERROR_NONE)
        return rc;
    s->len = 2 * n;
    s->pos = 0;
    return ORACLE_ERROR_NONE;
}

/* byte-granular endless repetition of the period; ref: src/snddev_sine.c:118-150 */
ssize_t oracle_sine_read(void *self, void *buffer, size_t len)
{
    oracle_sine_t *s = self;
    const unsigned char *period = (const unsigned char *)s->table;
    unsigned char *dst = buffer;
    size_t i, p = s->pos;

    for (i = 0; i < len; i++) {
        dst[i] = period[p];
        if (++p == s->len)
            p = 0;
    }
    s->pos = p;
    return (ssize_t)len;
}

ssize_t oracle_null_read(void *self, void *buffer, size_t len)   /* src/snddev_null.c:33-39 */
{
    (void)self;
    memset(buffer, 0, len);
    return (ssize_t)len;
}

ssize_t oracle_memsrc_read(void *self, void *buffer, size_t len)
{
    oracle_memsrc_t *m = self;
    size_t n = m->len - m->pos;

    if (n > len)
        n = len;
    if (m->chunk && n > m->chunk)
        n = m->chunk;
    memcpy(buffer, m->data + m->pos, n);
    m->pos += n;
    return (ssize_t)n;
}

int oracle_memsrc_eof(void *self)
{
    oracle_memsrc_t *m = self;
    return m->pos >= m->len;
}

uint32_t oracle_lcg_fill(uint32_t state, int16_t *out, size_t samples)
{
    size_t i;
    for (i = 0; i < samples; i++) {
        state = state * 1664525u + 1013904223u;
        out[i] = (int16_t)(state >> 16);
    }
    return state;
}

uint32_t oracle_lcg_skip(uint32_t state, uint64_t n)
{
    /* compose x -> a*x + c with itself by binary powers */
    uint32_t a = 1664525u, c = 1013904223u;
    while (n) {
        if (n & 1)
            state = state * a + c;
        c = c * (a + 1u);
        a = a * a;
        n >>= 1;
    }
    return state;
}

/* ------------------------------------------------------------------------- */
/* extensions (spec; parity unpinned)                                         */

void oracle_chmap_apply(const uint8_t *map, const int16_t *in, int16_t *out, size_t frames,
                        unsigned int channels)
{
    size_t f;
    unsigned int c;
    for (f = 0; f < frames; f++)
        for (c = 0; c < channels; c++)
            out[f * channels + c] = in[f * channels + map[c]];
}

void oracle_i16_to_f32_planar(const int16_t *in, float *out, size_t plane_stride, size_t frames,
                              unsigned int channels)
{
    size_t f;
    unsigned int c;
    for (f = 0; f < frames; f++)
        for (c = 0; c < channels; c++)
            out[c * plane_stride + f] = *in++ / 32768.f;     /* enc_vorbis.c:112 */
}

int16_t oracle_f32_to_i16(float y)
{
    float v = y * 32768.f;
    if (v != v)
        return 0;
    v = rintf(v);                       /* round-to-nearest-even in the default mode */
    if (v >= 32767.f)
        return 32767;
    if (v <= -32768.f)
        return -32768;
    return (int16_t)v;
}

void oracle_biquad_run(const oracle_biquad_t *q, float state[4], const float *in, float *out,
                       size_t n)
{
    float x1 = state[0], x2 = state[1], y1 = state[2], y2 = state[3];
    const float na1 = -q->a1, na2 = -q->a2;
    size_t i;

    for (i = 0; i < n; i++) {
        float x0 = in[i];
        float f = fmaf(q->b2, x2, fmaf(q->b1, x1, q->b0 * x0));
        float y = fmaf(na1, y1, fmaf(na2, y2, f));
        x2 = x1; x1 = x0;
        y2 = y1; y1 = y;
        out[i] = y;
    }
    state[0] = x1; state[1] = x2; state[2] = y1; state[3] = y2;
}

void oracle_biquad_design(oracle_biquad_t *q, int kind, double rate, double freq, double gain_db,
                          double Q)
{
    const double A = pow(10., gain_db / 40.);
    const double w0 = 2. * M_PI * freq / rate;
    const double cw = cos(w0), sw = sin(w0);
    double b0, b1, b2, a0, a1, a2;

    if (kind == 1) {                    /* peaking */
        const double alpha = sw / (2. * Q);
        b0 = 1. + alpha * A;  b1 = -2. * cw;  b2 = 1. - alpha * A;
        a0 = 1. + alpha / A;  a1 = -2. * cw;  a2 = 1. - alpha / A;
    } else {                            /* shelves, slope S = 1 */
        const double alpha = sw / 2. * sqrt(2.);
        const double k = 2. * sqrt(A) * alpha;
        if (kind == 0) {                /* low shelf */
            b0 = A * ((A + 1.) - (A - 1.) * cw + k);
            b1 = 2. * A * ((A - 1.) - (A + 1.) * cw);
            b2 = A * ((A + 1.) - (A - 1.) * cw - k);
            a0 = (A + 1.) + (A - 1.) * cw + k;
            a1 = -2. * ((A - 1.) + (A + 1.) * cw);
            a2 = (A + 1.) + (A - 1.) * cw - k;
        } else {                        /* high shelf */
            b0 = A * ((A + 1.) + (A - 1.) * cw + k);
            b1 = -2. * A * ((A - 1.) + (A + 1.) * cw);
            b2 = A * ((A + 1.) + (A - 1.) * cw - k);
            a0 = (A + 1.) - (A - 1.) * cw + k;
            a1 = 2. * ((A - 1.) - (A + 1.) * cw);
            a2 = (A + 1.) - (A - 1.) * cw - k;
        }
    }
    q->b0 = (float)(b0 / a0);
    q->b1 = (float)(b1 / a0);
    q->b2 = (float)(b2 / a0);
    q->a1 = (float)(a1 / a0);
    q->a2 = (float)(a2 / a0);
}

void oracle_eq3_design(oracle_biquad_t q[3], double rate)
{
    oracle_biquad_design(&q[0], 0, rate, 200., 3., 0.);
    oracle_biquad_design(&q[1], 1, rate, 1000., -2., 1.);
    oracle_biquad_design(&q[2], 2, rate, 6000., 2., 0.);
}

void oracle_eq_run_mono(const oracle_gain_t *g, const oracle_biquad_t *q, unsigned int nsec,
                        float *state, const int16_t *in, float *out_f32, int16_t *out_i16,
                        size_t n)
{
    size_t i;
    unsigned int s;

    for (i = 0; i < n; i++) {
        int16_t xi = in[i];
        float v;
        if (g && g->scale)
            oracle_gain_apply(g, &xi, 1, 1);
        v = xi / 32768.f;
        for (s = 0; s < nsec; s++) {
            float y;
            oracle_biquad_run(&q[s], state + 4 * s, &v, &y, 1);
            v = y;
        }
        if (out_f32)
            out_f32[i] = v;
        if (out_i16)
            out_i16[i] = oracle_f32_to_i16(v);
    }
}

/* ------------------------------------------------------------------------- */
/* VU presentation helpers.  ref: src/util.c:30-138                           */

static uint32_t oracle_component(double x)               /* :30-44 */
{
    uint32_t v;
    if (x >= 1.)
        x = 1.;
    else if (x <= 0.)
        x = 0.;
    v = (uint32_t)(x * 255.);
    if (v > 255)
        v = 255;
    return v;
}

uint32_t oracle_ahsv2argb(double alpha, double hue, double saturation, double value)
{
    int h1 = (int)(double)(hue / (M_PI / 3.));           /* :60 */
    double f = hue - (double)h1;                          /* :61, sextant number subtracted */
    double p = value * (1. - saturation);
    double q = value * (1. - saturation * f);
    double t = value * (1. - saturation * (1. - f));
    double rgb[3] = {0., 0., 0.};

    if (h1 == 0 || h1 == 6) { rgb[0] = value; rgb[1] = t; rgb[2] = p; }
    else if (h1 == 1)       { rgb[0] = q; rgb[1] = value; rgb[2] = p; }
    else if (h1 == 2)       { rgb[0] = p; rgb[1] = value; rgb[2] = t; }
    else if (h1 == 3)       { rgb[0] = p; rgb[1] = q; rgb[2] = value; }
    else if (h1 == 4)       { rgb[0] = t; rgb[1] = p; rgb[2] = value; }
    else if (h1 == 5)       { rgb[0] = value; rgb[1] = p; rgb[2] = q; }
    return (oracle_component(alpha) << 24) + (oracle_component(rgb[0]) << 16) +
           (oracle_component(rgb[1]) << 8) + oracle_component(rgb[2]);
}

double oracle_power2hue(double power)                    /* :108-120 */
{
    if (power < -20.)
        return M_PI * 2. / 3.;
    else if (power >= 0)
        return 0;
    return pow(sin(M_PI * power / 40.), 2.) * M_PI * 2. / 3.;
}

double oracle_peak2hue(int16_t peak)                     /* :124-138 */
{
    if (peak == -32768 || peak == 32767)
        return 0.;
    else if (peak < -30000 || peak > 30000)
        return 0.43;
    else if (peak < -28000 || peak > 28000)
        return 1.;
    return M_PI * 2. / 3.;
}

/* ------------------------------------------------------------------------- */
/* timing helpers for the cpu_baseline leg of bench.py                        */

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

typedef struct bench_job {
    unsigned int first, count, channels;
    size_t frames;
    const uint8_t *map;
    oracle_gain_t gain;
    uint32_t seed0;
    double busy;
    uint64_t checksum;
} bench_job_t;

static void *bench_worker(void *arg)
{
    bench_job_t *j = arg;
    const size_t n = j->frames * j->channels;
    int16_t *raw = malloc(n * sizeof(int16_t));
    int16_t *pcm = malloc(n * sizeof(int16_t));
    unsigned int s, c;

    j->busy = 0.;
    j->checksum = 0;
    if (!raw || !pcm) {
        free(raw); free(pcm);
        return NULL;
    }
    for (s = 0; s < j->count; s++) {
        oracle_vumeter_t vu;
        oracle_vu_result_t res;
        double t0;

        oracle_lcg_fill(j->seed0 + j->first + s, raw, n);    /* generation is not timed */
        t0 = now_s();
        if (j->map)
            oracle_chmap_apply(j->map, raw, pcm, j->frames, j->channels);
        else
            memcpy(pcm, raw, n * sizeof(int16_t));
        oracle_gain_apply(&j->gain, pcm, j->frames, j->channels);
        oracle_vumeter_init(&vu, 48000, j->channels, NULL);
        oracle_vumeter_accumulate(&vu, pcm, j->frames);
        for (c = 0; c < j->channels; c++)
            j->checksum += (uint64_t)vu.power[c] + (uint64_t)(uint16_t)vu.result.channel_peak[c];
        oracle_vumeter_result(&vu, &res);
        j->busy += now_s() - t0;
    }
    free(raw);
    free(pcm);
    return NULL;
}

double oracle_bench_block(unsigned int threads, unsigned int streams, unsigned int channels,
                          size_t frames, const uint8_t *map, uint16_t scale,
                          const uint16_t *gain, uint32_t seed0, uint64_t *checksum)
{
    pthread_t *tid;
    bench_job_t *jobs;
    unsigned int t, first = 0;
    double worst = 0.;
    uint64_t sum = 0;

    if (threads == 0)
        threads = 1;
    if (threads > streams)
        threads = streams;
    tid = calloc(threads, sizeof(*tid));
    jobs = calloc(threads, sizeof(*jobs));
    for (t = 0; t < threads; t++) {
        bench_job_t *j = &jobs[t];
        j->first = first;
        j->count = streams / threads + (t < streams % threads ? 1u : 0u);
        first += j->count;
        j->channels = channels;
        j->frames = frames;
        j->map = map;
        j->seed0 = seed0;
        oracle_gain_set(&j->gain, channels, channels, scale, gain);
        pthread_create(&tid[t], NULL, bench_worker, j);
    }
    for (t = 0; t < threads; t++) {
        pthread_join(tid[t], NULL);
        if (jobs[t].busy > worst)
            worst = jobs[t].busy;
        sum += jobs[t].checksum;
    }
    if (checksum)
        *checksum = sum;
    free(tid);
    free(jobs);
    return worst;
}

double oracle_bench_chain(size_t total_frames, uint16_t scale, uint16_t gain0, uint64_t *checksum)
{
    oracle_sine_t sine;
    oracle_handle_t hsrc, htr;
    oracle_transform_t tr;
    oracle_vumeter_t vu;
    oracle_vu_result_t res;
    size_t done = 0, reads = 0;
    uint64_t sum = 0;
    double t0;

    oracle_sine_init(&sine, 48000);
    hsrc.userdata = &sine; hsrc.read = oracle_sine_read; hsrc.eof = NULL;
    oracle_transform_init(&tr, 1, &hsrc);
    oracle_gain_set(&tr.gain, 1, 1, scale, &gain0);
    htr.userdata = &tr; htr.read = oracle_transform_read; htr.eof = oracle_transform_eof;
    oracle_vumeter_init(&vu, 48000, 1, &htr);

    t0 = now_s();
    while (done < total_frames) {
        ssize_t got = oracle_vumeter_read(&vu, -1);
        if (got <= 0)
            break;
        done += (size_t)got / 2;
        if (++reads % 20 == 0) {            /* ref: src/simple.c:370 default interval */
            sum += (uint64_t)vu.power[0];
            oracle_vumeter_result(&vu, &res);
        }
    }
    sum += (uint64_t)vu.power[0];
    if (checksum)
        *checksum = sum;
    return now_s() - t0;
}
