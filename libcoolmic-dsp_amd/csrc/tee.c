/* tee.c -- byte fan-out (contract: <coolmic-dsp/tee.h>; ref: src/tee.c). */
#define COOLMIC_COMPONENT "libcoolmic-dsp/tee"
#include "host_internal.h"
#include <coolmic-dsp/tee.h>

#include <stdlib.h>
#include <string.h>

#define TEE_BUFFER_MIN 1024
#define TEE_BUFFER_MAX 8192

struct coolmic_tee {
    igloo_ro_base_t __base;
    coolmic_iohandle_t *in;
    size_t readers;
    ssize_t next_reader;
    unsigned char *buffer;
    size_t capacity, fill;
    size_t pos[COOLMIC_DSP_TEE_MAX_READERS];    /* read position of each reader in buffer */
    /* Where the buffered bytes lie in the OUTPUT of an upstream transform (for a VU meter on one of the
     * readers that shares the transform's launch, vumeter.c): buffer[0] is byte `stream_base` of what this tee has
     * pulled since it was attached, and byte x of that is byte x + delta of the transform's output -- as long
     * as nobody else reads the transform's handle in between; `discont` counts the times that broke (or the
     * upstream changed), and whoever relies on the mapping stops relying on it then. */
    uint64_t stream_base;
    int64_t delta;
    int delta_valid;
    unsigned int discont;
};

typedef struct {
    coolmic_tee_t *tee;
    size_t index;
} tee_reader_t;

static void tee_destroy(void *self)
{
    coolmic_tee_t *t = self;
    coolmic_ro_unref(t->in);
    free(t->buffer);
}

COOLMIC_RO_TYPE(coolmic_tee_t, tee_destroy);

coolmic_tee_t *coolmic_tee_new(const char *name, igloo_ro_t associated, size_t readers)
{
    coolmic_tee_t *t;

    if (readers < 1 || readers > COOLMIC_DSP_TEE_MAX_READERS)
        return NULL;
    t = COOLMIC_RO_NEW(coolmic_tee_t, name, associated);
    if (t != NULL)
        t->readers = readers;
    return t;
}

int coolmic_tee_attach_iohandle(coolmic_tee_t *self, coolmic_iohandle_t *handle)
{
    if (self == NULL)
        return COOLMIC_ERROR_FAULT;
    coolmic_ro_unref(self->in);
    self->in = handle;
    coolmic_ro_ref(handle);
    self->delta_valid = 0;
    self->discont++;
    return COOLMIC_ERROR_NONE;
}

/* drop what every reader has seen, grow the buffer towards the request (clamped) */
static void tee_make_room(coolmic_tee_t *t, size_t want)
{
    size_t low = t->fill, i;

    if (want < TEE_BUFFER_MIN)
        want = TEE_BUFFER_MIN;
    else if (want > TEE_BUFFER_MAX)
        want = TEE_BUFFER_MAX;
    if (want > t->capacity) {
        unsigned char *grown = realloc(t->buffer, want);
        if (grown != NULL) {
            t->buffer = grown;
            t->capacity = want;
        } else {
            coolmic_logging_log(COOLMIC_LOGGING_LEVEL_ERROR, COOLMIC_ERROR_NOMEM,
                                "tee: growing the shared buffer failed (out of memory)");
        }
    }
    if (t->buffer == NULL)
        return;
    for (i = 0; i < t->readers; i++)
        if (t->pos[i] < low)
            low = t->pos[i];
    if (low > 0) {
        memmove(t->buffer, t->buffer + low, t->fill - low);
        t->fill -= low;
        t->stream_base += low;
        for (i = 0; i < t->readers; i++)
            t->pos[i] -= low;
    }
}

/* one upstream read into the free part of the buffer; <1 when nothing came */
static ssize_t tee_pull(coolmic_tee_t *t, size_t want)
{
    size_t room;
    ssize_t got;
    struct coolmic_transform *up;
    uint64_t up_off = 0;

    tee_make_room(t, want);
    room = t->capacity - t->fill;
    if (t->buffer == NULL || room == 0) {
        coolmic_logging_log(COOLMIC_LOGGING_LEVEL_ERROR, COOLMIC_ERROR_NOMEM,
                            "tee: no buffer to read upstream into (buffer=%p, room=%zu)", (void *)t->buffer, room);
        return -1;
    }
    if (room > want)
        room = want;
    up = coolmic_iohandle_as_transform(t->in);
    if (up != NULL)
        up_off = coolmic_transform_out_bytes(up);
    got = coolmic_iohandle_read(t->in, t->buffer + t->fill, room);
    if (got < 1)
        return got;
    if (up != NULL) {                  /* these bytes are [up_off, up_off + got) of the transform's output */
        const int64_t d = (int64_t)(up_off - (t->stream_base + t->fill));
        if (t->delta_valid && d != t->delta)
            t->discont++;              /* somebody else took bytes from the transform in between */
        t->delta = d;
        t->delta_valid = 1;
    }
    t->fill += (size_t)got;
    return got;
}

/* exported (not static) so that iohandle.c can recognise reader handles made here */
ssize_t coolmic_tee_reader_read(void *userdata, void *buffer, size_t len)
{
    tee_reader_t *r = userdata;
    coolmic_tee_t *t = r->tee;
    unsigned char *dst = buffer;
    size_t done = 0;

    while (len) {
        size_t have = t->fill - t->pos[r->index];
        if (have == 0) {
            if (tee_pull(t, len) < 1)
                break;
            have = t->fill - t->pos[r->index];
            if (have == 0)
                break;
        }
        if (have > len)
            have = len;
        memcpy(dst, t->buffer + t->pos[r->index], have);
        t->pos[r->index] += have;
        dst += have;
        done += have;
        len -= have;
    }
    return (ssize_t)done;
}

static int tee_reader_eof(void *userdata)
{
    tee_reader_t *r = userdata;
    if (r->tee->pos[r->index] < r->tee->fill)
        return 0;
    return coolmic_iohandle_eof(r->tee->in);
}

static int tee_reader_free(void *userdata)
{
    tee_reader_t *r = userdata;
    coolmic_ro_unref(r->tee);
    free(r);
    return 0;
}

coolmic_iohandle_t *coolmic_tee_get_iohandle(coolmic_tee_t *self, ssize_t index)
{
    tee_reader_t *r;
    coolmic_iohandle_t *h;

    if (self == NULL)
        return NULL;
    if (index == -1)
        index = self->next_reader;
    if (index < 0 || (size_t)index >= self->readers)
        return NULL;               /* the reference checks against 4, not the reader count */
    self->next_reader = index + 1;

    r = calloc(1, sizeof(*r));
    if (r == NULL)
        return NULL;
    coolmic_ro_ref(self);
    r->tee = self;
    r->index = (size_t)index;
    h = coolmic_iohandle_new(NULL, igloo_RO_NULL, r, tee_reader_free, coolmic_tee_reader_read, tee_reader_eof);
    if (h == NULL)
        tee_reader_free(r);
    return h;
}

/* ---- for a VU meter on a reader handle (internal: vumeter.c; declared in host_internal.h) -------- */

/* the transform right above the tee of reader handle `userdata` (NULL: none), the position of that reader's
 * next byte in the transform's output, and the discontinuity count to compare later */
struct coolmic_transform *coolmic_tee_reader_upstream(void *userdata, uint64_t *next_off, unsigned int *discont)
{
    tee_reader_t *r = userdata;
    coolmic_tee_t *t = r->tee;
    struct coolmic_transform *up = coolmic_iohandle_as_transform(t->in);

    if (up == NULL)
        return NULL;
    if (!t->delta_valid) {             /* nothing pulled from this upstream yet: the next pull continues its output */
        t->delta = (int64_t)(coolmic_transform_out_bytes(up) - (t->stream_base + t->fill));
        t->delta_valid = 1;
    }
    *next_off = (uint64_t)((int64_t)(t->stream_base + t->pos[r->index]) + t->delta);
    *discont = t->discont;
    return up;
}

unsigned int coolmic_tee_reader_discont(void *userdata)
{
    tee_reader_t *r = userdata;
    return r->tee->discont;
}
