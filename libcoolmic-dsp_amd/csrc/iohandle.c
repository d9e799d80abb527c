/* iohandle.c -- the pull-read handle (contract: <coolmic-dsp/iohandle.h>;
 * ref: src/iohandle.c:31-113). */
#include "host_internal.h"

struct coolmic_iohandle {
    igloo_ro_base_t __base;
    void *userdata;
    int (*free_fn)(void *userdata);
    ssize_t (*read_fn)(void *userdata, void *buffer, size_t len);
    int (*eof_fn)(void *userdata);
};

static void iohandle_destroy(void *self)
{
    coolmic_iohandle_t *h = self;
    if (h->free_fn != NULL)
        h->free_fn(h->userdata);             /* backend cleanup, once */
}

COOLMIC_RO_TYPE(coolmic_iohandle_t, iohandle_destroy);

coolmic_iohandle_t *coolmic_iohandle_new(const char *name, igloo_ro_t associated, void *userdata,
                                         int (*free)(void *), ssize_t (*read)(void *, void *, size_t),
                                         int (*eof)(void *))
{
    coolmic_iohandle_t *h;

    if (read == NULL)                        /* a handle nobody can read is refused */
        return NULL;
    h = COOLMIC_RO_NEW(coolmic_iohandle_t, name, associated);
    if (h == NULL)
        return NULL;
    h->userdata = userdata;
    h->free_fn = free;
    h->read_fn = read;
    h->eof_fn = eof;
    return h;
}

/* (internal: what a handle reads from -- two handles over the same userdata share their backend's state) */
const void *coolmic_iohandle_backend(const coolmic_iohandle_t *self)
{
    return self != NULL ? self->userdata : NULL;
}

ssize_t coolmic_iohandle_read(coolmic_iohandle_t *self, void *buffer, size_t len)
{
    unsigned char *dst = buffer;
    size_t total = 0;

    if (self == NULL || buffer == NULL)
        return COOLMIC_ERROR_FAULT;
    if (len == 0)
        return 0;
    if (self->read_fn == NULL)
        return COOLMIC_ERROR_NOSYS;

    while (total < len) {
        ssize_t got = self->read_fn(self->userdata, dst + total, len - total);
        if (got < 0)
            return total ? (ssize_t)total : got;
        if (got == 0)
            break;
        total += (size_t)got;
    }
    return (ssize_t)total;
}

int coolmic_iohandle_eof(coolmic_iohandle_t *self)
{
    if (self == NULL)
        return COOLMIC_ERROR_FAULT;
    return self->eof_fn ? self->eof_fn(self->userdata) : 0;
}

/* lets a downstream stage of this library recognise a transform's handle */
struct coolmic_transform *coolmic_iohandle_as_transform(coolmic_iohandle_t *h)
{
    if (h != NULL && h->read_fn == coolmic_transform_handle_read)
        return h->userdata;
    return NULL;
}

/* ... and a tee's reader handle: -> the reader's userdata for coolmic_tee_reader_upstream() */
void *coolmic_iohandle_as_tee_reader(coolmic_iohandle_t *h)
{
    if (h != NULL && h->read_fn == coolmic_tee_reader_read)
        return h->userdata;
    return NULL;
}
