/*
 * iohandle.h -- the pull-read operator every stage of the chain speaks.
 * Same contract as the reference (ref: include/coolmic-dsp/iohandle.h:41-66,
 * src/iohandle.c:54-113):
 *
 *   backend read():  <0 error, 0 nothing right now (NOT end of stream), else the
 *                    number of bytes stored, at most len
 *   backend eof():   -1 error, 1 end of stream, 0 otherwise; NULL = endless
 *   backend free():  called once with userdata when the handle dies
 *
 *   coolmic_iohandle_read() calls the backend repeatedly until len bytes are
 *   there, a 0 comes back (returns what it has) or an error comes back (returns
 *   what it has, or the error when it has nothing).  NULL handle or buffer gives
 *   COOLMIC_ERROR_FAULT, len 0 gives 0, a handle without read gives NOSYS.
 */
#ifndef __COOLMIC_DSP_IOHANDLE_H__
#define __COOLMIC_DSP_IOHANDLE_H__

#include <unistd.h>
#include "ro-compat.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct coolmic_iohandle coolmic_iohandle_t;

coolmic_iohandle_t *coolmic_iohandle_new(const char *name, igloo_ro_t associated, void *userdata,
                                         int (*free)(void *), ssize_t (*read)(void *, void *, size_t),
                                         int (*eof)(void *));
ssize_t             coolmic_iohandle_read(coolmic_iohandle_t *self, void *buffer, size_t len);
int                 coolmic_iohandle_eof(coolmic_iohandle_t *self);

#ifdef __cplusplus
}
#endif
#endif
