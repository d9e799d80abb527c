/*
 * tee.h -- one upstream handle, up to four downstream handles that each see every byte.
 * Same API and behaviour as the reference's tee (ref: include/coolmic-dsp/tee.h:36-45,
 * src/tee.c:83-289): a shared buffer of 1024..8192 bytes, one read position per reader;
 * a reader that has consumed everything pulls more from upstream; when the slowest reader
 * lags by a full buffer the faster ones get short (possibly empty) reads until it moves.
 * Pure byte plumbing between transform and {encoder, vumeter}
 * (ref: src/simple.c:196,217-229) -- no arithmetic, stays on the CPU.
 */
#ifndef __COOLMIC_DSP_TEE_H__
#define __COOLMIC_DSP_TEE_H__

#include <sys/types.h>
#include "ro-compat.h"
#include "iohandle.h"

#ifdef __cplusplus
extern "C" {
#endif

#define COOLMIC_DSP_TEE_MAX_READERS 4

typedef struct coolmic_tee coolmic_tee_t;

/* NULL unless 1 <= readers <= 4 */
coolmic_tee_t      *coolmic_tee_new(const char *name, igloo_ro_t associated, size_t readers);
int                 coolmic_tee_attach_iohandle(coolmic_tee_t *self, coolmic_iohandle_t *handle);
/* handle of reader `index`; -1 = the one after the last handed out.  NULL when out of range. */
coolmic_iohandle_t *coolmic_tee_get_iohandle(coolmic_tee_t *self, ssize_t index);

#ifdef __cplusplus
}
#endif
#endif
