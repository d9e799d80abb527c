/*
 * ro-compat.h -- the small part of libigloo's refcounted-object ("RO") surface
 * that the transform/vumeter/iohandle/snddev API needs.
 *
 * The reference passes an `igloo_ro_t associated` to every constructor and lets
 * callers hand objects to igloo_ro_ref()/igloo_ro_unref()
 * (ref: include/coolmic-dsp/iohandle.h:54, src/simple.c:212-229).  libigloo is not
 * part of this repository's environment, so this header supplies source-level
 * equivalents backed by coolmic_ro_* functions in libcoolmic-dsp-hip.so:
 *
 *   - every object starts with a coolmic_ro_base_t header, is zero-filled and is
 *     born with one reference (relied on at src/transform.c:72, src/vumeter.c:76);
 *   - ref/unref of NULL is a harmless error (relied on at src/transform.c:57,90);
 *   - the type's free callback runs when the last reference goes, then the
 *     memory is released.
 *
 * Build with -DCOOLMIC_DSP_USE_LIBIGLOO to take the names from a real libigloo
 * instead (then link it too); the coolmic_* API keeps the same signatures.
 */
#ifndef __COOLMIC_DSP_RO_COMPAT_H__
#define __COOLMIC_DSP_RO_COMPAT_H__

#ifdef COOLMIC_DSP_USE_LIBIGLOO
#include <igloo/ro.h>
#else

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct coolmic_ro_type {
    const char *name;                 /* type name, for diagnostics */
    size_t size;                      /* allocation size, header included */
    void (*free_cb)(void *self);      /* may be NULL */
} coolmic_ro_type_t;

typedef struct coolmic_ro_base {
    const coolmic_ro_type_t *type;
    unsigned int refc;
    char *name;
    void *associated;                 /* holds a reference while we live */
} coolmic_ro_base_t;

void *coolmic_ro_new_raw(const coolmic_ro_type_t *type, const char *name, void *associated);
int   coolmic_ro_ref(void *self);     /* COOLMIC_ERROR_NONE or COOLMIC_ERROR_FAULT */
int   coolmic_ro_unref(void *self);
unsigned int coolmic_ro_refcount(void *self);   /* diagnostics / tests */

/* libigloo spellings used by callers of the reference API */
typedef void *igloo_ro_t;
typedef coolmic_ro_base_t igloo_ro_base_t;
#define igloo_RO_NULL            ((igloo_ro_t)NULL)
#define igloo_ro_ref(x)          coolmic_ro_ref((void *)(x))
#define igloo_ro_unref(x)        coolmic_ro_unref((void *)(x))

#ifdef __cplusplus
}
#endif
#endif /* COOLMIC_DSP_USE_LIBIGLOO */
#endif
