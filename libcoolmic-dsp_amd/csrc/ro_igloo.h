/* ro_igloo.h -- the libigloo back end of ro_glue.h: only with -DCOOLMIC_DSP_USE_LIBIGLOO, i.e. when this
 * library's transform.c / vumeter.c / iohandle.c / tee.c are compiled INTO the reference's build in place of
 * its own (`make dropin IGLOO=1 REF=<reference tree>`).  Never part of the default build,
 * and csrc/ro.c is left out of that one.
 *
 * Every igloo identifier used here is one the reference's own sources use, with the same number of arguments
 * (tests/test_abi.py::test_igloo_glue_uses_the_references_igloo_surface holds this file against them):
 *
 *   type registration   igloo_RO_PUBLIC_TYPE(type, igloo_RO_TYPEDECL_FREE(cb))   ref: src/transform.c:60-62,
 *                                                    src/vumeter.c:65-67, src/iohandle.c:50-52, src/tee.c:79-81
 *   destructor          static void cb(igloo_ro_t self) + igloo_RO_TO_TYPE(self, type)   ref: src/transform.c:54-58
 *   allocation          igloo_ro_new_raw(type, name, associated)   ref: src/transform.c:72, src/vumeter.c:76,
 *                                                    src/iohandle.c:62, src/tee.c:227
 *   references          igloo_ro_ref(x), igloo_ro_unref(x), NULL a harmless error   ref: src/transform.c:57,88-90
 *
 * libigloo is not installed in this repository's environment: this file has been read against the
 * reference's usage, not compiled.  What it relies on beyond the names: igloo_ro_new_raw() returns zero-filled
 * memory with one reference, and the free callback runs once, before the memory is released -- both relied on
 * by the reference's own stages at the lines cited. */
#ifndef COOLMIC_RO_IGLOO_H
#define COOLMIC_RO_IGLOO_H

#ifndef COOLMIC_DSP_USE_LIBIGLOO
#error "ro_igloo.h is the libigloo back end: build with -DCOOLMIC_DSP_USE_LIBIGLOO (make dropin IGLOO=1)"
#endif

/* the reference's private header: <coolmic-dsp/types.h>, igloo_RO_APPTYPES, <igloo/types.h>, <igloo/ro.h> in
 * the order libigloo wants them (ref: src/types_private.h:31-34); found through -I$(REF)/src */
#include "types_private.h"

#define COOLMIC_RO_TYPE(type, destroy) \
    static void type##__free(igloo_ro_t self) \
    { \
        destroy(igloo_RO_TO_TYPE(self, type)); \
    } \
    igloo_RO_PUBLIC_TYPE(type, igloo_RO_TYPEDECL_FREE(type##__free))

#define COOLMIC_RO_NEW(type, name, associated) igloo_ro_new_raw(type, (name), (associated))

/* (always with a pointer to one of the registered types in hand, as libigloo's igloo_ro_t wants it) */
#define coolmic_ro_ref(x)    igloo_ro_ref(x)
#define coolmic_ro_unref(x)  igloo_ro_unref(x)

#endif
