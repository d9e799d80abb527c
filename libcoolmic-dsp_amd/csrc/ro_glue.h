/* ro_glue.h -- how the stages of this library become refcounted objects (private to csrc/).
 *
 * A stage declares its struct with `igloo_ro_base_t __base;` first, as the reference's structs do
 * (ref: src/transform.c:38, src/vumeter.c:37, src/iohandle.c:33, src/tee.c:47), registers its type once with
 *
 *     COOLMIC_RO_TYPE(coolmic_transform_t, transform_destroy);
 *
 * and creates objects with `COOLMIC_RO_NEW(coolmic_transform_t, name, associated)`: zero-filled, one
 * reference, the destroy callback runs when the last reference goes (ref: src/transform.c:54-62,72).
 *
 * Two back ends.  The stand-alone library (the default build) backs them with csrc/ro.c through
 * <coolmic-dsp/ro-compat.h>.  Inside the reference's own build (`make dropin IGLOO=1`) the
 * objects must BE libigloo objects -- simple.c, enc.c, shout.c hand them to igloo_ro_ref / igloo_ro_unref
 * (ref: src/simple.c:164-170,212-229) -- and csrc/ro_igloo.h maps the same three macros and
 * coolmic_ro_ref / coolmic_ro_unref onto libigloo's own. */
#ifndef COOLMIC_RO_GLUE_H
#define COOLMIC_RO_GLUE_H

#ifdef COOLMIC_DSP_USE_LIBIGLOO
#include "ro_igloo.h"
#else
#include <coolmic-dsp/ro-compat.h>

#define COOLMIC_RO_TYPE(type, destroy) \
    static const coolmic_ro_type_t type##__rotype = {#type, sizeof(type), destroy}
#define COOLMIC_RO_NEW(type, name, associated) \
    ((type *)coolmic_ro_new_raw(&type##__rotype, (name), (void *)(associated)))
#endif

#endif
