/* ro.c -- refcounted objects behind <coolmic-dsp/ro-compat.h>.
 *
 * Provides what the reference takes from libigloo for the objects on this path
 * (ref: src/iohandle.c:50-52,62; src/transform.c:54-62,72; src/vumeter.c:59-67,76):
 * zero-filled allocation with one reference, ref/unref, a per-type destructor that
 * runs when the last reference goes, NULL tolerated as a reported error. */
#include "host_internal.h"

#include <stdlib.h>
#include <string.h>

void *coolmic_ro_new_raw(const coolmic_ro_type_t *type, const char *name, void *associated)
{
    coolmic_ro_base_t *base;

    if (type == NULL || type->size < sizeof(coolmic_ro_base_t))
        return NULL;
    base = calloc(1, type->size);
    if (base == NULL)
        return NULL;
    base->type = type;
    base->refc = 1;
    if (name != NULL) {
        base->name = strdup(name);
        if (base->name == NULL) {
            free(base);
            return NULL;
        }
    }
    if (associated != NULL && coolmic_ro_ref(associated) == COOLMIC_ERROR_NONE)
        base->associated = associated;
    return base;
}

int coolmic_ro_ref(void *self)
{
    coolmic_ro_base_t *base = self;

    if (base == NULL)
        return COOLMIC_ERROR_FAULT;
    __atomic_add_fetch(&base->refc, 1, __ATOMIC_RELAXED);
    return COOLMIC_ERROR_NONE;
}

int coolmic_ro_unref(void *self)
{
    coolmic_ro_base_t *base = self;

    if (base == NULL)
        return COOLMIC_ERROR_FAULT;
    if (__atomic_sub_fetch(&base->refc, 1, __ATOMIC_ACQ_REL) != 0)
        return COOLMIC_ERROR_NONE;
    if (base->type->free_cb != NULL)
        base->type->free_cb(base);
    if (base->associated != NULL)
        coolmic_ro_unref(base->associated);
    free(base->name);
    free(base);
    return COOLMIC_ERROR_NONE;
}

unsigned int coolmic_ro_refcount(void *self)
{
    coolmic_ro_base_t *base = self;
    return base ? __atomic_load_n(&base->refc, __ATOMIC_RELAXED) : 0;
}
