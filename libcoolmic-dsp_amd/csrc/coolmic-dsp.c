/* coolmic-dsp.c -- error texts and the feature string
 * (contract: <coolmic-dsp/coolmic-dsp.h>; ref: src/coolmic-dsp.c:30-112). */
#include "host_internal.h"

#include <string.h>

const char *coolmic_error2string(const int error)
{
    switch (error) {
    case COOLMIC_ERROR_NONE:        return "No error";
    case COOLMIC_ERROR_GENERIC:     return "Generic, unknown error";
    case COOLMIC_ERROR_NOSYS:       return "Function not implemented";
    case COOLMIC_ERROR_FAULT:       return "Bad address";
    case COOLMIC_ERROR_INVAL:       return "Invalid argument";
    case COOLMIC_ERROR_NOMEM:       return "Not enough space";
    case COOLMIC_ERROR_BUSY:        return "Device or resource busy";
    case COOLMIC_ERROR_PERM:        return "Operation not permitted";
    case COOLMIC_ERROR_CONNREFUSED: return "Connection refused";
    case COOLMIC_ERROR_CONNECTED:   return "Connected.";
    case COOLMIC_ERROR_UNCONNECTED: return "Unconnected.";
    case COOLMIC_ERROR_NOTLS:       return "TLS requested but not supported by peer";
    case COOLMIC_ERROR_TLSBADCERT:
        return "TLS connection can not be established because of bad certificate";
    case COOLMIC_ERROR_BADRQC:      return "Invalid request code";
    case COOLMIC_ERROR_RETRY:       return "Retry last action";
    }
    return "(unknown)";
}

/* The stand-alone library names the drivers it brings itself.  Inside the reference's build (`make dropin`,
 * INTEGRATION.md 3) the drivers and encoders are the host's own: its Makefile passes their tokens in
 * COOLMIC_HOST_FEATURES (what ref: src/coolmic-dsp.c:64-83 would have put together from its HAVE_* flags),
 * and this unit adds the one thing it knows: which path computes transform and VU. */
#ifndef COOLMIC_HOST_FEATURES
#define COOLMIC_HOST_FEATURES COOLMIC_FEATURE_DRIVER_NULL " " COOLMIC_FEATURE_DRIVER_SINE " " COOLMIC_FEATURE_DRIVER_STDIO
#endif

const char *coolmic_features(void)
{
    return "features " COOLMIC_HOST_FEATURES " " COOLMIC_FEATURE_ACCEL_HIP;
}

/* 1 when the list holds `feature` at the start of a word and a blank or the end behind it.  The text is compared as
 * it stands, so a query of several words matches where those words follow each other, and after a match that is not
 * followed by a blank the search goes on behind the matched text -- both as the reference does it
 * (ref: src/coolmic-dsp.c:85-112; held against a build of that file in tests/test_ref_core.py). */
int coolmic_feature_check(const char *feature)
{
    const char *at = coolmic_features();
    size_t want;

    if (feature == NULL)
        return COOLMIC_ERROR_FAULT;
    if (*feature == 0)
        return COOLMIC_ERROR_INVAL;
    want = strlen(feature);
    for (;;) {
        if (strncmp(at, feature, want) == 0) {
            if (at[want] == 0 || at[want] == ' ')
                return 1;
            at += want;
        }
        at = strchr(at, ' ');
        if (at == NULL)
            return 0;
        at++;
    }
}
