/* hostdev.c -- which GPU the per-object stages use. */
#include "host_internal.h"

#include <stdlib.h>

int coolmic_hip_default_device(void)
{
    const char *e = getenv("COOLMIC_HIP_DEVICE");
    return e ? atoi(e) : 0;
}
