/* hostdev.c -- which GPU the per-object stages use. */
#include "host_internal.h"

#include <stdlib.h>
#include <coolmic_hip.h>

int coolmic_hip_default_device(void)
{
    const char *e = getenv("COOLMIC_HIP_DEVICE");
    return e ? atoi(e) : 0;
}

/* a stage's own choice (stored as device + 1 in zero-filled objects: 0 = none made) or the default */
int coolmic_hip_stage_device(int chosen_plus1)
{
    return chosen_plus1 > 0 ? chosen_plus1 - 1 : coolmic_hip_default_device();
}

/* COOLMIC_ERROR_NONE for a device this process sees */
int coolmic_hip_check_device(int device)
{
    return device >= 0 && device < cmhip_device_count() ? COOLMIC_ERROR_NONE : COOLMIC_ERROR_INVAL;
}
