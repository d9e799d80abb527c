/*
 * util.h -- presentation helpers for VU results: level -> hue, AHSV -> ARGB.
 * Same API and numeric behaviour as the reference (ref: include/coolmic-dsp/util.h:33-46,
 * src/util.c:59-138).  Per VU event, not per sample: plain host code in double.
 * coolmic_util_vu_argb() is an addition for hosts with thousands of meters.
 */
#ifndef __COOLMIC_DSP_UTIL_H__
#define __COOLMIC_DSP_UTIL_H__

#include <stddef.h>
#include <stdint.h>
#include "vumeter.h"

#ifdef __cplusplus
extern "C" {
#endif

#define COOLMIC_UTIL_PROFILE_DEFAULT      "default"

typedef uint32_t coolmic_argb_t;

/* alpha, saturation, value in 0..1, hue in radians (0 red, 2pi/3 green); each output byte is
 * the clamped component times 255, truncated */
coolmic_argb_t coolmic_util_ahsv2argb(double alpha, double hue, double saturation, double value);
/* power in dB: green below -20 dB, red from 0 dB, sin^2 ramp in between; other profiles: red */
double         coolmic_util_power2hue(double power, const char *profile);
/* peak: red at full scale, orange / yellow steps above 30000 / 28000, green otherwise */
double         coolmic_util_peak2hue(int16_t peak, const char *profile);

/* colours of n results at once (global power and global peak of each): two ARGB words per
 * result, fully opaque and saturated -- what a meter bank draws per VU event */
void           coolmic_util_vu_argb(const coolmic_vumeter_result_t *results, size_t n,
                                    const char *profile, coolmic_argb_t *power_argb,
                                    coolmic_argb_t *peak_argb);

#ifdef __cplusplus
}
#endif
#endif
