/* util.c -- VU presentation helpers (contract: <coolmic-dsp/util.h>; ref: src/util.c). */
#include "host_internal.h"
#include <coolmic-dsp/util.h>

#include <math.h>
#include <string.h>

/* 0..1 -> 0..255 (truncating), saturating at both ends */
static coolmic_argb_t unit_to_byte(double x)
{
    coolmic_argb_t b;

    if (x >= 1.)
        x = 1.;
    else if (x <= 0.)
        x = 0.;
    b = (coolmic_argb_t)(x * 255.);
    return b > 255 ? 255 : b;
}

coolmic_argb_t coolmic_util_ahsv2argb(double alpha, double hue, double saturation, double value)
{
    /* sextant of the colour wheel; like the reference, the blend factor is the hue minus
     * the sextant NUMBER (not the fractional sextant), ref: src/util.c:60-61 */
    const int sextant = (int)(double)(hue / (M_PI / 3.));
    const double f = hue - (double)sextant;
    const double lo = value * (1. - saturation);
    const double falling = value * (1. - saturation * f);
    const double rising = value * (1. - saturation * (1. - f));
    double r = 0., g = 0., b = 0.;

    switch (sextant) {
    case 0: case 6: r = value;   g = rising;  b = lo;      break;
    case 1:         r = falling; g = value;   b = lo;      break;
    case 2:         r = lo;      g = value;   b = rising;  break;
    case 3:         r = lo;      g = falling; b = value;   break;
    case 4:         r = rising;  g = lo;      b = value;   break;
    case 5:         r = value;   g = lo;      b = falling; break;
    default:        break;                       /* outside the wheel: black */
    }
    return (unit_to_byte(alpha) << 24) + (unit_to_byte(r) << 16) + (unit_to_byte(g) << 8) +
           unit_to_byte(b);
}

double coolmic_util_power2hue(double power, const char *profile)
{
    if (strcmp(profile, COOLMIC_UTIL_PROFILE_DEFAULT) != 0)
        return 0.;
    if (power < -20.)
        return M_PI * 2. / 3.;
    if (power >= 0)
        return 0;
    return pow(sin(M_PI * power / 40.), 2.) * M_PI * 2. / 3.;
}

double coolmic_util_peak2hue(int16_t peak, const char *profile)
{
    if (strcmp(profile, COOLMIC_UTIL_PROFILE_DEFAULT) != 0)
        return 0.;
    if (peak == -32768 || peak == 32767)
        return 0.;
    if (peak < -30000 || peak > 30000)
        return 0.43;
    if (peak < -28000 || peak > 28000)
        return 1.;
    return M_PI * 2. / 3.;
}

void coolmic_util_vu_argb(const coolmic_vumeter_result_t *results, size_t n, const char *profile,
                          coolmic_argb_t *power_argb, coolmic_argb_t *peak_argb)
{
    size_t i;

    for (i = 0; i < n; i++) {
        if (power_argb != NULL)
            power_argb[i] = coolmic_util_ahsv2argb(1., coolmic_util_power2hue(results[i].global_power, profile), 1., 1.);
        if (peak_argb != NULL)
            peak_argb[i] = coolmic_util_ahsv2argb(1., coolmic_util_peak2hue(results[i].global_peak, profile), 1., 1.);
    }
}
