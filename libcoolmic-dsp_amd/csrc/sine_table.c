/* sine_table.c -- the 1 kHz test tone of the "sine" source.
 *
 * The reference stores one literal period per supported rate
 * (ref: src/snddev_sine.c:36-99).  All of them are trunc(A*sin(2*pi*k/N)) with
 * N = rate/1000 for any A in [32766.938, 32767), so the tables are computed
 * instead of stored; tests/test_oracle_golden.py checks the formula against the
 * reference text when it is available. */
#include "host_internal.h"

#include <math.h>

#define COOLMIC_SINE_AMPLITUDE 32766.97

int coolmic_sine_period(uint_least32_t rate, int16_t *table, size_t *samples)
{
    size_t n, k;

    switch (rate) {
    case 8000: case 16000: case 24000: case 32000:
    case 44000: case 44100: case 48000: case 96000:
        n = rate / 1000;
        break;
    default:
        return COOLMIC_ERROR_NOSYS;
    }
    for (k = 0; k < n; k++)
        table[k] = (int16_t)trunc(COOLMIC_SINE_AMPLITUDE * sin(2. * M_PI * (double)k / (double)n));
    *samples = n;
    return COOLMIC_ERROR_NONE;
}
