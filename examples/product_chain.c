/*
 * product_chain.c -- the reference's LIVE pipeline up to the encoder, written like ref: src/simple.c:183-236:
 *
 *     snddev "sine" -> transform -> tee -+-> reader 0: the encoder branch (1024-byte pulls, ref: src/enc_vorbis.c:91)
 *                                        +-> reader 1: vumeter, a result every 20 reads (ref: src/simple.c:370,486-491)
 *
 * and, for comparison, BASELINE config 1 (sine -> transform -> vumeter, no tee).  Prints what one 1024-byte pull
 * costs in each wiring and the last VU window.  In both the meter shares the transform's launch -- directly, or
 * through the tee by window records -- so a pull is one launch and one wait.
 *
 *   cc -I include examples/product_chain.c -L libcoolmic-dsp_amd/lib -lcoolmic-dsp-hip \
 *      -Wl,-rpath,$PWD/libcoolmic-dsp_amd/lib -o product_chain && ./product_chain [pulls]
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <coolmic-dsp/coolmic-dsp.h>
#include <coolmic-dsp/snddev.h>
#include <coolmic-dsp/tee.h>
#include <coolmic-dsp/transform.h>
#include <coolmic-dsp/vumeter.h>

static double now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* wiring: 0 sine -> transform -> vumeter, 1 through a tee with an encoder-branch reader beside the meter,
 * 2 sine -> transform alone, pulled by this program (no meter anywhere: what the VU window adds to a pull) */
static int run(int with_tee, int with_gain, long pulls)
{
    static const uint16_t gain = 900;
    coolmic_snddev_t *dev = coolmic_snddev_new("source", igloo_RO_NULL, COOLMIC_DSP_SNDDEV_DRIVER_SINE, NULL, 48000, 1,
                                               COOLMIC_DSP_SNDDEV_RX, -1);
    coolmic_transform_t *tr = coolmic_transform_new("transform", igloo_RO_NULL, 48000, 1);
    const int no_meter = with_tee == 2;
    coolmic_vumeter_t *vu = coolmic_vumeter_new("vumeter", igloo_RO_NULL, 48000, 1);
    coolmic_tee_t *tee = with_tee == 1 ? coolmic_tee_new("tee", igloo_RO_NULL, 2) : NULL;
    coolmic_iohandle_t *h, *enc = NULL;
    coolmic_vumeter_result_t r;
    unsigned char pcm[1024];
    double t0 = 0., dt;
    long i, results = 0;
    const long warm = 200;

    if (!dev || !tr || !vu || (with_tee == 1 && !tee))
        return 2;
    /* attach, then drop our reference: the consumer now owns the handle (ref: src/simple.c:212-229) */
    h = coolmic_snddev_get_iohandle(dev);
    coolmic_transform_attach_iohandle(tr, h);
    igloo_ro_unref(h);
    h = coolmic_transform_get_iohandle(tr);
    if (no_meter) {
        enc = h;                       /* this program is the only reader */
        h = NULL;
    } else if (with_tee) {
        coolmic_tee_attach_iohandle(tee, h);
        igloo_ro_unref(h);
        enc = coolmic_tee_get_iohandle(tee, 0);
        h = coolmic_tee_get_iohandle(tee, 1);
    }
    if (!no_meter) {
        coolmic_vumeter_attach_iohandle(vu, h);
        igloo_ro_unref(h);
    }
    if (with_gain)
        coolmic_transform_set_master_gain(tr, 1, 1000, &gain);

    for (i = 0; i < warm + pulls; i++) {
        if (i == warm)
            t0 = now();
        if (enc && coolmic_iohandle_read(enc, pcm, sizeof(pcm)) != (ssize_t)sizeof(pcm)) {
            fprintf(stderr, "encoder branch: short read (no GPU?)\n");
            return 1;
        }
        if (no_meter)
            continue;
        if (coolmic_vumeter_read(vu, -1) != 1024) {
            fprintf(stderr, "vumeter: read failed (no GPU?)\n");
            return 1;
        }
        if (i % 20 == 19) {
            if (coolmic_vumeter_result(vu, &r) != COOLMIC_ERROR_NONE)
                return 1;
            results++;
        }
    }
    dt = now() - t0;
    if (no_meter) {
        printf("no meter gain %s: %.2f us per 1024-byte pull, %.2f Msamples/s\n", with_gain ? "on " : "off",
               dt / (double)pulls * 1e6, 512. * (double)pulls / dt / 1e6);
        results = 1;
    } else {
        printf("%s gain %s: %.2f us per 1024-byte pull, %.2f Msamples/s; last window: frames %zu peak %d power %.17g\n",
               with_tee ? "tee   " : "direct", with_gain ? "on " : "off", dt / (double)pulls * 1e6,
               512. * (double)pulls / dt / 1e6, r.frames, (int)r.global_peak, r.global_power);
    }
    igloo_ro_unref(enc);
    igloo_ro_unref(vu);
    igloo_ro_unref(tee);
    igloo_ro_unref(tr);
    igloo_ro_unref(dev);
    return results ? 0 : 1;
}

int main(int argc, char **argv)
{
    const long pulls = argc > 1 ? atol(argv[1]) : 4000;
    int gain, tee, rc = 0;

    for (gain = 0; gain < 2 && rc == 0; gain++)
        for (tee = 0; tee < 2 && rc == 0; tee++)
            rc = run(tee, gain, pulls > 0 ? pulls : 4000);
    if (rc == 0)                       /* (gain off and no meter would be the reference's early-out: no launch at all) */
        rc = run(2, 1, pulls > 0 ? pulls : 4000);
    return rc;
}
