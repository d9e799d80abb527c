/*
 * config1_chain.c -- BASELINE config 1 written exactly like a client of the reference:
 * snddev "sine" -> transform(gain 1000/1000) -> vumeter, 1 channel int16 48 kHz, wired
 * through coolmic_iohandle_t the way ref: src/simple.c:198-229 wires its pipeline.
 *
 *   cc -I include examples/config1_chain.c -L libcoolmic-dsp_amd/lib -lcoolmic-dsp-hip \
 *      -Wl,-rpath,$PWD/libcoolmic-dsp_amd/lib -o config1_chain && ./config1_chain
 *
 * Prints the three VU windows of SURVEY 8(c) G1..G3 (gain 1.0, 0.5, 2.0).  Needs an
 * MI355X: the arithmetic runs in the HIP kernels, there is no CPU path.
 */
#include <stdio.h>
#include <coolmic-dsp/coolmic-dsp.h>
#include <coolmic-dsp/logging.h>
#include <coolmic-dsp/snddev.h>
#include <coolmic-dsp/transform.h>
#include <coolmic-dsp/vumeter.h>

static int log_cb(coolmic_logging_level_t level, const char *msg)
{
    if (level <= COOLMIC_LOGGING_LEVEL_WARNING)
        fprintf(stderr, "%s\n", msg);
    return 0;
}

int main(void)
{
    static const struct { uint16_t gain; int reads; } windows[] = {{1000, 94}, {500, 1}, {2000, 1}};
    coolmic_snddev_t *dev;
    coolmic_transform_t *tr;
    coolmic_vumeter_t *vu;
    coolmic_iohandle_t *h;
    coolmic_vumeter_result_t r;
    size_t w;

    coolmic_logging_set_cb_simple(log_cb);
    dev = coolmic_snddev_new("source", igloo_RO_NULL, COOLMIC_DSP_SNDDEV_DRIVER_SINE, NULL, 48000, 1,
                             COOLMIC_DSP_SNDDEV_RX, -1);
    tr = coolmic_transform_new("transform", igloo_RO_NULL, 48000, 1);
    vu = coolmic_vumeter_new("vumeter", igloo_RO_NULL, 48000, 1);
    if (!dev || !tr || !vu)
        return 2;

    /* attach, then drop our reference: the consumer now owns the handle */
    h = coolmic_snddev_get_iohandle(dev);
    coolmic_transform_attach_iohandle(tr, h);
    igloo_ro_unref(h);
    h = coolmic_transform_get_iohandle(tr);
    coolmic_vumeter_attach_iohandle(vu, h);
    igloo_ro_unref(h);

    for (w = 0; w < sizeof(windows) / sizeof(windows[0]); w++) {
        int i, rc;
        coolmic_transform_set_master_gain(tr, 1, 1000, &windows[w].gain);
        for (i = 0; i < windows[w].reads; i++) {
            if (coolmic_vumeter_read(vu, -1) != 1024) {
                fprintf(stderr, "read failed (no GPU?)\n");
                return 1;
            }
        }
        rc = coolmic_vumeter_result(vu, &r);
        if (rc != COOLMIC_ERROR_NONE) {
            fprintf(stderr, "result: %s\n", coolmic_error2string(rc));
            return 1;
        }
        printf("frames=%zu peak=%d power=%.17g\n", r.frames, (int)r.global_peak, r.global_power);
    }

    igloo_ro_unref(vu);
    igloo_ro_unref(tr);
    igloo_ro_unref(dev);
    return 0;
}
