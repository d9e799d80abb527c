/*
 * group_server.c -- a many-stream host on the operator API: N "sine" sound devices, one
 * coolmic_group_t, and the loop of a capture server -- pump a block, read every stream's
 * transformed PCM through its coolmic_iohandle_t, take a VU result per stream now and then.
 * While the readers drain block k the GPU already works on block k+1 (group.h).
 *
 *   cc -I include examples/group_server.c -L libcoolmic-dsp_amd/lib -lcoolmic-dsp-hip \
 *      -Wl,-rpath,$PWD/libcoolmic-dsp_amd/lib -o group_server && ./group_server [streams] [block] [rounds] [pull threads]
 *
 * Prints the per-block times of the loop and, for stream 0 and the last stream, what the
 * golden vector G1 of SURVEY 8(c) says a 1 kHz sine at gain 1000/1000 must give.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <coolmic-dsp/coolmic-dsp.h>
#include <coolmic-dsp/snddev.h>
#include <coolmic-dsp/group.h>

static double now_ms(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

int main(int argc, char **argv)
{
    const unsigned streams = argc > 1 ? (unsigned)atoi(argv[1]) : 1024;
    const size_t block = argc > 2 ? (size_t)atoi(argv[2]) : 4096;
    const unsigned rounds = argc > 3 ? (unsigned)atoi(argv[3]) : 16;
    const unsigned pull_threads = argc > 4 ? (unsigned)atoi(argv[4]) : 1;   /* the pump's reads of the sources */
    static const uint16_t unity[1] = {1000};
    const size_t nbytes = block * 2;                      /* mono int16 */
    coolmic_group_t *grp = coolmic_group_new(NULL, igloo_RO_NULL, 48000, 1, streams, block, 2);
    coolmic_iohandle_t **out = calloc(streams, sizeof(*out));
    int16_t *buf = malloc(nbytes);
    double t_pump = 0, t_read = 0, t0, t1, t2;
    unsigned long long sum = 0;
    unsigned s, r;

    if (!grp) {
        fprintf(stderr, "no group (no GPU?)\n");
        return 1;
    }
    if (coolmic_group_set_pull_threads(grp, pull_threads) != COOLMIC_ERROR_NONE)
        return 1;
    for (s = 0; s < streams; s++) {
        coolmic_snddev_t *dev = coolmic_snddev_new(NULL, igloo_RO_NULL, "sine", NULL, 48000, 1,
                                                   COOLMIC_DSP_SNDDEV_RX, -1);
        coolmic_iohandle_t *h = coolmic_snddev_get_iohandle(dev);
        const int slot = coolmic_group_add_stream(grp, h);
        igloo_ro_unref(h);
        igloo_ro_unref(dev);
        if (slot != (int)s || coolmic_group_set_master_gain(grp, s, 1, 1000, unity) != COOLMIC_ERROR_NONE)
            return 1;
        out[s] = coolmic_group_get_iohandle(grp, s);
    }
    for (r = 0; r < rounds + 2; r++) {                    /* two warm-up rounds */
        t0 = now_ms();
        if (coolmic_group_pump(grp) < 0)
            return 1;
        t1 = now_ms();
        if (r > 0) {                                      /* block r-1 is in the queues by now */
            for (s = 0; s < streams; s++) {
                if (coolmic_iohandle_read(out[s], buf, nbytes) != (ssize_t)nbytes)
                    return 2;
                sum += (unsigned short)buf[block / 2];
            }
        }
        t2 = now_ms();
        if (r >= 2) {
            t_pump += t1 - t0;
            t_read += t2 - t1;
        }
    }
    for (s = 0; s < streams; s++)                          /* the last block */
        if (coolmic_iohandle_read(out[s], buf, nbytes) != (ssize_t)nbytes)
            return 2;
    printf("streams %u block %zu pull threads %u: pump %.3f ms, readers %.3f ms per block -> %.0f Msamples/s (checksum %llu)\n",
           streams, block, pull_threads, t_pump / rounds, t_read / rounds,
           (double)streams * block / ((t_pump + t_read) / rounds * 1e-3) / 1e6, sum);
    for (s = 0; s < streams; s += streams - 1 ? streams - 1 : 1) {
        coolmic_vumeter_result_t res;
        if (coolmic_group_vumeter_result(grp, s, &res) != COOLMIC_ERROR_NONE)
            return 3;
        printf("stream %u: frames %zu peak %d power %.17g\n", s, res.frames, (int)res.global_peak,
               res.global_power);
    }
    for (s = 0; s < streams; s++)
        igloo_ro_unref(out[s]);
    igloo_ro_unref(grp);
    free(out);
    free(buf);
    return 0;
}
