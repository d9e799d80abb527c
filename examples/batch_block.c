/*
 * batch_block.c -- the batch C ABI (include/coolmic_hip.h) from plain C: 256 stereo streams,
 * one block, transformed PCM back on the host and one VU result per stream.
 *
 *   cc -I include examples/batch_block.c -L libcoolmic-dsp_amd/lib -lcoolmic-dsp-hip \
 *      -Wl,-rpath,$PWD/libcoolmic-dsp_amd/lib -o batch_block && ./batch_block
 */
#include <stdio.h>
#include <stdlib.h>
#include <coolmic-dsp/coolmic-dsp.h>
#include <coolmic_hip.h>

enum { STREAMS = 256, FRAMES = 4096 };

int main(void)
{
    cmhip_batch_desc_t d = {0};
    static const uint16_t gain[2] = {750, 1250};
    static const uint8_t swap[2] = {1, 0};
    coolmic_vumeter_result_t *res = calloc(STREAMS, sizeof(*res));
    int *rcs = calloc(STREAMS, sizeof(*rcs));
    int16_t *pcm = malloc(sizeof(int16_t) * 2 * FRAMES);
    cmhip_batch_t *b;
    unsigned s;

    d.device = 0; d.streams = STREAMS; d.channels = 2; d.rate = 48000; d.max_frames = FRAMES;
    d.flags = CMHIP_OUT_PCM | CMHIP_VU;
    b = cmhip_batch_new(&d);
    if (!b) {
        fprintf(stderr, "cmhip_batch_new: %s\n", cmhip_last_error());
        return 1;
    }
    cmhip_batch_set_gain(b, -1, 2, 1000, gain);
    cmhip_batch_set_chmap(b, -1, swap);
    for (s = 0; s < STREAMS; s++) {                 /* any PCM; here the G4 noise per stream */
        uint32_t st = 12345u + s;
        int i;
        for (i = 0; i < 2 * FRAMES; i++) {
            st = st * 1664525u + 1013904223u;
            pcm[i] = (int16_t)(st >> 16);
        }
        if (cmhip_batch_upload(b, s, pcm, FRAMES) != COOLMIC_ERROR_NONE)
            return 1;
    }
    if (cmhip_batch_run(b, FRAMES, NULL) != COOLMIC_ERROR_NONE ||
        cmhip_batch_vu_results(b, res, rcs) != COOLMIC_ERROR_NONE ||
        cmhip_batch_download(b, 0, pcm, FRAMES) != COOLMIC_ERROR_NONE) {
        fprintf(stderr, "batch failed: %s\n", cmhip_last_error());
        return 1;
    }
    printf("stream 0: first frame %d %d, peak %d, power %.17g dB\n", pcm[0], pcm[1],
           (int)res[0].global_peak, res[0].global_power);
    printf("stream %d: peak %d, power %.17g dB\n", STREAMS - 1, (int)res[STREAMS - 1].global_peak,
           res[STREAMS - 1].global_power);
    cmhip_batch_free(b);
    free(res); free(rcs); free(pcm);
    return 0;
}
