/*
 * node_vu.c -- BASELINE config 5 from plain C: the streams of a node sharded round-robin
 * over its GPUs (stream s lives on GPU s % N), one host thread per GPU, a step loop of
 * transform -> VU per block, and the node-global VU peak / RMS reduced over the GPUs by RCCL
 * (cmhip_node_*: one ncclAllReduce(int64, sum) + one ncclAllReduce(uint64, max) per `NB`
 * blocks).  No Python, no torch.
 *
 *   cc -I include examples/node_vu.c -L libcoolmic-dsp_amd/lib -lcoolmic-dsp-hip -lpthread \
 *      -Wl,-rpath,$PWD/libcoolmic-dsp_amd/lib -o node_vu
 *   ./node_vu [streams [frames [blocks [gpus]]]]      (gpus: default all of cmhip_device_count())
 *
 * Every rank ends with the same combined records; rank 0 prints the node-global result of
 * each block.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <coolmic-dsp/coolmic-dsp.h>
#include <coolmic_hip.h>

enum { NB = 4 };                                  /* blocks whose records travel in one exchange */

struct rank_arg {
    int rank, nranks;
    unsigned streams_total, frames, blocks;
    const unsigned char *id;
    int64_t *combined;                            /* rank 0: [blocks][CMHIP_NODE_WORDS] */
    int rc;
};

static void *rank_main(void *p)
{
    struct rank_arg *a = p;
    const uint16_t gain[1] = {900};
    cmhip_batch_desc_t d = {0};
    cmhip_batch_t *b = NULL;
    cmhip_node_t *n = NULL;
    int64_t words[NB * CMHIP_NODE_WORDS];
    unsigned blk;

    a->rc = 1;
    /* streams rank, rank + N, rank + 2N, ... of the node */
    d.device = a->rank;
    d.streams = a->streams_total / a->nranks + ((unsigned)a->rank < a->streams_total % a->nranks ? 1u : 0u);
    d.channels = 1; d.rate = 48000; d.max_frames = a->frames;
    d.flags = CMHIP_OUT_PCM | CMHIP_VU;
    b = cmhip_batch_new(&d);
    n = cmhip_node_new(a->rank, a->nranks, a->rank, a->id, NB);      /* collective: every rank is here */
    if (!b || !n) {
        fprintf(stderr, "rank %d: %s\n", a->rank, cmhip_last_error());
        goto out;
    }
    cmhip_batch_set_gain(b, -1, 1, 1000, gain);
    for (blk = 0; blk < a->blocks; blk++) {
        const unsigned set = (blk / NB) & 1u, slot = blk % NB;
        /* this block's PCM: per-stream LCG noise, continued from block to block */
        if (cmhip_batch_generate(b, CMHIP_GEN_NOISE, 12345, a->frames, (uint64_t)a->rank, (uint64_t)a->nranks,
                                 (uint64_t)blk * a->frames) != COOLMIC_ERROR_NONE ||
            cmhip_batch_run(b, a->frames, NULL) != COOLMIC_ERROR_NONE ||
            cmhip_node_partial(n, b, set, slot, (uint64_t)a->rank, (uint64_t)a->nranks) != COOLMIC_ERROR_NONE) {
            fprintf(stderr, "rank %d block %u: %s\n", a->rank, blk, cmhip_last_error());
            goto out;
        }
        if (cmhip_batch_vu_reset(b, -1) != COOLMIC_ERROR_NONE)       /* one window per block */
            goto out;
        if (slot == NB - 1 || blk + 1 == a->blocks) {
            const unsigned count = slot + 1;
            if (cmhip_node_allreduce(n, set, count, b) != COOLMIC_ERROR_NONE ||
                cmhip_node_fetch(n, set, count, words) != COOLMIC_ERROR_NONE) {
                fprintf(stderr, "rank %d exchange: %s\n", a->rank, cmhip_last_error());
                goto out;
            }
            if (a->rank == 0)
                memcpy(a->combined + (size_t)(blk - slot) * CMHIP_NODE_WORDS, words,
                       sizeof(int64_t) * CMHIP_NODE_WORDS * count);
        }
    }
    a->rc = 0;
out:
    cmhip_node_free(n);
    cmhip_batch_free(b);
    return NULL;
}

int main(int argc, char **argv)
{
    const unsigned streams = argc > 1 ? (unsigned)atoi(argv[1]) : 1024;
    const unsigned frames = argc > 2 ? (unsigned)atoi(argv[2]) : 4096;
    const unsigned blocks = argc > 3 ? (unsigned)atoi(argv[3]) : 6;
    int gpus = argc > 4 ? atoi(argv[4]) : cmhip_device_count();
    unsigned char id[CMHIP_NODE_ID_BYTES];
    struct rank_arg *args;
    pthread_t *th;
    int64_t *combined;
    int r, bad = 0;
    unsigned blk;

    if (gpus < 1 || gpus > cmhip_device_count() || !streams || !frames || !blocks) {
        fprintf(stderr, "node_vu: %d GPUs asked for, %d present\n", gpus, cmhip_device_count());
        return 1;
    }
    if (cmhip_node_unique_id(id) != COOLMIC_ERROR_NONE) {
        fprintf(stderr, "node_vu: %s\n", cmhip_last_error());
        return 1;
    }
    args = calloc((size_t)gpus, sizeof(*args));
    th = calloc((size_t)gpus, sizeof(*th));
    combined = calloc((size_t)blocks * CMHIP_NODE_WORDS, sizeof(int64_t));
    for (r = 0; r < gpus; r++) {
        args[r].rank = r; args[r].nranks = gpus;
        args[r].streams_total = streams; args[r].frames = frames; args[r].blocks = blocks;
        args[r].id = id; args[r].combined = combined;
        pthread_create(&th[r], NULL, rank_main, &args[r]);
    }
    for (r = 0; r < gpus; r++) {
        pthread_join(th[r], NULL);
        bad |= args[r].rc;
    }
    if (bad)
        return 1;
    printf("gpus %d streams %u frames %u blocks %u\n", gpus, streams, frames, blocks);
    for (blk = 0; blk < blocks; blk++) {
        coolmic_vumeter_result_t res;
        if (cmhip_node_finish(combined + (size_t)blk * CMHIP_NODE_WORDS, 1, 48000, &res) != COOLMIC_ERROR_NONE) {
            fprintf(stderr, "node_vu: block %u has no frames\n", blk);
            return 1;
        }
        printf("block %u: frames %zu peak %d power %.17g\n", blk, res.frames, (int)res.global_peak,
               res.global_power);
    }
    free(args); free(th); free(combined);
    return 0;
}
