/*
 * group_server.c -- a many-stream host on the operator API: N "sine" sound devices behind
 * coolmic_group_t, and the loop of a capture server -- pump a block, read every stream's
 * transformed PCM through its coolmic_iohandle_t, take a VU result per stream now and then.
 * While the readers drain block k the GPU already works on block k+1 (group.h).
 *
 * One process drives every GPU it is given: capture stream s lives in the group of GPU s % N
 * (coolmic_group_new_on), one pump-and-read thread per GPU, no traffic between the GPUs -- the
 * round-robin sharding of SURVEY 8(e) behind the library's own operator API.  With a fifth
 * argument the run ends with the node-global VU over all groups: every group's record of the
 * whole run (cmhip_batch_vu_node_record on coolmic_group_engine()), combined on the host
 * (cmhip_node_merge_host) and, where librccl is there, by the RCCL exchange as well
 * (cmhip_node_*), which must give the same words.
 *
 *   cc -I include examples/group_server.c -L libcoolmic-dsp_amd/lib -lcoolmic-dsp-hip -lpthread \
 *      -Wl,-rpath,$PWD/libcoolmic-dsp_amd/lib -o group_server
 *   ./group_server [streams] [block] [rounds] [pull threads] [gpus]     (gpus: 0 = all the process sees)
 *
 * Prints the per-block times of the loop and, for stream 0 and the last stream, what the
 * golden vector G1 of SURVEY 8(c) says a 1 kHz sine at gain 1000/1000 must give.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <coolmic-dsp/coolmic-dsp.h>
#include <coolmic-dsp/snddev.h>
#include <coolmic-dsp/group.h>
#include <coolmic_hip.h>

static double now_ms(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

struct gpu_arg {
    int gpu, ngpus;
    unsigned streams_total, rounds, pull_threads;
    size_t block;
    int want_node;
    const unsigned char *node_id;
    pthread_barrier_t *start;
    /* results */
    int rc;
    unsigned streams;                                     /* streams of this GPU */
    double t_pump, t_read;
    unsigned long long sum;
    int64_t record[CMHIP_NODE_WORDS];                     /* this group's share of the node-global VU */
    int64_t combined[CMHIP_NODE_WORDS];                   /* ... after the RCCL exchange */
    int rccl;                                             /* 1: combined[] is valid */
    coolmic_vumeter_result_t first, last;                 /* global stream 0 / the last stream, where they live here */
    int has_first, has_last;
};

static void *gpu_main(void *p)
{
    struct gpu_arg *a = p;
    static const uint16_t unity[1] = {1000};
    const size_t nbytes = a->block * 2;                   /* mono int16 */
    /* global streams gpu, gpu + N, gpu + 2N, ... -> local slots 0, 1, 2, ... */
    const unsigned streams = a->streams_total / a->ngpus + ((unsigned)a->gpu < a->streams_total % a->ngpus ? 1u : 0u);
    coolmic_group_t *grp = NULL;
    coolmic_iohandle_t **out = calloc(streams ? streams : 1, sizeof(*out));
    int16_t *buf = malloc(nbytes);
    cmhip_node_t *node = NULL;
    double t0, t1, t2;
    unsigned s, r;

    a->rc = 1;
    a->streams = streams;
    if (streams)
        grp = coolmic_group_new_on(a->gpu, NULL, igloo_RO_NULL, 48000, 1, streams, a->block, 2);
    if (a->want_node)                                     /* collective over the GPU threads: all of them are here */
        node = cmhip_node_new(a->gpu, a->ngpus, a->gpu, a->node_id, 1);
    if (!out || !buf || (streams && !grp)) {
        fprintf(stderr, "gpu %d: no group (no GPU?)\n", a->gpu);
        pthread_barrier_wait(a->start);
        goto done;
    }
    if (grp && coolmic_group_set_pull_threads(grp, a->pull_threads) != COOLMIC_ERROR_NONE)
        goto done_barrier;
    for (s = 0; s < streams; s++) {
        coolmic_snddev_t *dev = coolmic_snddev_new(NULL, igloo_RO_NULL, "sine", NULL, 48000, 1,
                                                   COOLMIC_DSP_SNDDEV_RX, -1);
        coolmic_iohandle_t *h = coolmic_snddev_get_iohandle(dev);
        const int slot = coolmic_group_add_stream(grp, h);
        igloo_ro_unref(h);
        igloo_ro_unref(dev);
        if (slot != (int)s || coolmic_group_set_master_gain(grp, s, 1, 1000, unity) != COOLMIC_ERROR_NONE)
            goto done_barrier;
        out[s] = coolmic_group_get_iohandle(grp, s);
    }
    pthread_barrier_wait(a->start);
    for (r = 0; grp && r < a->rounds + 2; r++) {          /* two warm-up rounds */
        t0 = now_ms();
        if (coolmic_group_pump(grp) < 0)
            goto done;
        t1 = now_ms();
        if (r > 0) {                                      /* block r-1 is in the queues by now */
            for (s = 0; s < streams; s++) {
                if (coolmic_iohandle_read(out[s], buf, nbytes) != (ssize_t)nbytes) {
                    a->rc = 2;
                    goto done;
                }
                a->sum += (unsigned short)buf[a->block / 2];
            }
        }
        t2 = now_ms();
        if (r >= 2) {
            a->t_pump += t1 - t0;
            a->t_read += t2 - t1;
        }
    }
    for (s = 0; s < streams; s++)                          /* the last block */
        if (coolmic_iohandle_read(out[s], buf, nbytes) != (ssize_t)nbytes) {
            a->rc = 2;
            goto done;
        }
    if (a->want_node) {
        /* the node-global VU of the whole run, before the per-stream results clear their windows: this
         * group's record to the host, and -- with a communicator -- the exchange over the GPUs */
        memset(a->record, 0, sizeof(a->record));
        if (grp && cmhip_batch_vu_node_record(coolmic_group_engine(grp), a->record, (uint64_t)a->gpu,
                                              (uint64_t)a->ngpus) != COOLMIC_ERROR_NONE) {
            fprintf(stderr, "gpu %d: node record: %s\n", a->gpu, cmhip_last_error());
            goto done;
        }
        if (node && grp &&
            cmhip_node_partial(node, coolmic_group_engine(grp), 0, 0, (uint64_t)a->gpu, (uint64_t)a->ngpus) == COOLMIC_ERROR_NONE &&
            cmhip_node_allreduce(node, 0, 1, coolmic_group_engine(grp)) == COOLMIC_ERROR_NONE &&
            cmhip_node_fetch(node, 0, 1, a->combined) == COOLMIC_ERROR_NONE)
            a->rccl = 1;
    }
    /* global stream 0 is slot 0 of GPU 0; the last global stream S-1 is slot (S-1) / N of GPU (S-1) % N */
    if (a->gpu == 0 && streams) {
        if (coolmic_group_vumeter_result(grp, 0, &a->first) != COOLMIC_ERROR_NONE) {
            a->rc = 3;
            goto done;
        }
        a->has_first = 1;
    }
    if ((a->streams_total - 1) % (unsigned)a->ngpus == (unsigned)a->gpu && a->streams_total > 1) {
        if (coolmic_group_vumeter_result(grp, (a->streams_total - 1) / (unsigned)a->ngpus, &a->last) != COOLMIC_ERROR_NONE) {
            a->rc = 3;
            goto done;
        }
        a->has_last = 1;
    }
    a->rc = 0;
    goto done;
done_barrier:
    pthread_barrier_wait(a->start);
done:
    for (s = 0; out && s < streams; s++)
        igloo_ro_unref(out[s]);
    cmhip_node_free(node);
    igloo_ro_unref(grp);
    free(out);
    free(buf);
    return NULL;
}

int main(int argc, char **argv)
{
    const unsigned streams = argc > 1 ? (unsigned)atoi(argv[1]) : 1024;
    const size_t block = argc > 2 ? (size_t)atoi(argv[2]) : 4096;
    const unsigned rounds = argc > 3 ? (unsigned)atoi(argv[3]) : 16;
    const unsigned pull_threads = argc > 4 ? (unsigned)atoi(argv[4]) : 1;   /* the pump's reads of the sources */
    const int want_node = argc > 5;
    int ngpus = argc > 5 ? atoi(argv[5]) : 1, g;
    unsigned char node_id[CMHIP_NODE_ID_BYTES] = {0};
    pthread_barrier_t start;
    struct gpu_arg *args;
    pthread_t *tid;
    double t_pump = 0, t_read = 0, t_slowest = 0;
    unsigned long long sum = 0;
    int rc = 0, have_node_id = 0;

    if (ngpus <= 0)
        ngpus = cmhip_device_count();
    if (ngpus < 1 || ngpus > cmhip_device_count() || streams == 0) {
        fprintf(stderr, "%d GPU(s) asked for, %d visible\n", ngpus, cmhip_device_count());
        return 1;
    }
    if (want_node)                                        /* (no librccl: the host merge alone) */
        have_node_id = cmhip_node_unique_id(node_id) == COOLMIC_ERROR_NONE;
    args = calloc((size_t)ngpus, sizeof(*args));
    tid = calloc((size_t)ngpus, sizeof(*tid));
    pthread_barrier_init(&start, NULL, (unsigned)ngpus);
    for (g = 0; g < ngpus; g++) {
        args[g].gpu = g; args[g].ngpus = ngpus;
        args[g].streams_total = streams; args[g].rounds = rounds; args[g].pull_threads = pull_threads;
        args[g].block = block; args[g].want_node = want_node && have_node_id; args[g].node_id = node_id;
        args[g].start = &start;
        pthread_create(&tid[g], NULL, gpu_main, &args[g]);
    }
    for (g = 0; g < ngpus; g++) {
        pthread_join(tid[g], NULL);
        rc = rc ? rc : args[g].rc;
        t_pump += args[g].t_pump / ngpus;
        t_read += args[g].t_read / ngpus;
        if (args[g].t_pump + args[g].t_read > t_slowest)
            t_slowest = args[g].t_pump + args[g].t_read;
        sum += args[g].sum;
    }
    if (rc)
        return rc;
    printf("streams %u block %zu pull threads %u: pump %.3f ms, readers %.3f ms per block -> %.0f Msamples/s (checksum %llu)%s",
           streams, block, pull_threads, t_pump / rounds, t_read / rounds,
           (double)streams * block / (t_slowest / rounds * 1e-3) / 1e6, sum, ngpus > 1 || want_node ? "" : "\n");
    if (ngpus > 1 || want_node)
        printf(" on %d GPU(s), stream s on GPU s %% %d, one pump thread each\n", ngpus, ngpus);
    for (g = 0; g < ngpus; g++)
        if (args[g].has_first)
            printf("stream 0: frames %zu peak %d power %.17g\n", args[g].first.frames, (int)args[g].first.global_peak,
                   args[g].first.global_power);
    for (g = 0; g < ngpus; g++)
        if (args[g].has_last)
            printf("stream %u: frames %zu peak %d power %.17g\n", streams - 1, args[g].last.frames,
                   (int)args[g].last.global_peak, args[g].last.global_power);
    if (want_node) {
        int64_t *records = calloc((size_t)ngpus * CMHIP_NODE_WORDS, sizeof(*records)), merged[CMHIP_NODE_WORDS];
        coolmic_vumeter_result_t res;
        int all_rccl = have_node_id, same = 1;
        for (g = 0; g < ngpus; g++) {
            memcpy(records + (size_t)g * CMHIP_NODE_WORDS, args[g].record, sizeof(args[g].record));
            all_rccl = all_rccl && args[g].rccl;
        }
        if (cmhip_node_merge_host(records, (unsigned)ngpus, merged) != COOLMIC_ERROR_NONE ||
            cmhip_node_finish(merged, 1, 48000, &res) != COOLMIC_ERROR_NONE)
            return 4;
        for (g = 0; all_rccl && g < ngpus; g++)
            same = same && memcmp(args[g].combined, merged, sizeof(merged)) == 0;
        printf("node: frames %zu peak %d power %.17g (%d GPU(s); host merge%s)\n", res.frames, (int)res.global_peak,
               res.global_power, ngpus, !all_rccl ? ", no RCCL exchange" : same ? " == RCCL exchange" : " != RCCL exchange");
        free(records);
        if (all_rccl && !same)
            return 5;
    }
    pthread_barrier_destroy(&start);
    free(args);
    free(tid);
    return 0;
}
