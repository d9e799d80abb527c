// cmhip_measure.hip -- measurement hooks of the engine: kernel timing from the dispatch's own events, plain HBM
// ceilings on a batch's buffers.
#include "cmhip_engine.h"

// ---------------------------------------------------------------------------
// measurement

extern "C" int cmhip_batch_timing(cmhip_batch_t *b, int enable)
{
    if (!b)
        return fail(COOLMIC_ERROR_FAULT, "timing: batch is NULL");
    b->timing = enable != 0;
    b->timing_every = enable > 1 ? (unsigned)enable : 1u;
    b->timing_count = 0;
    return COOLMIC_ERROR_NONE;
}

extern "C" int cmhip_batch_timing_read(cmhip_batch_t *b, double *kernel_ms, unsigned int *launches)
{
    if (!b)
        return fail(COOLMIC_ERROR_FAULT, "timing_read: batch is NULL");
    if (use(b))
        return COOLMIC_ERROR_GENERIC;
    HIP_TRY(hipStreamSynchronize(b->stream));
    double ms = 0.;
    for (auto &e : b->ev_used) {
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, e.a, e.b));
        ms += t;
    }
    if (kernel_ms)
        *kernel_ms = ms;
    if (launches)
        *launches = (unsigned)b->ev_used.size();
    b->ev_free.insert(b->ev_free.end(), b->ev_used.begin(), b->ev_used.end());
    b->ev_used.clear();
    return COOLMIC_ERROR_NONE;
}

extern "C" double cmhip_batch_ceiling(cmhip_batch_t *b, int mode, size_t frames, int iters)
{
    if (!b || iters <= 0 || frames > b->d.max_frames || (mode == 1 && (!b->d_out || b->d_out == b->d_in))) {
        fail(COOLMIC_ERROR_INVAL, "ceiling: bad arguments (copy needs a separate PCM output)");
        return -1.;
    }
    if (hipSetDevice(b->d.device) != hipSuccess)
        return -1.;
    // whole slots, so that the byte count is exact and contiguous
    (void)frames;
    const size_t bytes = (size_t)b->d.streams * b->stride * sizeof(int16_t);
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
        return -1.;
    for (int i = 0; i < 2; i++)
        (void)launch_ceiling(mode, b->d_in, b->d_out, bytes, b->d_sink, b->stream);
    (void)hipEventRecord(e0, b->stream);
    for (int i = 0; i < iters; i++)
        (void)launch_ceiling(mode, b->d_in, b->d_out, bytes, b->d_sink, b->stream);
    (void)hipEventRecord(e1, b->stream);
    float ms = 0.f;
    if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) {
        fail(COOLMIC_ERROR_GENERIC, "ceiling: event timing failed");
        return -1.;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    const double moved = (double)bytes * (mode == 1 ? 2. : 1.) * iters;
    return moved / (ms * 1e-3) / 1e9;
}

