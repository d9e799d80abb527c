// cmhip_place.hip -- the opt-in placement search for a batch's two PCM arrays (CMHIP_PLACE_SEARCH).
#include "cmhip_engine.h"

#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <mutex>

// Placement of a batch's two PCM arrays.  On MI355X a kernel that streams one large array in and another
// out runs 3-5 % faster when the two lie in different stretches of the card's memory (measured:
// tools/placement_*.py, DESIGN 4.1 -- physical memory falls into stretches of up to 32 GiB of
// three kinds; reads and writes that go to the same kind get in each other's way, and of the pairs
// of different kinds one is better than the others).  Nothing but the virtual address is visible from
// here, so the arrays are chosen by probing (place_arrays_apart, below): more candidates behind spacer
// allocations, the batch's own run on every pair of them.
// The search is the CALLER's decision (CMHIP_PLACE_SEARCH in the batch's flags): a library must not, by
// default, take tens of GiB for a moment and seconds of a constructor.  Only for arrays of 256 MiB and
// more, never more than PLACE_BUDGET_FRAC of the memory reported free, allocations stop after 0.3 s;
// what it did is in cmhip_batch_placement().
constexpr size_t PLACE_MIN_BYTES = 256ull << 20;
// Spacers before candidates 2, 3, ...: 68 GiB in all reach past two whole stretches.  (Larger ones reach further
// -- 4 ... 32 GiB, 124 in all, found the best kind of pair more often -- but allocating from memory that this or
// an earlier process has freed is slow on this driver, which hands out cleared pages: single allocations of
// 16-32 GiB were seen to take 3-6 s.)
constexpr size_t PLACE_SPACER_GIB[] = {0, 4, 8, 16, 16, 24};
constexpr int PLACE_TRIES = 6;
constexpr double PLACE_BUDGET_S = 0.3;
constexpr double PLACE_BUDGET_FRAC = 0.5;

// a probe: the batch's own run (as created: no gain, no maps), full slots, from one candidate into another
static double place_probe_ms(cmhip_batch_t *b, const void *src, void *dst, hipEvent_t e0, hipEvent_t e1)
{
    RunArgs a;
    memset(&a, 0, sizeof(a));
    a.in = (const int16_t *)src;
    a.out = (int16_t *)dst;
    a.f32 = b->d_f32;
    a.param = b->d_param;
    a.vu = (b->d.flags & CMHIP_VU) ? b->d_vu : nullptr;
    a.frames = (uint32_t)b->d.max_frames;
    a.streams = b->d.streams;
    a.channels = b->d.channels;
    a.stride = b->stride;
    a.plane = b->plane;
    a.identity_maps = 1;
    a.identity_gains = 1;
    const int n = 6;
    for (int i = 0; i < 2; i++)
        if (launch_run(a, b->tune, b->stream) != hipSuccess)
            return -1.;
    if (hipEventRecord(e0, b->stream) != hipSuccess)
        return -1.;
    for (int i = 0; i < n; i++)
        if (launch_run(a, b->tune, b->stream) != hipSuccess)
            return -1.;
    float ms = 0.f;
    if (hipEventRecord(e1, b->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
        hipEventElapsedTime(&ms, e0, e1) != hipSuccess)
        return -1.;
    b->place.probe_launches += n + 2;
    return (double)ms / n;
}

// Who searches: a batch created with CMHIP_PLACE_SEARCH, always (the caller asked).  $CMHIP_PLACE, for
// experiments: 0 nobody, 1 also the first large batch of a device in this process without the flag, 2 every
// large batch.
static bool place_search_allowed(const cmhip_batch_t *b)
{
    static std::mutex mu;
    static bool searched[64];
    if (b->tune.place_env == 0)
        return false;
    if ((b->d.flags & CMHIP_PLACE_SEARCH) || b->tune.place_env == 2)
        return true;
    if (b->tune.place_env != 1)
        return false;
    std::lock_guard<std::mutex> g(mu);
    const int d = b->d.device;
    if (d < 0 || d >= 64 || searched[d])
        return false;
    searched[d] = true;
    return true;
}

// The batch has its two PCM arrays where hipMalloc first put them (candidates 0 and 1).  More candidates
// follow behind spacers; every pair of candidates is a possible (input, output) -- nothing is in the arrays
// yet -- and the pair the batch's own run is fastest on is kept if it beats the first by more than 2 %.
int cmhip_engine_place_arrays_apart(cmhip_batch_t *b, size_t bytes)
{
    bool probed = false;
    void *cand[PLACE_TRIES + 1] = {nullptr}, *spacer[PLACE_TRIES + 1] = {nullptr};
    cand[0] = b->d_in;
    cand[1] = b->d_out;
    int n = 2, in = 0, out = 1;
    size_t free_b = 0, total_b = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    cmhip_placement_t &rec = b->place;
    rec.chosen_in = 0;
    rec.chosen_out = 1;
    rec.candidates = 2;
    if (bytes >= PLACE_MIN_BYTES && place_search_allowed(b) && hipMemGetInfo(&free_b, &total_b) == hipSuccess &&
        hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
        probed = true;
        rec.searched = 1;
        rec.bytes_free_before = free_b;
        // never more than a stated share of what the card reports free, spacers and candidates together:
        // on a fuller card the search reaches less far (fewer candidates), it does not crowd a neighbour out
        const size_t budget = (size_t)((double)free_b * PLACE_BUDGET_FRAC);
        size_t asked = 0;
        // (allocations of this size are normally a few milliseconds; from memory that has been used and freed
        // the driver has been seen to take seconds: then what there is by then decides)
        const auto t_begin = std::chrono::steady_clock::now();
        auto elapsed = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); };
        for (int k = 1; k < PLACE_TRIES && elapsed() <= PLACE_BUDGET_S; k++) {
            const size_t sp = PLACE_SPACER_GIB[k] << 30;
            if (asked + sp + bytes > budget)
                break;
            if (hipMalloc(&spacer[n], sp) != hipSuccess || hipMalloc(&cand[n], bytes) != hipSuccess) {
                (void)hipGetLastError();                  // no room after all
                if (spacer[n])
                    asked += sp;
                break;
            }
            asked += sp + bytes;
            n++;
        }
        rec.bytes_requested = asked;
        rec.candidates = n;
        // Samples that are not zero: a tile of silence adds nothing to its window and skips its atomics, and
        // without them the kinds of pairs lie closer together (3.5 % instead of 5 %: probes on cleared arrays
        // took a pair of one kind for a good one).
        for (int k = 0; k < n; k++)
            if (hipMemsetAsync(cand[k], 0x5a, bytes, b->stream) != hipSuccess)
                break;
        // (the card may come from idle: the probes compare like with like only at settled clocks)
        for (int i = 0; i < 12; i++)
            if (place_probe_ms(b, cand[0], cand[1], e0, e1) < 0.)
                break;
        // (only the best kind of pair is worth taking: 3-7 % faster than a pair of one kind; differences of
        // 1-2 % between pairs do not last)
        double refs[PLACE_TRIES + 1], tbest = 0.;
        int nref = 0, bi = 0, bj = 1;
        for (int i = 0; i < n; i++) {
            const double ref = place_probe_ms(b, cand[0], cand[1], e0, e1);     // (again per row: clocks drift)
            if (ref > 0.)
                refs[nref++] = ref;
            for (int j = i + 1; j < n && ref > 0.; j++) {
                if (i == 0 && j == 1)
                    continue;
                const double t = place_probe_ms(b, cand[i], cand[j], e0, e1);
                if (b->tune.place_debug)
                    fprintf(stderr, "cmhip place: %d -> %d: %.4f ms (0 -> 1: %.4f ms)\n", i, j, t, ref);
                if (t > 0. && (tbest == 0. || t < tbest)) {
                    tbest = t;
                    bi = i;
                    bj = j;
                }
            }
        }
        // the first pair's time: the median of its samples (they scatter by 1 %); the fastest other pair is
        // taken if it is 2 % faster than that (a pair of the best kind is 5-8 % faster than one of one kind, a
        // middling one 3 %; moving the arrays for nothing costs nothing)
        std::sort(refs, refs + nref);
        if (nref)
            rec.first_pair_ms = refs[nref / 2];
        rec.best_pair_ms = tbest;
        if (nref && tbest > 0. && tbest < 0.98 * refs[nref / 2]) {
            in = bi;
            out = bj;
        }
        rec.chosen_in = in;
        rec.chosen_out = out;
        rec.search_ms = 1e3 * elapsed();
        if (b->tune.place_debug)
            fprintf(stderr, "cmhip place: input = candidate %d, output = candidate %d, %.0f ms, %.1f GiB asked of %.1f free\n",
                    in, out, rec.search_ms, (double)asked / (1ull << 30), (double)free_b / (1ull << 30));
    }
    if (e0)
        (void)hipEventDestroy(e0);
    if (e1)
        (void)hipEventDestroy(e1);
    for (int k = 0; k <= PLACE_TRIES; k++) {              // (all of them: a spacer may be there without its candidate)
        if (spacer[k])
            (void)hipFree(spacer[k]);
        if (cand[k] && k != in && k != out)
            (void)hipFree(cand[k]);
    }
    if (!probed)
        return COOLMIC_ERROR_NONE;
    // the probes ran the batch's kernel: whatever they left in the arrays, the windows and the float planes goes
    b->d_in = (int16_t *)cand[in];
    b->d_out = (int16_t *)cand[out];
    HIP_TRY(hipMemsetAsync(b->d_in, 0, bytes, b->stream));
    HIP_TRY(hipMemsetAsync(b->d_out, 0, bytes, b->stream));
    if (b->d.flags & CMHIP_VU)
        for (int i = 0; i < 3; i++)
            HIP_TRY(hipMemsetAsync(b->d_vu2[i], 0, b->d.streams * sizeof(VuState), b->stream));
    if (b->d_f32)
        HIP_TRY(hipMemsetAsync(b->d_f32, 0, b->d.streams * b->d.channels * b->plane * sizeof(float), b->stream));
    return COOLMIC_ERROR_NONE;
}

extern "C" int cmhip_batch_placement(const cmhip_batch_t *b, cmhip_placement_t *out)
{
    if (!b || !out)
        return fail(COOLMIC_ERROR_FAULT, "placement: NULL argument");
    *out = b->place;
    return COOLMIC_ERROR_NONE;
}

