// cmhip_engine.h -- the batch object and what the engine's translation units share (not part of the C ABI):
//   cmhip_batch.hip    the object, parameters, transfers, the run
//   cmhip_place.hip    the opt-in placement search for a batch's two PCM arrays
//   cmhip_vu.hip       VU windows: results, packed snapshots and their collect, window records, node records
//   cmhip_measure.hip  kernel timing and the plain HBM ceilings
#pragma once

#include "cmhip_internal.h"

#include <coolmic-dsp/coolmic-dsp.h>
#include <coolmic_hip.h>

#include <vector>

#include "work_pool.h"
#include "host_internal.h"

using namespace cmhip;

#define fail cmhip_fail

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(COOLMIC_ERROR_GENERIC, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                \
    } while (0)

constexpr unsigned STAGE_SLOTS = 4;
constexpr size_t STAGE_BYTES = 64 * 1024;

struct EventPair {
    hipEvent_t a, b;
};

struct cmhip_batch {
    cmhip_batch_desc_t d;
    hipStream_t stream;
    bool own_stream;
    size_t stride;                 // samples between slots
    size_t plane;                  // floats between f32 planes

    int16_t *d_in, *d_out;
    int16_t *h_in, *h_out;         // CMHIP_HOSTPCM: the slots live in pinned host memory (d_* alias them)
    bool in_flight;                // a launch may still be using the slots (CMHIP_HOSTPCM)
    // CMHIP_HOSTPCM: a launch of one workgroup reports its end through a word in pinned, device-mapped host
    // memory (RunArgs::done_flag) and the host spins on it -- 4-5 us less per pull than waiting for the stream
    uint32_t *h_done, *d_done;
    uint32_t done_seq;             // the last sequence number handed to a launch
    bool done_flagged;             // ... and that launch carries the flag
    float *d_f32;
    StreamParam *d_param;
    VuState *d_vu;                         // the window runs accumulate into (= d_vu2[cur])
    VuState *d_vu2[3];                     // three sets in rotation: accumulating / being copied out / cleared
    unsigned int cur;
    hipStream_t copy_stream;               // snapshots travel here, beside the next run
    hipEvent_t ev_main, ev_reset[3];
    // A node partial (cmhip_node_partial) reads the current windows on the copy stream, beside the next
    // run; node_reading is set until the window set has rotated (snapshot) or the main stream has been
    // made to wait for the copy stream (settle_node: before anything on the main stream touches them).
    hipEvent_t ev_node;
    bool node_reading;
    // The end of the last run as its own dispatch stamped it (hipExtLaunchKernelGGL): what a snapshot
    // makes the copy stream wait for instead of an event recorded behind the kernel -- one packet less
    // on the main stream per step.  nullptr once anything else on the main stream touched the windows.
    hipEvent_t ev_done[4], last_done;
    unsigned done_next;
    bool reset_pending[3];
    struct WorkPool *pool;
    uint32_t *d_nframes;
    EqParam *d_eq;
    EqState *d_eqstate;
    unsigned long long *d_sink;
    long long *d_node_scratch;             // one node record, for cmhip_batch_vu_node_record (made on first use)
    // ring mode (cmhip_batch_vu_ring): every run accumulates into a window of its own
    VuState *d_ring, *h_ring;              // ring_slots x S windows on the device / pinned staging for a fetch
    unsigned int ring_slots;
    uint64_t ring_seq;                     // sequence number of the next run
    uint64_t ring_fetched;                 // runs below this sequence number have been fetched: their slots are clear
    unsigned long long *d_dbg;             // 64 words, written only by diagnostic builds

    std::vector<StreamParam> h_param;
    std::vector<uint16_t> h_scale;         // the reference's master_gain_scale per stream
    std::vector<uint16_t> h_gain;          // [S][16]
    bool param_dirty;
    bool all_identity;                     // no stream has a channel map (recomputed on upload)
    bool all_gain_identity;                // no stream has a gain (same)
    std::vector<EqParam> h_eq;
    unsigned int nsec;
    bool eq_dirty;

    // three snapshots may be pending (one being finished by the helper threads, one waiting, one on its way):
    // each a packed copy of a window set, [1 + 2C][S] words in pinned, device-mapped host memory that
    // k_vu_pack writes itself (h_pack / d_pack: host / device view)
    unsigned long long *h_pack[3], *d_pack[3];
    unsigned int snap_set2[3];             // which of the three window sets the snapshot closed (its event: ev_reset)
    bool collecting;                       // between cmhip_batch_vu_collect_begin and _end
    coolmic_vumeter_result_t *job_out;
    int *job_rc;
    unsigned int job_slot;
    unsigned int snap_head, snap_count;    // ring of pending snapshots (oldest = head)
    unsigned char *h_stage;                // pinned upload ring, STAGE_SLOTS x STAGE_BYTES
    hipEvent_t stage_ev[4];
    bool stage_busy[4];
    unsigned int stage_next;
    unsigned int parity;                   // current slot of VuState::samples

    bool timing;
    unsigned int timing_every, timing_count;     // every n-th run carries the events (cmhip_batch_timing)
    std::vector<EventPair> ev_used, ev_free;
    RunTune tune;                          // launcher knobs, read once at creation
    cmhip_placement_t place;               // what the placement search did (cmhip_batch_placement)
    bool vu_off;                           // runs leave the windows alone for now (cmhip_batch_vu_pause)
};


static inline int use(cmhip_batch_t *b)
{
    HIP_TRY(hipSetDevice(b->d.device));
    return COOLMIC_ERROR_NONE;
}

// the main stream is about to touch the windows a node partial may still be reading on the copy stream
CMHIP_INTERNAL int cmhip_engine_settle_node(cmhip_batch_t *b);
// parameter and equaliser tables to the device, when a setter has run since the last upload
CMHIP_INTERNAL int cmhip_engine_flush_params(cmhip_batch_t *b);
// cmhip_place.hip: called once, at the end of a batch's creation, when it has PCM arrays of its own
CMHIP_INTERNAL int cmhip_engine_place_arrays_apart(cmhip_batch_t *b, size_t bytes);
