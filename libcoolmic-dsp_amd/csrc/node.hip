// node.hip -- the node-global VU exchange of BASELINE config 5 in C over RCCL:
// cmhip_node_* of include/coolmic_hip.h.  One cmhip_node_t per GPU; the records of B blocks are
// reduced over the ranks with one ncclAllReduce(ncclInt64, ncclSum) over their sums and one
// ncclAllReduce(ncclUint64, ncclMax) over their packed peak keys, fused in one RCCL group on
// the node's own HIP stream.  No torch, no Python: a C host drives it directly
// (examples/node_vu.c); bench.py --workload c5 goes through the same entry points.
//
// librccl.so.1 is a 570 MB library, so it is not a link-time dependency of the engine: it is
// loaded when the first node is created and only its six entry points are resolved.
#include "cmhip_internal.h"

#include <coolmic-dsp/coolmic-dsp.h>
#include <coolmic_hip.h>

#include <rccl/rccl.h>

#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

using namespace cmhip;

#define fail cmhip_fail
#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(COOLMIC_ERROR_GENERIC, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                \
    } while (0)

namespace {

struct Rccl {
    void *lib;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *);
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*GroupStart)(void);
    ncclResult_t (*GroupEnd)(void);
    const char *(*GetErrorString)(ncclResult_t);
};

Rccl g_rccl;
std::once_flag g_rccl_once;
char g_rccl_error[256];

char g_rccl_path[512], g_hip_path[512];

void load_rccl()
{
    // $CMHIP_RCCL_LIB first; then the librccl that lies NEXT TO the HIP runtime this engine is bound to
    // (dladdr of a HIP entry point): a process may hold two HIP runtimes -- the system's and the one inside
    // a torch wheel -- each with a librccl of the same soname, and device pointers of one runtime mean
    // nothing to a librccl bound to the other.  By soname only after that.
    std::string a, b;
    Dl_info info;
    if (dladdr((void *)&hipGetDeviceCount, &info) && info.dli_fname) {
        snprintf(g_hip_path, sizeof(g_hip_path), "%s", info.dli_fname);
        std::string dir(info.dli_fname);
        const size_t cut = dir.rfind('/');
        if (cut != std::string::npos) {
            a = dir.substr(0, cut) + "/librccl.so.1";
            b = dir.substr(0, cut) + "/librccl.so";
        }
    }
    const char *names[] = {getenv("CMHIP_RCCL_LIB"), a.c_str(), b.c_str(), "librccl.so.1",
                           "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    void *h = nullptr;
    for (const char *n : names) {
        if (n && *n && (h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) {
            snprintf(g_rccl_path, sizeof(g_rccl_path), "%s", n);
            break;
        }
    }
    if (!h) {
        snprintf(g_rccl_error, sizeof(g_rccl_error), "librccl not found: %s", dlerror());
        return;
    }
    Rccl r{};
    r.lib = h;
#define SYM(field, name)                                                       \
    do {                                                                       \
        *(void **)(&r.field) = dlsym(h, name);                                 \
        if (!r.field) {                                                        \
            snprintf(g_rccl_error, sizeof(g_rccl_error), "librccl lacks %s", name); \
            return;                                                            \
        }                                                                      \
    } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllReduce, "ncclAllReduce");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl = r;
}

const Rccl *rccl()
{
    std::call_once(g_rccl_once, load_rccl);
    return g_rccl.lib ? &g_rccl : nullptr;
}

}  // namespace

#define NCCL_TRY(expr)                                                                      \
    do {                                                                                    \
        ncclResult_t r_ = (expr);                                                           \
        if (r_ != ncclSuccess)                                                              \
            return fail(COOLMIC_ERROR_GENERIC, "%s: %s (%s:%d)", #expr, rc->GetErrorString(r_), \
                        __FILE__, __LINE__);                                                \
    } while (0)

constexpr unsigned NODE_SETS = 2;
constexpr unsigned HALF = CMHIP_NODE_SUM_WORDS;           // 17 sums, 17 keys per record

struct cmhip_node {
    int device, nranks, rank;
    unsigned int max_records;
    ncclComm_t comm;
    hipStream_t stream;                    // the collectives run here, beside the batches' streams
    long long *d_words;                    // [set][ sums: max_records x 17 | keys: max_records x 17 ]
    long long *h_words;                    // pinned, same shape, for fetch
    hipEvent_t ev_filled;                  // "the batch has written its records" (recorded on its stream)
    hipEvent_t ev_done[NODE_SETS];         // the exchange of a set has finished
    bool exchanged[NODE_SETS];             // ev_done[set] has been recorded since the set was last filled
    // A slot is cleared before a batch adds into it -- not slot by slot (two memsets per block) but
    // the whole set at once, when the first block after an exchange arrives; `filled` marks the
    // slots written since.
    std::vector<bool> filled[NODE_SETS];
    bool side;                             // records are built beside the batch's next run (its copy stream)
};

static long long *set_sums(const cmhip_node_t *n, unsigned set)
{
    return n->d_words + (size_t)set * 2u * HALF * n->max_records;
}
static long long *set_keys(const cmhip_node_t *n, unsigned set)
{
    return set_sums(n, set) + (size_t)HALF * n->max_records;
}

extern "C" int cmhip_node_unique_id(void *id128)
{
    if (!id128)
        return fail(COOLMIC_ERROR_FAULT, "node_unique_id: NULL argument");
    const Rccl *rc = rccl();
    if (!rc)
        return fail(COOLMIC_ERROR_NOSYS, "node_unique_id: %s", g_rccl_error);
    static_assert(sizeof(ncclUniqueId) == CMHIP_NODE_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    NCCL_TRY(rc->GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return COOLMIC_ERROR_NONE;
}

extern "C" void cmhip_node_free(cmhip_node_t *n)
{
    if (!n)
        return;
    (void)hipSetDevice(n->device);
    if (n->stream)
        (void)hipStreamSynchronize(n->stream);
    if (n->comm && rccl())
        (void)rccl()->CommDestroy(n->comm);
    for (unsigned i = 0; i < NODE_SETS; i++)
        if (n->ev_done[i])
            (void)hipEventDestroy(n->ev_done[i]);
    if (n->ev_filled)
        (void)hipEventDestroy(n->ev_filled);
    (void)hipFree(n->d_words);
    if (n->h_words)
        (void)hipHostFree(n->h_words);
    if (n->stream)
        (void)hipStreamDestroy(n->stream);
    delete n;
}

static int node_init(cmhip_node_t *n, const Rccl *rc, const void *id128)
{
    HIP_TRY(hipSetDevice(n->device));
    {
        int least = 0, greatest = 0;             // (beside the batches' long kernels: see their copy streams)
        HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_TRY(hipStreamCreateWithPriority(&n->stream, hipStreamNonBlocking,
                                            getenv("CMHIP_SIDE_PRIORITY_OFF") ? least : greatest));
    }
    const size_t bytes = (size_t)NODE_SETS * 2u * HALF * n->max_records * sizeof(long long);
    HIP_TRY(hipMalloc((void **)&n->d_words, bytes));
    HIP_TRY(hipMemset(n->d_words, 0, bytes));
    HIP_TRY(hipHostMalloc((void **)&n->h_words, bytes, hipHostMallocDefault));
    HIP_TRY(hipEventCreateWithFlags(&n->ev_filled, hipEventDisableTiming));
    for (unsigned i = 0; i < NODE_SETS; i++)
        HIP_TRY(hipEventCreateWithFlags(&n->ev_done[i], hipEventDisableTiming));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    NCCL_TRY(rc->CommInitRank(&n->comm, n->nranks, id, n->rank));
    return COOLMIC_ERROR_NONE;
}

extern "C" cmhip_node_t *cmhip_node_new(int device, int nranks, int rank, const void *id128,
                                        unsigned int max_records)
{
    if (!id128) {
        fail(COOLMIC_ERROR_FAULT, "node_new: id is NULL");
        return nullptr;
    }
    if (nranks < 1 || rank < 0 || rank >= nranks || max_records == 0 || max_records > 4096) {
        fail(COOLMIC_ERROR_INVAL, "node_new: rank %d of %d, %u records: out of range", rank, nranks, max_records);
        return nullptr;
    }
    if (device < 0 || device >= cmhip_device_count()) {
        fail(COOLMIC_ERROR_NOSYS, "node_new: no HIP device %d", device);
        return nullptr;
    }
    const Rccl *rc = rccl();
    if (!rc) {
        fail(COOLMIC_ERROR_NOSYS, "node_new: %s", g_rccl_error);
        return nullptr;
    }
    cmhip_node_t *n = new cmhip_node();
    n->comm = nullptr;
    n->stream = nullptr;
    n->d_words = n->h_words = nullptr;
    n->ev_filled = nullptr;
    for (unsigned i = 0; i < NODE_SETS; i++) {
        n->ev_done[i] = nullptr;
        n->exchanged[i] = false;
        n->filled[i].assign(max_records, false);       // (the buffer starts zeroed)
    }
    n->side = getenv("CMHIP_NODE_MAIN_STREAM") == nullptr;       // (A/B knob, read once)
    n->device = device;
    n->nranks = nranks;
    n->rank = rank;
    n->max_records = max_records;
    if (node_init(n, rc, id128) != COOLMIC_ERROR_NONE) {
        cmhip_node_free(n);
        return nullptr;
    }
    return n;
}

extern "C" int cmhip_node_ranks(const cmhip_node_t *n) { return n ? n->nranks : 0; }

// diagnostics: which HIP runtime the engine is bound to and which librccl it resolved (loads librccl)
extern "C" const char *cmhip_node_runtime(void)
{
    static char text[1100];
    const Rccl *rc = rccl();
    snprintf(text, sizeof(text), "hip=%s rccl=%s", g_hip_path[0] ? g_hip_path : "?", rc ? g_rccl_path : g_rccl_error);
    return text;
}

extern "C" int cmhip_node_partial(cmhip_node_t *n, cmhip_batch_t *b, unsigned int set, unsigned int slot,
                                  uint64_t first_global, uint64_t global_step)
{
    if (!n || !b)
        return fail(COOLMIC_ERROR_FAULT, "node_partial: NULL argument");
    if (set >= NODE_SETS || slot >= n->max_records)
        return fail(COOLMIC_ERROR_INVAL, "node_partial: set %u / slot %u out of range", set, slot);
    if (cmhip_batch_device(b) != n->device || !(cmhip_batch_flags(b) & CMHIP_VU))
        return fail(COOLMIC_ERROR_INVAL, "node_partial: the batch must have VU windows on device %d", n->device);
    HIP_TRY(hipSetDevice(n->device));
    hipStream_t bs = (hipStream_t)(n->side ? cmhip_batch_side_stream(b) : cmhip_batch_hip_stream(b));
    if (n->exchanged[set]) {               // the set's last exchange must be through before it is refilled
        if (hipEventQuery(n->ev_done[set]) != hipSuccess)
            HIP_TRY(hipStreamWaitEvent(bs, n->ev_done[set], 0));
        n->exchanged[set] = false;
        HIP_TRY(hipMemsetAsync(set_sums(n, set), 0, 2u * (size_t)HALF * n->max_records * sizeof(long long), bs));
        n->filled[set].assign(n->max_records, false);
    } else if (n->filled[set][slot]) {     // written twice without an exchange between: start it afresh
        HIP_TRY(hipMemsetAsync(set_sums(n, set) + (size_t)slot * HALF, 0, HALF * sizeof(long long), bs));
        HIP_TRY(hipMemsetAsync(set_keys(n, set) + (size_t)slot * HALF, 0, HALF * sizeof(long long), bs));
    }
    n->filled[set][slot] = true;
    if (!n->side)
        return cmhip_batch_node_partial_split(b, set_sums(n, set) + (size_t)slot * HALF,
                                              set_keys(n, set) + (size_t)slot * HALF, first_global, global_step, 0);
    return cmhip_batch_node_partial_side(b, set_sums(n, set) + (size_t)slot * HALF,
                                         set_keys(n, set) + (size_t)slot * HALF, first_global, global_step);
}

extern "C" int cmhip_node_allreduce(cmhip_node_t *n, unsigned int set, unsigned int count, cmhip_batch_t *after)
{
    if (!n)
        return fail(COOLMIC_ERROR_FAULT, "node_allreduce: node is NULL");
    if (set >= NODE_SETS || count == 0 || count > n->max_records)
        return fail(COOLMIC_ERROR_INVAL, "node_allreduce: set %u / %u records out of range", set, count);
    const Rccl *rc = rccl();
    HIP_TRY(hipSetDevice(n->device));
    if (after) {
        HIP_TRY(hipEventRecord(n->ev_filled, (hipStream_t)(n->side ? cmhip_batch_side_stream(after)
                                                                   : cmhip_batch_hip_stream(after))));
        HIP_TRY(hipStreamWaitEvent(n->stream, n->ev_filled, 0));
    }
    // sums of all slots, then keys of all slots: two collectives, one launch (keys are below 2^63
    // either way; ncclUint64 is what they are)
    NCCL_TRY(rc->GroupStart());
    ncclResult_t r1 = rc->AllReduce(set_sums(n, set), set_sums(n, set), (size_t)count * HALF, ncclInt64, ncclSum,
                                    n->comm, n->stream);
    ncclResult_t r2 = rc->AllReduce(set_keys(n, set), set_keys(n, set), (size_t)count * HALF, ncclUint64, ncclMax,
                                    n->comm, n->stream);
    NCCL_TRY(rc->GroupEnd());
    NCCL_TRY(r1);
    NCCL_TRY(r2);
    HIP_TRY(hipEventRecord(n->ev_done[set], n->stream));
    n->exchanged[set] = true;
    return COOLMIC_ERROR_NONE;
}

extern "C" int cmhip_node_fetch(cmhip_node_t *n, unsigned int set, unsigned int count, int64_t *words)
{
    if (!n || !words)
        return fail(COOLMIC_ERROR_FAULT, "node_fetch: NULL argument");
    if (set >= NODE_SETS || count == 0 || count > n->max_records)
        return fail(COOLMIC_ERROR_INVAL, "node_fetch: set %u / %u records out of range", set, count);
    HIP_TRY(hipSetDevice(n->device));
    const size_t set_words = 2u * (size_t)HALF * n->max_records;
    long long *h = n->h_words + set * set_words;
    // on the node's stream: behind the exchange if one was queued
    HIP_TRY(hipMemcpyAsync(h, set_sums(n, set), set_words * sizeof(long long), hipMemcpyDeviceToHost, n->stream));
    HIP_TRY(hipStreamSynchronize(n->stream));
    for (unsigned i = 0; i < count; i++) {
        memcpy(words + (size_t)i * CMHIP_NODE_WORDS, h + (size_t)i * HALF, HALF * sizeof(int64_t));
        memcpy(words + (size_t)i * CMHIP_NODE_WORDS + HALF, h + (size_t)HALF * n->max_records + (size_t)i * HALF,
               HALF * sizeof(int64_t));
    }
    return COOLMIC_ERROR_NONE;
}

extern "C" int cmhip_node_merge_host(const int64_t *records, unsigned int nranks, int64_t *out)
{
    if (!records || !out)
        return fail(COOLMIC_ERROR_FAULT, "node_merge_host: NULL argument");
    if (nranks == 0)
        return fail(COOLMIC_ERROR_INVAL, "node_merge_host: no records");
    for (unsigned w = 0; w < CMHIP_NODE_WORDS; w++) {
        unsigned long long acc = 0;
        for (unsigned r = 0; r < nranks; r++) {
            const unsigned long long v = (unsigned long long)records[(size_t)r * CMHIP_NODE_WORDS + w];
            acc = w < HALF ? acc + v : (v > acc ? v : acc);
        }
        out[w] = (int64_t)acc;
    }
    return COOLMIC_ERROR_NONE;
}
