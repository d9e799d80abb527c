// k_misc.hip -- small kernels beside the path: synthetic inputs, the node-global VU partial,
// plain HBM ceilings.
#include "cmhip_device.h"

namespace cmhip {

// ---------------------------------------------------------------------------
// Synthetic inputs, bit-identical to the host generators.

__constant__ u32 c_lcg_a[32];
__constant__ u32 c_lcg_c[32];

__device__ __forceinline__ u32 lcg_skip(u32 state, u64 n)
{
    for (int i = 0; i < 32 && n; i++, n >>= 1)
        if (n & 1)
            state = state * c_lcg_a[i] + c_lcg_c[i];
    return state;
}

__global__ __launch_bounds__(256) void k_generate(GenArgs g, int mode, u32 vec_per_stream)
{
    const u64 gid = (u64)blockIdx.x * 256u + threadIdx.x;
    const u32 s = (u32)(gid / vec_per_stream);
    const u32 v = (u32)(gid - (u64)s * vec_per_stream);
    if (s >= g.streams)
        return;
    const u64 gs = g.first_global + (u64)s * g.global_step;      // global stream id
    const u32 nsamp = g.frames * g.channels;
    int16_t *dst = g.dst + (u64)s * g.stride + (u64)v * 8;
    int16_t val[8];
    if (mode == 2) {
        u32 st = lcg_skip(g.seed + (u32)gs, g.frame_offset * g.channels + (u64)v * 8);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            st = st * 1664525u + 1013904223u;
            val[j] = (int16_t)(st >> 16);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const u64 f = g.frame_offset + ((u64)v * 8 + j) / g.channels;
            val[j] = mode == 1 ? g.sine[(f + 7ull * gs) % 48ull] : (int16_t)0;
        }
    }
    if (v * 8u + 8u <= nsamp) {
        uint4 w;
        w.x = (u32)(uint16_t)val[0] | ((u32)(uint16_t)val[1] << 16);
        w.y = (u32)(uint16_t)val[2] | ((u32)(uint16_t)val[3] << 16);
        w.z = (u32)(uint16_t)val[4] | ((u32)(uint16_t)val[5] << 16);
        w.w = (u32)(uint16_t)val[6] | ((u32)(uint16_t)val[7] << 16);
        *reinterpret_cast<uint4 *>(dst) = w;
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (v * 8u + (u32)j < nsamp)
                dst[j] = val[j];
    }
}

hipError_t launch_generate(const GenArgs &g, int mode, hipStream_t st)
{
    static bool tables = false;
    if (!tables) {
        u32 a[32], c[32];
        u32 aa = 1664525u, cc = 1013904223u;
        for (int i = 0; i < 32; i++) {
            a[i] = aa;
            c[i] = cc;
            cc = cc * (aa + 1u);
            aa = aa * aa;
        }
        hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_lcg_a), a, sizeof(a));
        if (e != hipSuccess)
            return e;
        e = hipMemcpyToSymbol(HIP_SYMBOL(c_lcg_c), c, sizeof(c));
        if (e != hipSuccess)
            return e;
        tables = true;
    }
    const u32 vps = (g.frames * g.channels + 7u) / 8u;
    const u64 total = (u64)vps * g.streams;
    if (total == 0)
        return hipSuccess;
    hipLaunchKernelGGL(k_generate, dim3((u32)((total + 255) / 256)), dim3(256), 0, st, g, mode, vps);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Node partial: one record for all streams of the batch.
//   node key = |peak| << 46 | (2^29-1 - min(frame,2^29-1)) << 17 | (65535 - stream%65536) << 1 | neg

// (the 17 sums and the 17 keys of a record go to two destinations: contiguous for the one-record
// form, [slot][17] arrays of a cmhip_node_t set for the RCCL exchange)
__device__ __forceinline__ u64 wave_sum(u64 v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ u64 wave_max(u64 v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const u64 w = __shfl_down(v, o, 64);
        v = w > v ? w : v;
    }
    return v;
}

// NC: the batch's channel count or an upper bound of it (1, 2, MAX_CH) -- the record has MAX_CH + 1 slots, a
// mono or stereo batch fills two or three of them, and the kernel runs beside the batch's next run, where
// every instruction it does not execute counts (config 5: 17 slots reduced for one channel took 30-300 us
// there instead of 9 alone).  One wave per workgroup: it takes whatever slot a CU has free.
template <u32 NC>
__global__ __launch_bounds__(64) void k_node_partial(const VuState *vu, u32 streams, u32 channels,
                                                     u32 parity, u64 first_global, u64 global_step,
                                                     long long *dst_sum, long long *dst_key)
{
    u64 sum[NC + 1], key[NC + 1];                // [NC]: frames / the peak over all channels
#pragma unroll
    for (u32 c = 0; c <= NC; c++) {
        sum[c] = 0;
        key[c] = 0;
    }
    for (u32 s = blockIdx.x * 64u + threadIdx.x; s < streams; s += gridDim.x * 64u) {
        const u64 gs = first_global + (u64)s * global_step;
        sum[NC] += NC == 1 ? vu[s].samples[parity] : vu[s].samples[parity] / channels;
#pragma unroll
        for (u32 c = 0; c < NC; c++) {
            if (c < channels) {
                sum[c] += vu[s].power[c];
                const u64 k0 = vu[s].key[c];
                if (k0) {
                    const u64 mag = k0 >> KEY_ABS_SHIFT;
                    const u64 idx = ~(k0 >> 1) & KEY_IDX_MASK;
                    u64 fr = NC == 1 ? idx : NC == 2 ? (channels == 2 ? idx >> 1 : idx) : idx / channels;
                    fr = fr > 0x1fffffffull ? 0x1fffffffull : fr;
                    const u64 nk = (mag << 46) | ((0x1fffffffull - fr) << 17) |
                                   ((65535ull - (gs & 65535ull)) << 1) | (k0 & 1ull);
                    key[c] = nk > key[c] ? nk : key[c];
                    key[NC] = nk > key[NC] ? nk : key[NC];
                }
            }
        }
    }
#pragma unroll
    for (u32 c = 0; c <= NC; c++) {
        const u64 ws = wave_sum(sum[c]);
        const u64 wk = wave_max(key[c]);
        const u32 slot = c == NC ? (u32)MAX_CH : c;
        if (threadIdx.x == 0 && (c == NC || c < channels)) {
            if (ws)
                atomicAdd(reinterpret_cast<u64 *>(dst_sum) + slot, ws);
            if (wk)
                atomicMax(reinterpret_cast<u64 *>(dst_key) + slot, wk);
        }
    }
}

hipError_t launch_node_partial(const VuState *vu, u32 streams, u32 channels, u32 parity,
                               uint64_t first_global, uint64_t global_step, long long *dst_sum,
                               long long *dst_key, bool clear, hipStream_t st, hipEvent_t ev_stop)
{
    if (clear) {                                 // (a cmhip_node_t clears a whole record set at once instead)
        hipError_t e = hipMemsetAsync(dst_sum, 0, sizeof(long long) * (MAX_CH + 1), st);
        if (e != hipSuccess)
            return e;
        e = hipMemsetAsync(dst_key, 0, sizeof(long long) * (MAX_CH + 1), st);
        if (e != hipSuccess)
            return e;
    }
    u32 grid = (streams + 63) / 64;
    if (grid > 1024)
        grid = 1024;
    if (channels == 1)
        hipExtLaunchKernelGGL(k_node_partial<1>, dim3(grid), dim3(64), 0, st, nullptr, ev_stop, 0, vu, streams, channels,
                              parity, first_global, global_step, dst_sum, dst_key);
    else if (channels == 2)
        hipExtLaunchKernelGGL(k_node_partial<2>, dim3(grid), dim3(64), 0, st, nullptr, ev_stop, 0, vu, streams, channels,
                              parity, first_global, global_step, dst_sum, dst_key);
    else
        hipExtLaunchKernelGGL(k_node_partial<MAX_CH>, dim3(grid), dim3(64), 0, st, nullptr, ev_stop, 0, vu, streams,
                              channels, parity, first_global, global_step, dst_sum, dst_key);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Snapshot of a set of VU windows: what the host needs of a window -- the samples accounted, C sums of
// squares, C packed peak keys; 1 + 2C words instead of the 33 of a VuState -- written by the kernel itself
// into pinned, device-mapped host memory, word-major ([word][stream]: a wave stores 512 contiguous bytes per
// instruction, whole lines over PCIe), and the device copy cleared for the set's next turn.  One launch on
// the copy stream replaces a 264 B-per-stream copy, a clear of the same size and two event records.
template <u32 NC>
__global__ __launch_bounds__(64) void k_vu_pack(VuState *vu, u32 streams, u32 channels, u32 parity, u64 *dst)
{
    const u32 s = blockIdx.x * 64u + threadIdx.x;
    if (s >= streams)
        return;
    VuState *v = vu + s;
    dst[s] = v->samples[parity];
#pragma unroll
    for (u32 c = 0; c < NC; c++) {
        if (c < channels) {
            dst[(u64)(1u + c) * streams + s] = v->power[c];
            dst[(u64)(1u + channels + c) * streams + s] = v->key[c];
            v->power[c] = 0;
            v->key[c] = 0;
        }
    }
    v->samples[0] = 0;
    v->samples[1] = 0;
}

hipError_t launch_vu_pack(VuState *vu, u32 streams, u32 channels, u32 parity, unsigned long long *dst_host_mapped,
                          hipStream_t st, hipEvent_t ev_stop)
{
    const u32 grid = (streams + 63u) / 64u;
    if (channels == 1)
        hipExtLaunchKernelGGL(k_vu_pack<1>, dim3(grid), dim3(64), 0, st, nullptr, ev_stop, 0, vu, streams, channels, parity,
                              dst_host_mapped);
    else if (channels == 2)
        hipExtLaunchKernelGGL(k_vu_pack<2>, dim3(grid), dim3(64), 0, st, nullptr, ev_stop, 0, vu, streams, channels, parity,
                              dst_host_mapped);
    else
        hipExtLaunchKernelGGL(k_vu_pack<MAX_CH>, dim3(grid), dim3(64), 0, st, nullptr, ev_stop, 0, vu, streams, channels,
                              parity, dst_host_mapped);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Plain HBM ceilings over the same buffers: read-only sum, and copy.
// Same access shape as k_run_fast -- one short-lived wave per 4 KiB tile, four
// non-temporal 16-byte accesses per lane -- with no arithmetic, so the numbers are what
// the memory system gives this pattern (the best of the shapes in tools/ubench_copy*.hip).

__global__ __launch_bounds__(64) void k_ceiling_read(const u32x4 *src, u64 nvec, u64 *sink)
{
    const u64 v0 = (u64)blockIdx.x * TILE_VEC + threadIdx.x;
    u32 acc = 0;
#pragma unroll
    for (u32 u = 0; u < TILE_U; u++) {
        const u64 i = v0 + 64u * u;
        if (i < nvec) {
            const u32x4 w = __builtin_nontemporal_load(src + i);
            acc += w.x ^ w.y ^ w.z ^ w.w;
        }
    }
    if (acc == 0x9e3779b9u)          // practically never: keeps the loads alive
        atomicAdd(sink, 1ull);
}

__global__ __launch_bounds__(64) void k_ceiling_copy(const u32x4 *src, u32x4 *dst, u64 nvec)
{
    const u64 v0 = (u64)blockIdx.x * TILE_VEC + threadIdx.x;
    u32x4 w[TILE_U];
#pragma unroll
    for (u32 u = 0; u < TILE_U; u++) {
        const u64 i = v0 + 64u * u;
        if (i < nvec)
            w[u] = __builtin_nontemporal_load(src + i);
    }
#pragma unroll
    for (u32 u = 0; u < TILE_U; u++) {
        const u64 i = v0 + 64u * u;
        if (i < nvec)
            __builtin_nontemporal_store(w[u], dst + i);
    }
}

hipError_t launch_ceiling(int mode, const void *src, void *dst, size_t bytes, u64 *sink,
                          hipStream_t st)
{
    const u64 nvec = bytes / 16;
    const u64 tiles = (nvec + TILE_VEC - 1) / TILE_VEC;
    if (tiles == 0 || tiles >= (1ull << 31))
        return hipErrorInvalidValue;
    if (mode == 0)
        hipLaunchKernelGGL(k_ceiling_read, dim3((u32)tiles), dim3(64), 0, st,
                           reinterpret_cast<const u32x4 *>(src), nvec, sink);
    else
        hipLaunchKernelGGL(k_ceiling_copy, dim3((u32)tiles), dim3(64), 0, st,
                           reinterpret_cast<const u32x4 *>(src), reinterpret_cast<u32x4 *>(dst),
                           nvec);
    return hipGetLastError();
}

}  // namespace cmhip
