// cmhip_kernels.hip -- gfx950 (MI355X, wave64) kernels of the transform -> vumeter path.
//
// Everything here is pointwise + reduction over packed int16, so the bound is HBM
// bandwidth (no MFMA).  Design rules followed (cdna_hip_programming.md G11-G13,
// Appendix B "Reduction"):
//   * 16 bytes per lane per load/store (global_load_dwordx4), 1 KiB per wave
//     instruction, four loads in flight per lane before the first use;
//   * a wave owns a contiguous chunk of ONE stream, so stream parameters live in
//     SGPRs and there is no barrier and no LDS traffic in the hot kernel;
//   * exact integer arithmetic only: 24-bit multiplies, one mul_hi for the division,
//     64-bit integer atomics for the VU window (order independent => bit exact).
//
// Reference semantics restated (never copied):
//   gain     ref: src/transform.c:101-124   q = trunc(x*g/scale) saturated
//   VU       ref: src/vumeter.c:161-177     first max-|x| peak, sum of squares
//   float    ref: src/enc_vorbis.c:108-115  x / 32768.f, planar
#include "cmhip_internal.h"
#include <stdlib.h>

namespace cmhip {

using u32 = uint32_t;
using u64 = unsigned long long;


__device__ __forceinline__ u32 uniform(u32 v) { return __builtin_amdgcn_readfirstlane(v); }

// magnitude of trunc(x*g/scale) after saturation; sgn = 0 or -1
__device__ __forceinline__ u32 gain_mag(int x, u32 g2, u32 magic, u32 shift, int &sgn)
{
    sgn = x >> 31;
    const u32 ax = (u32)((x ^ sgn) - sgn);          // |x| <= 32768
    const u32 n2 = __umul24(ax, g2);                // 2*|x|*gain < 2^32
    const u32 qa = __umulhi(n2, magic) >> shift;    // floor(|x|*gain/scale)
    const u32 lim = 32767u - (u32)sgn;              // 32767, or 32768 for negatives
    return qa < lim ? qa : lim;
}

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

__device__ __forceinline__ u64 make_key(u32 mag, u64 index, u32 neg)
{
    if (mag == 0)
        return 0;
    return ((u64)mag << KEY_ABS_SHIFT) | ((~index & KEY_IDX_MASK) << 1) | (u64)neg;
}

// ---------------------------------------------------------------------------
// Fast path: mono and stereo, any stereo channel map, slots 16-byte aligned.
//
// Every integer VALU op costs about the same on gfx950 (tools/ubench_valu.hip: mul_hi,
// mul_lo, 24-bit multiplies, packed-16 ops and three-operand ops all issue in ~4 cycles
// per wave, only two-operand 32-bit adds and fp32 multiplies are quicker), so the kernel
// is built to minimise the instruction count per sample: the two int16 halves of a dword
// are handled by packed-16 instructions wherever no 32-bit intermediate is needed
// (sign masks, magnitudes, saturation, sign restore, running maximum), and only the
// exact division (24-bit multiply, mul_hi, shift) is done per sample.
//
// Peak tracking costs ~1 op per sample: per 16-byte vector a packed running maximum of
// the magnitudes is folded into a per-lane key (magnitude, vector ordinal); which sample
// of the winning vector came first, and its sign, is found once per wave by looking at
// that one vector again (locate_peak).

typedef short v2s __attribute__((ext_vector_type(2)));
typedef unsigned short v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u32 pk_sign(u32 w)          // 0xffff in each negative half
{
    v2s v = __builtin_bit_cast(v2s, w);
    v = v >> (short)15;
    return __builtin_bit_cast(u32, v);
}
__device__ __forceinline__ u32 pk_sub(u32 a, u32 b)
{
    v2u x = __builtin_bit_cast(v2u, a) - __builtin_bit_cast(v2u, b);
    return __builtin_bit_cast(u32, x);
}
__device__ __forceinline__ u32 pk_min(u32 a, u32 b)
{
    v2u x = __builtin_elementwise_min(__builtin_bit_cast(v2u, a), __builtin_bit_cast(v2u, b));
    return __builtin_bit_cast(u32, x);
}
__device__ __forceinline__ u32 pk_max(u32 a, u32 b)
{
    v2u x = __builtin_elementwise_max(__builtin_bit_cast(v2u, a), __builtin_bit_cast(v2u, b));
    return __builtin_bit_cast(u32, x);
}

// one dword = two samples: returns the packed magnitudes after gain + saturation,
// `out` receives the packed signed result
__device__ __forceinline__ u32 gain2(u32 w, u32 g2lo, u32 g2hi, u32 magic, u32 shift, u32 &out)
{
    const u32 sg = pk_sign(w);
    const u32 aw = pk_sub(w ^ sg, sg);                       // |x| per half (u16, 32768 ok)
    const u32 n0 = __umul24(aw & 0xffffu, g2lo);             // 2*|x|*gain < 2^32
    const u32 n1 = __umul24(aw >> 16, g2hi);
    const u32 q0 = __umulhi(n0, magic) >> shift;             // floor(|x|*gain/scale)
    const u32 q1 = __umulhi(n1, magic) >> shift;
    u32 qw = __builtin_bit_cast(u32, __builtin_amdgcn_cvt_pk_u16(q0, q1));   // saturates at 65535
    qw = pk_min(qw, pk_sub(0x7fff7fffu, sg));                // 32767, or 32768 for negatives
    out = pk_sub(qw ^ sg, sg);
    return qw;
}

// sum of squares with as few 64-bit additions as exactness allows: three squares
// (each <= 2^30) fit a u32
struct PowAcc {
    u64 total;
    u32 part;
    u32 n;
    __device__ __forceinline__ void add(u32 mag)
    {
        part += mag * mag;
        if (++n == 3)
            flush();
    }
    // square of one 16-bit half of a packed pair added in a single v_mad_u32_u16
    __device__ __forceinline__ void add_lo(u32 pair)
    {
        asm("v_mad_u32_u16 %0, %1, %1, %0 op_sel:[0,0,0,0]" : "+v"(part) : "v"(pair));
        if (++n == 3)
            flush();
    }
    __device__ __forceinline__ void add_hi(u32 pair)
    {
        asm("v_mad_u32_u16 %0, %1, %1, %0 op_sel:[1,1,0,0]" : "+v"(part) : "v"(pair));
        if (++n == 3)
            flush();
    }
    __device__ __forceinline__ void flush()
    {
        total += part;
        part = 0;
        n = 0;
    }
};

template <int C>
__device__ __forceinline__ void store_f32(float *f32s, u64 plane, u32 v, const u32 (&o)[4])
{
    constexpr float k = 1.0f / 32768.0f;                 // exact scaling == x / 32768.f
    float f[8];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        f[2 * i] = (float)(int)(short)(o[i] & 0xffffu) * k;
        f[2 * i + 1] = (float)((int)o[i] >> 16) * k;
    }
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    if constexpr (C == 1) {
        // two 16-byte halves of one 32-byte run per lane: each instruction writes half of
        // every line, so these stay ordinary stores and L2 merges them (non-temporal ones
        // measured 20 % slower here; the stereo planes below write whole lines and gain 5 %)
        f32x4 *p = reinterpret_cast<f32x4 *>(f32s + (u64)v * 8);
        const f32x4 lo = {f[0], f[1], f[2], f[3]}, hi = {f[4], f[5], f[6], f[7]};
        p[0] = lo;
        p[1] = hi;
    } else {
        const f32x4 l = {f[0], f[2], f[4], f[6]}, r = {f[1], f[3], f[5], f[7]};
        __builtin_nontemporal_store(l, reinterpret_cast<f32x4 *>(f32s + (u64)v * 4));
        __builtin_nontemporal_store(r, reinterpret_cast<f32x4 *>(f32s + plane + (u64)v * 4));
    }
}

// scalar form of the same arithmetic, used for the samples of a ragged tail
__device__ __forceinline__ int gain1(int x, u32 g2, u32 magic, u32 shift, u32 &mag)
{
    int sg;
    mag = gain_mag(x, g2, magic, shift, sg);
    return (int)((mag ^ (u32)sg) - (u32)sg);
}

// value of another lane by DPP (0 where the source lane is outside the row)
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// wave64 reductions on the VALU (DPP), result valid in lane 63
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ u32 dpp0(u32 v)       // lanes without a source read 0
{
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true);
}
__device__ __forceinline__ u32 wave_max_u32(u32 v)
{
    v = max(v, dpp0<0x111, 0xf>(v));             // row_shr:1
    v = max(v, dpp0<0x112, 0xf>(v));             // row_shr:2
    v = max(v, dpp0<0x114, 0xf>(v));             // row_shr:4
    v = max(v, dpp0<0x118, 0xf>(v));             // row_shr:8  -> lane 15 of each row
    v = max(v, dpp0<0x142, 0xa>(v));             // row_bcast:15 into rows 1 and 3
    v = max(v, dpp0<0x143, 0xc>(v));             // row_bcast:31 into rows 2 and 3
    return (u32)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ u32 wave_add_u32(u32 v)
{
    v += dpp0<0x111, 0xf>(v);
    v += dpp0<0x112, 0xf>(v);
    v += dpp0<0x114, 0xf>(v);
    v += dpp0<0x118, 0xf>(v);
    v += dpp0<0x142, 0xa>(v);
    v += dpp0<0x143, 0xc>(v);
    return (u32)__builtin_amdgcn_readlane((int)v, 63);
}
// 64-bit sum of per-lane values below 2^40, as two 32-bit reductions
__device__ __forceinline__ u64 wave_add_u40(u64 v)
{
    const u32 lo = wave_add_u32((u32)v & 0xffffffu);          // 64 * 2^24 fits
    const u32 hi = wave_add_u32((u32)(v >> 24));              // 64 * 2^16 fits
    return (u64)lo + ((u64)hi << 24);
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr u32 TILE_U = 4;                        // 16-byte vectors per lane when PCM is written
constexpr u32 TILE_VEC = 64 * TILE_U;            // vectors per wave: 4 KiB of PCM
// read-only runs take bigger tiles to amortise the epilogue (picked per channel count from
// interleaved A/B runs, tools/ab_tiles.py); $CMHIP_VU_TILE (4, 8, 16) overrides for tuning
constexpr u32 TILE_U_VUONLY_MONO = 8;
constexpr u32 TILE_U_VUONLY_STEREO = 16;

// One wave = one 4 KiB tile of one stream, one pass: four non-temporal 16-byte loads per
// lane, arithmetic, four non-temporal stores, then a short epilogue.  Short-lived waves
// over small tiles keep the chip-wide access window compact; on MI355X that is worth
// ~15 % of HBM bandwidth over waves that each stream through tens of KiB
// (tools/ubench_copy*.hip: 6.3-6.5 TB/s against 5.0-5.4 TB/s for read+write).
// FULL: the tile lies completely inside the stream's whole vectors -- no bounds tests, no
// zero padding, no ragged tail; this is the case for all but the last tile of a stream.
template <int C, bool WRITE_PCM, bool WRITE_F32, bool DO_VU, int U, bool FULL>
__device__ __forceinline__ void fast_tile(const RunArgs &a, u32 s, u32 k, u32 nsamp, u32 nfull, u32 ntail,
                                          u64 base, VuState *vs)
{
    constexpr u32 TILE_U = U;
    constexpr u32 TILE_VEC = 64 * TILE_U;
    const u32 lane = threadIdx.x;
    const u32 v0 = k * TILE_VEC;
    (void)nsamp;

    const StreamParam *p = a.param + s;
    const u32 magic = p->magic, shift = p->shift, perm2 = p->perm2;
    const u32 g2lo = p->gain2[0], g2hi = p->gain2[C - 1];    // gains of the two dword halves

    const int16_t *ins = a.in + (u64)s * a.stride;
    const u32x4 *src = reinterpret_cast<const u32x4 *>(ins);
    int16_t *outs = WRITE_PCM ? a.out + (u64)s * a.stride : nullptr;
    u32x4 *dst = reinterpret_cast<u32x4 *>(outs);
    float *f32s = WRITE_F32 ? a.f32 + (u64)s * a.plane * C : nullptr;

    // ---- load: everything this lane will touch, before anything is stored (in-place safe)
    u32 x[TILE_U][4];
    bool full[TILE_U], tail[TILE_U];
#pragma unroll
    for (u32 u = 0; u < TILE_U; u++) {
        const u32 v = v0 + 64u * u + lane;
        full[u] = FULL || v < nfull;
        tail[u] = !FULL && ntail && v == nfull;
        u32x4 w = {0, 0, 0, 0};
        if (full[u])
            w = __builtin_nontemporal_load(src + v);
        x[u][0] = w.x; x[u][1] = w.y; x[u][2] = w.z; x[u][3] = w.w;
        if (tail[u]) {                           // ragged end: sample by sample, zero padded
            for (u32 j = 0; j < ntail; j++) {
                const u32 val = (u32)(uint16_t)ins[(u64)v * 8 + j];
#pragma unroll
                for (u32 i = 0; i < 4; i++)
                    if (i == (j >> 1))
                        x[u][i] |= val << (16u * (j & 1u));
            }
        }
    }

    // ---- arithmetic
    u32 qw[TILE_U][4];                           // packed magnitudes, kept for the epilogue
    PowAcc pw[2] = {{0, 0, 0}, {0, 0, 0}};
    u32 best[2] = {0, 0};                        // (magnitude << 16) | (U-1-u) << 6 | (63-lane)
#pragma unroll
    for (u32 u = 0; u < TILE_U; u++) {
        const u32 v = v0 + 64u * u + lane;
        u32 o[4], vmax = 0;
#pragma unroll
        for (u32 i = 0; i < 4; i++) {
            if constexpr (C == 2)
                x[u][i] = __builtin_amdgcn_perm(x[u][i], x[u][i], perm2);   // stereo channel map
            qw[u][i] = gain2(x[u][i], g2lo, g2hi, magic, shift, o[i]);
            if constexpr (DO_VU) {
                vmax = pk_max(vmax, qw[u][i]);
                pw[0].add_lo(qw[u][i]);
                pw[C - 1].add_hi(qw[u][i]);
            }
        }
        if constexpr (DO_VU) {
            const u32 tag = ((TILE_U - 1u - u) << 6) | (63u - lane);
            const u32 k0 = (vmax << 16) | tag;
            const u32 k1 = (vmax & 0xffff0000u) | tag;
            best[0] = max(best[0], k0);
            best[1] = max(best[1], k1);
        }
        if (full[u]) {
            if constexpr (WRITE_PCM) {
                const u32x4 ov = {o[0], o[1], o[2], o[3]};
                __builtin_nontemporal_store(ov, dst + v);
            }
            if constexpr (WRITE_F32)
                store_f32<C>(f32s, a.plane, v, o);
        } else if (tail[u]) {
            for (u32 j = 0; j < ntail; j++) {
                u32 ow = 0;
#pragma unroll
                for (u32 i = 0; i < 4; i++)
                    if (i == (j >> 1))
                        ow = o[i];
                const int q = (int)(short)((ow >> (16u * (j & 1u))) & 0xffffu);
                if constexpr (WRITE_PCM)
                    outs[(u64)v * 8 + j] = (int16_t)q;
                if constexpr (WRITE_F32)
                    f32s[(u64)(j % (u32)C) * a.plane + ((u64)v * 8 + j) / (u32)C] = q * (1.0f / 32768.0f);
            }
        }
    }

    // ---- epilogue: one add and one max per channel into the stream's window
    if constexpr (DO_VU) {
        pw[0].flush();
        pw[1].flush();
        u64 sum[2];
        u32 wkey[2];
        if constexpr (C == 1) {
            sum[0] = wave_add_u40(pw[0].total + pw[1].total);
            wkey[0] = wave_max_u32(max(best[0], best[1]));
            sum[1] = 0;
            wkey[1] = 0;
        } else {
            sum[0] = wave_add_u40(pw[0].total);
            sum[1] = wave_add_u40(pw[1].total);
            wkey[0] = wave_max_u32(best[0]);
            wkey[1] = wave_max_u32(best[1]);
        }
        u64 gkey[2] = {0, 0};
#pragma unroll
        for (int c = 0; c < C; c++) {
            const u32 mag = wkey[c] >> 16;
            if (mag == 0)
                continue;
            // the winning vector: lowest ordinal, then lowest lane; fetch it into SGPRs
            const u32 uw = TILE_U - 1u - ((wkey[c] >> 6) & (TILE_U - 1u));
            const u32 lw = 63u - (wkey[c] & 63u);
            u32 Q[4], X[4];
#pragma unroll
            for (u32 u = 0; u < TILE_U; u++) {
                if (uw == u) {
#pragma unroll
                    for (u32 i = 0; i < 4; i++) {
                        Q[i] = (u32)__builtin_amdgcn_readlane((int)qw[u][i], (int)lw);
                        X[i] = (u32)__builtin_amdgcn_readlane((int)x[u][i], (int)lw);
                    }
                }
            }
            // first sample of this channel with that magnitude, and its sign
            u32 first = 8, neg = 0;
#pragma unroll
            for (u32 j = 0; j < 8; j++) {
                if (C == 2 && (j & 1u) != (u32)c)
                    continue;
                const u32 m = (Q[j >> 1] >> (16u * (j & 1u))) & 0xffffu;
                if (m == mag && first == 8) {
                    first = j;
                    neg = (X[j >> 1] >> (16u * (j & 1u) + 15u)) & 1u;
                }
            }
            gkey[c] = make_key(mag, base + 8ull * (v0 + 64u * uw + lw) + first, neg);
        }
        if (lane < (u32)C) {
            const u64 ssum = lane == 0 ? sum[0] : sum[1];
            const u64 skey = lane == 0 ? gkey[0] : gkey[1];
            if (ssum)
                atomicAdd(&vs->power[lane], ssum);
            if (skey)
                atomicMax(&vs->key[lane], skey);
        }
    }
}

template <int C, bool WRITE_PCM, bool WRITE_F32, bool DO_VU, int U>
__global__ __launch_bounds__(64) void k_run_fast(RunArgs a)
{
    constexpr u32 TILE_VEC = 64 * U;
    const u32 lane = threadIdx.x;
    const u32 s = blockIdx.x / a.chunks;         // stream
    const u32 k = blockIdx.x - s * a.chunks;     // tile inside the stream

    const u32 nfr = a.nframes ? a.nframes[s] : a.frames;
    const u32 nsamp = nfr * (u32)C;
    const u32 nfull = nsamp >> 3;                // whole 16-byte vectors
    const u32 ntail = nsamp & 7u;                // samples in the partial last vector
    const u32 v0 = k * TILE_VEC;

    VuState *vs = DO_VU ? a.vu + s : nullptr;
    u64 base = 0;
    if constexpr (DO_VU) {
        // window position: read from one slot, the stream's first tile writes the other
        base = vs->samples[a.parity];
        if (k == 0 && lane == 0)
            vs->samples[a.parity ^ 1u] = base + nsamp;
    }
    if (v0 >= nfull + (ntail ? 1u : 0u))
        return;
    if (v0 + TILE_VEC <= nfull)
        fast_tile<C, WRITE_PCM, WRITE_F32, DO_VU, U, true>(a, s, k, nsamp, nfull, ntail, base, vs);
    else
        fast_tile<C, WRITE_PCM, WRITE_F32, DO_VU, U, false>(a, s, k, nsamp, nfull, ntail, base, vs);
}

// ---------------------------------------------------------------------------
// Wide path: 4 or 8 channels with the identity channel map (the template also covers 16,
// which k_run_rows now serves faster).  Same tile scheme and the
// same packed arithmetic as k_run_fast; a 16-byte vector holds 8/C frames, so every vector
// position has a fixed channel (for 16 channels: fixed per lane parity) and the per-channel
// accumulators live in registers.  NS = min(C, 8) accumulator slots per lane: the half h of
// dword i feeds slot 2*(i % (NS/2)) + h.

template <int C, bool WRITE_PCM, bool WRITE_F32, bool DO_VU, int U>
__global__ __launch_bounds__(64) void k_run_wide(RunArgs a)
{
    constexpr u32 TILE_U = U;
    constexpr u32 TILE_VEC = 64 * TILE_U;
    constexpr u32 NS = C < 8 ? C : 8;            // accumulator slots per lane
    constexpr u32 NG = NS / 2;                   // dword groups
    constexpr u32 NCLS = C == 16 ? 2 : 1;        // lane classes (vector parity) for 16 channels
    const u32 lane = threadIdx.x;
    const u32 s = blockIdx.x / a.chunks;
    const u32 k = blockIdx.x - s * a.chunks;

    const u32 nfr = a.nframes ? a.nframes[s] : a.frames;
    const u32 nsamp = nfr * (u32)C;
    const u32 nfull = nsamp >> 3;
    const u32 ntail = nsamp & 7u;                // only possible for 4 channels (one frame)
    const u32 v0 = k * TILE_VEC;

    VuState *vs = DO_VU ? a.vu + s : nullptr;
    u64 base = 0;
    if constexpr (DO_VU) {
        base = vs->samples[a.parity];
        if (k == 0 && lane == 0)
            vs->samples[a.parity ^ 1u] = base + nsamp;
    }
    if (v0 >= nfull + (ntail ? 1u : 0u))
        return;

    const StreamParam *p = a.param + s;
    const u32 magic = p->magic, shift = p->shift;
    const u32 cls = C == 16 ? (lane & 1u) : 0u;  // v0 and 64*u are even: vector parity = lane parity
    u32 g2[NS];
#pragma unroll
    for (u32 i = 0; i < NS; i++)
        g2[i] = p->gain2[i + 8u * cls];

    const int16_t *ins = a.in + (u64)s * a.stride;
    const u32x4 *src = reinterpret_cast<const u32x4 *>(ins);
    int16_t *outs = WRITE_PCM ? a.out + (u64)s * a.stride : nullptr;
    u32x4 *dst = reinterpret_cast<u32x4 *>(outs);
    float *f32s = WRITE_F32 ? a.f32 + (u64)s * a.plane * C : nullptr;

    u32 x[TILE_U][4];
    bool full[TILE_U], tail[TILE_U];
#pragma unroll
    for (u32 u = 0; u < TILE_U; u++) {
        const u32 v = v0 + 64u * u + lane;
        full[u] = v < nfull;
        tail[u] = ntail && v == nfull;
        u32x4 w = {0, 0, 0, 0};
        if (full[u])
            w = __builtin_nontemporal_load(src + v);
        x[u][0] = w.x; x[u][1] = w.y; x[u][2] = w.z; x[u][3] = w.w;
        if (tail[u]) {
            for (u32 j = 0; j < ntail; j++) {
                const u32 val = (u32)(uint16_t)ins[(u64)v * 8 + j];
#pragma unroll
                for (u32 i = 0; i < 4; i++)
                    if (i == (j >> 1))
                        x[u][i] |= val << (16u * (j & 1u));
            }
        }
    }

    u32 qw[TILE_U][4];
    PowAcc pw[NS];
    u32 best[NS];
#pragma unroll
    for (u32 i = 0; i < NS; i++) {
        pw[i] = PowAcc{0, 0, 0};
        best[i] = 0;
    }
#pragma unroll
    for (u32 u = 0; u < TILE_U; u++) {
        const u32 v = v0 + 64u * u + lane;
        u32 o[4], vmax[NG];
#pragma unroll
        for (u32 g = 0; g < NG; g++)
            vmax[g] = 0;
#pragma unroll
        for (u32 i = 0; i < 4; i++) {
            constexpr u32 dummy = 0;
            (void)dummy;
            const u32 g = i % NG;
            qw[u][i] = gain2(x[u][i], g2[2 * g], g2[2 * g + 1], magic, shift, o[i]);
            if constexpr (DO_VU) {
                vmax[g] = pk_max(vmax[g], qw[u][i]);
                pw[2 * g].add_lo(qw[u][i]);
                pw[2 * g + 1].add_hi(qw[u][i]);
            }
        }
        if constexpr (DO_VU) {
            const u32 tag = ((TILE_U - 1u - u) << 6) | (63u - lane);
#pragma unroll
            for (u32 g = 0; g < NG; g++) {
                best[2 * g] = max(best[2 * g], (vmax[g] << 16) | tag);
                best[2 * g + 1] = max(best[2 * g + 1], (vmax[g] & 0xffff0000u) | tag);
            }
        }
        const u32 cnt = full[u] ? 8u : (tail[u] ? ntail : 0u);
        if (full[u]) {
            if constexpr (WRITE_PCM) {
                const u32x4 ov = {o[0], o[1], o[2], o[3]};
                __builtin_nontemporal_store(ov, dst + v);
            }
        } else if (tail[u]) {
            if constexpr (WRITE_PCM) {
                for (u32 j = 0; j < ntail; j++) {
                    u32 ow = 0;
#pragma unroll
                    for (u32 i = 0; i < 4; i++)
                        if (i == (j >> 1))
                            ow = o[i];
                    outs[(u64)v * 8 + j] = (int16_t)((ow >> (16u * (j & 1u))) & 0xffffu);
                }
            }
        }
        if constexpr (WRITE_F32) {
            // planar float: consecutive lanes hold consecutive frames, so each of these
            // stores writes a contiguous run of a plane
#pragma unroll
            for (u32 j = 0; j < 8; j++) {
                if (j < cnt) {
                    const u32 idx = v * 8u + j;
                    const int q = (int)(short)((o[j >> 1] >> (16u * (j & 1u))) & 0xffffu);
                    f32s[(u64)(idx % (u32)C) * a.plane + idx / (u32)C] = q * (1.0f / 32768.0f);
                }
            }
        }
    }

    if constexpr (DO_VU) {
#pragma unroll
        for (u32 i = 0; i < NS; i++)
            pw[i].flush();
#pragma unroll
        for (u32 c = 0; c < NCLS; c++) {
            const bool mine = NCLS == 1 || cls == c;
#pragma unroll
            for (u32 sl = 0; sl < NS; sl++) {
                const u64 sum = wave_add_u40(mine ? pw[sl].total : 0ull);
                const u32 wkey = wave_max_u32(mine ? best[sl] : 0u);
                const u32 ch = sl + 8u * c;
                const u32 mag = wkey >> 16;
                u64 gkey = 0;
                if (mag) {
                    const u32 uw = TILE_U - 1u - ((wkey >> 6) & (TILE_U - 1u));
                    const u32 lw = 63u - (wkey & 63u);
                    u32 Q[4], X[4];
#pragma unroll
                    for (u32 u = 0; u < TILE_U; u++) {
                        if (uw == u) {
#pragma unroll
                            for (u32 i = 0; i < 4; i++) {
                                Q[i] = (u32)__builtin_amdgcn_readlane((int)qw[u][i], (int)lw);
                                X[i] = (u32)__builtin_amdgcn_readlane((int)x[u][i], (int)lw);
                            }
                        }
                    }
                    u32 first = 8, neg = 0;
#pragma unroll
                    for (u32 j = 0; j < 8; j++) {
                        if (j % NS != sl)
                            continue;
                        const u32 m = (Q[j >> 1] >> (16u * (j & 1u))) & 0xffffu;
                        if (m == mag && first == 8) {
                            first = j;
                            neg = (X[j >> 1] >> (16u * (j & 1u) + 15u)) & 1u;
                        }
                    }
                    gkey = make_key(mag, base + 8ull * (v0 + 64u * uw + lw) + first, neg);
                }
                if (lane == 0) {
                    if (sum)
                        atomicAdd(&vs->power[ch], sum);
                    if (gkey)
                        atomicMax(&vs->key[ch], gkey);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Row path: any channel count, identity channel map (3, 5, 6 = 5.1, 7, 9...16 channels; 1, 2,
// 4 and 8 have the kernels above).  The channel of a sample is (8*v + j) mod C for
// vector v and position j, which changes from vector to vector -- unless the vectors a lane
// visits are a multiple of P = C / gcd(C, 8) apart.  So a wave walks ROWS of W = 64 - 64 % P
// vectors (63, 60, 55 or 52 of the 64 lanes work): rows are contiguous, loads and stores stay
// coalesced, and position j of a lane has the same channel in every row.  Gains and
// per-position accumulators are then per-lane constants / registers, exactly the packed
// arithmetic of the kernels above; only the final merge differs (positions of different
// lanes hold different channels: LDS atomics by channel, once per wave).

//
// MAP: the streams carry channel maps.  A row holds whole frames, so a mapped sample's source
// lies in the same row: the raw row goes through LDS and every position gathers its source
// with a 16-bit read at a per-lane constant offset.

template <bool WRITE_PCM, bool WRITE_F32, bool DO_VU, bool MAP>
__global__ __launch_bounds__(64) void k_run_rows(RunArgs a, u32 W, u32 rows_per_tile)
{
    constexpr u32 UR = 4;                        // rows in flight
    __shared__ u64 lsum[MAX_CH];
    __shared__ u64 lkey[MAX_CH];
    __shared__ u32x4 raw[MAP ? UR * 64 : 1];     // the rows as loaded (MAP only)
    const u32 lane = threadIdx.x;
    const u32 s = blockIdx.x / a.chunks;
    const u32 k = blockIdx.x - s * a.chunks;
    const u32 C = a.channels;

    const u32 nfr = a.nframes ? a.nframes[s] : a.frames;
    const u32 nsamp = nfr * C;
    const u32 nfull = nsamp >> 3;                // whole 16-byte vectors
    const u32 ntail = nsamp & 7u;                // samples in the partial last vector
    const u32 nvec = nfull + (ntail ? 1u : 0u);
    const u32 row0 = k * rows_per_tile;

    VuState *vs = DO_VU ? a.vu + s : nullptr;
    u64 base = 0;
    if constexpr (DO_VU) {
        base = vs->samples[a.parity];
        if (k == 0 && lane == 0)
            vs->samples[a.parity ^ 1u] = base + nsamp;
    }
    if ((u64)row0 * W >= nvec)
        return;
    if constexpr (DO_VU) {
        if (lane < MAX_CH) {
            lsum[lane] = 0;
            lkey[lane] = 0;
        }
    }

    const StreamParam *p = a.param + s;
    const u32 magic = p->magic, shift = p->shift;
    const bool active = lane < W;
    const u32 lane_fr = 8u * lane / C;           // whole frames before this lane's vector in a row
    const u32 phase = 8u * lane - lane_fr * C;   // channel of its position 0
    const u32 FW = 8u * W / C;                   // frames per row (8W is a multiple of C)
    u32 ch[8], df[8], g2[8];
    u32 so[8];                                   // MAP: byte offset of position j's source in its row
#pragma unroll
    for (u32 j = 0; j < 8; j++) {
        const u32 t = phase + j;
        df[j] = t / C;
        ch[j] = t - df[j] * C;
        g2[j] = p->gain2[ch[j]];
        so[j] = MAP ? 2u * ((lane_fr + df[j]) * C + p->chmap[ch[j]]) : 0u;
    }

    const int16_t *ins = a.in + (u64)s * a.stride;
    const u32x4 *src = reinterpret_cast<const u32x4 *>(ins);
    int16_t *outs = WRITE_PCM ? a.out + (u64)s * a.stride : nullptr;
    u32x4 *dst = reinterpret_cast<u32x4 *>(outs);
    float *f32s = WRITE_F32 ? a.f32 + (u64)s * a.plane * C : nullptr;

    PowAcc pw[8];
    u32 best[8];                                 // |peak| << 16 | (0x7fff - row in tile) << 1 | negative
#pragma unroll
    for (u32 j = 0; j < 8; j++) {
        pw[j] = PowAcc{0, 0, 0};
        best[j] = 0;
    }

    for (u32 r0 = 0; r0 < rows_per_tile; r0 += UR) {
        if ((u64)(row0 + r0) * W >= nvec)
            break;
        u32 x[UR][4];
        bool full[UR], tail[UR];
#pragma unroll
        for (u32 u = 0; u < UR; u++) {
            const u32 v = (row0 + r0 + u) * W + lane;
            full[u] = active && v < nfull;
            tail[u] = active && ntail && v == nfull;
            u32x4 w = {0, 0, 0, 0};
            if (full[u])
                w = __builtin_nontemporal_load(src + v);
            x[u][0] = w.x; x[u][1] = w.y; x[u][2] = w.z; x[u][3] = w.w;
            if (tail[u]) {
                for (u32 j = 0; j < ntail; j++) {
                    const u32 val = (u32)(uint16_t)ins[(u64)v * 8 + j];
#pragma unroll
                    for (u32 i = 0; i < 4; i++)
                        if (i == (j >> 1))
                            x[u][i] |= val << (16u * (j & 1u));
                }
            }
        }
        if constexpr (MAP) {
            __syncthreads();                     // the previous rows have been gathered
#pragma unroll
            for (u32 u = 0; u < UR; u++)
                raw[u * 64u + lane] = u32x4{x[u][0], x[u][1], x[u][2], x[u][3]};
            __syncthreads();
            if (active) {
#pragma unroll
                for (u32 u = 0; u < UR; u++) {
                    const unsigned char *rowb = reinterpret_cast<const unsigned char *>(raw + u * 64u);
#pragma unroll
                    for (u32 i = 0; i < 4; i++) {
                        const u32 lo = *reinterpret_cast<const uint16_t *>(rowb + so[2 * i]);
                        const u32 hi = *reinterpret_cast<const uint16_t *>(rowb + so[2 * i + 1]);
                        x[u][i] = lo | (hi << 16);
                    }
                }
            }
        }
#pragma unroll
        for (u32 u = 0; u < UR; u++) {
            const u32 row = row0 + r0 + u;
            const u32 v = row * W + lane;
            u32 o[4];
            const u32 tag = (0x7fffu - (r0 + u)) << 1;
#pragma unroll
            for (u32 i = 0; i < 4; i++) {
                const u32 qw = gain2(x[u][i], g2[2 * i], g2[2 * i + 1], magic, shift, o[i]);
                if constexpr (DO_VU) {
                    best[2 * i] = max(best[2 * i], (qw << 16) | tag | ((x[u][i] >> 15) & 1u));
                    best[2 * i + 1] = max(best[2 * i + 1], (qw & 0xffff0000u) | tag | (x[u][i] >> 31));
                    pw[2 * i].add_lo(qw);
                    pw[2 * i + 1].add_hi(qw);
                }
            }
            if (full[u]) {
                if constexpr (WRITE_PCM) {
                    const u32x4 ov = {o[0], o[1], o[2], o[3]};
                    __builtin_nontemporal_store(ov, dst + v);
                }
            } else if (tail[u]) {
                if constexpr (WRITE_PCM) {
                    for (u32 j = 0; j < ntail; j++) {
                        u32 ow = 0;
#pragma unroll
                        for (u32 i = 0; i < 4; i++)
                            if (i == (j >> 1))
                                ow = o[i];
                        outs[(u64)v * 8 + j] = (int16_t)((ow >> (16u * (j & 1u))) & 0xffffu);
                    }
                }
            }
            if constexpr (WRITE_F32) {
                const u32 cnt = full[u] ? 8u : (tail[u] ? ntail : 0u);
                const u32 fr = row * FW + lane_fr;
#pragma unroll
                for (u32 j = 0; j < 8; j++) {
                    if (j < cnt) {
                        const int q = (int)(short)((o[j >> 1] >> (16u * (j & 1u))) & 0xffffu);
                        f32s[(u64)ch[j] * a.plane + fr + df[j]] = q * (1.0f / 32768.0f);
                    }
                }
            }
        }
    }

    if constexpr (DO_VU) {
        __syncthreads();                         // accumulators cleared (one wave: cheap)
#pragma unroll
        for (u32 j = 0; j < 8; j++) {
            pw[j].flush();
            if (pw[j].total)
                atomicAdd(reinterpret_cast<unsigned long long *>(&lsum[ch[j]]), (unsigned long long)pw[j].total);
            const u32 mag = best[j] >> 16;
            if (mag) {
                const u32 rr = 0x7fffu - ((best[j] >> 1) & 0x7fffu);
                const u64 v = (u64)(row0 + rr) * W + lane;
                const u64 key = make_key(mag, base + 8ull * v + j, best[j] & 1u);
                atomicMax(reinterpret_cast<unsigned long long *>(&lkey[ch[j]]), (unsigned long long)key);
            }
        }
        __syncthreads();
        if (lane < C) {
            if (lsum[lane])
                atomicAdd(&vs->power[lane], lsum[lane]);
            if (lkey[lane])
                atomicMax(&vs->key[lane], lkey[lane]);
        }
    }
}

// ---------------------------------------------------------------------------
// Launcher of the block kernels: by channel count and by what the batch asks for.

hipError_t launch_run(const RunArgs &a, hipStream_t st)
{
    const bool pcm = a.out != nullptr, f32 = a.f32 != nullptr, vu = a.vu != nullptr;
    if (a.streams == 0 || a.frames == 0)
        return hipSuccess;
    if (a.channels <= 2) {
        // one 64-thread block per tile: 4 KiB when PCM or float is written, larger read-only
        RunArgs b = a;
        u32 tile_u = TILE_U;
        if (!pcm && !f32) {
            tile_u = a.channels == 1 ? TILE_U_VUONLY_MONO : TILE_U_VUONLY_STEREO;
            const char *e = getenv("CMHIP_VU_TILE");
            if (e) {
                const int v = atoi(e);
                if (v == 4 || v == 8 || v == 16)
                    tile_u = (u32)v;
            }
        }
        const u64 nvec = ((u64)a.frames * a.channels + 7) / 8;
        b.chunks = (u32)((nvec + 64ull * tile_u - 1) / (64ull * tile_u));
        if (b.chunks == 0)
            b.chunks = 1;
        if ((u64)b.chunks * a.streams >= (1ull << 31))
            return hipErrorInvalidValue;
        const u32 grid = a.streams * b.chunks;
#define CMHIP_FAST(C, P, F, V, U)                                                  \
    hipLaunchKernelGGL((k_run_fast<C, P, F, V, U>), dim3(grid), dim3(64), 0, st, b)
#define CMHIP_FAST_C(C)                                                            \
    do {                                                                           \
        if (pcm && !f32 && vu) CMHIP_FAST(C, true, false, true, 4);                \
        else if (!pcm && !f32 && vu && tile_u == 4) CMHIP_FAST(C, false, false, true, 4);   \
        else if (!pcm && !f32 && vu && tile_u == 8) CMHIP_FAST(C, false, false, true, 8);   \
        else if (!pcm && !f32 && vu) CMHIP_FAST(C, false, false, true, 16);        \
        else if (pcm && !f32 && !vu) CMHIP_FAST(C, true, false, false, 4);         \
        else if (pcm && f32 && vu) CMHIP_FAST(C, true, true, true, 4);             \
        else if (!pcm && f32 && vu) CMHIP_FAST(C, false, true, true, 4);           \
        else if (pcm && f32 && !vu) CMHIP_FAST(C, true, true, false, 4);           \
        else if (!pcm && f32 && !vu) CMHIP_FAST(C, false, true, false, 4);         \
    } while (0)
        if (a.channels == 1)
            CMHIP_FAST_C(1);
        else
            CMHIP_FAST_C(2);
#undef CMHIP_FAST_C
#undef CMHIP_FAST
    } else if ((a.channels == 4 || a.channels == 8) && a.identity_maps) {
        RunArgs b = a;
        // tile size: read-only runs take 16 KiB tiles, the rest 8 KiB (tools/bench_generic.py);
        // 16 channels run faster on k_run_rows below (5.6 against 4.7 TB/s)
        const u32 wu = (!pcm && !f32) ? 16u : 8u;
        const u64 nvec = ((u64)a.frames * a.channels + 7) / 8;
        b.chunks = (u32)((nvec + 64ull * wu - 1) / (64ull * wu));
        if (b.chunks == 0)
            b.chunks = 1;
        if ((u64)b.chunks * a.streams >= (1ull << 31))
            return hipErrorInvalidValue;
        const u32 grid = a.streams * b.chunks;
#define CMHIP_WIDE(C, P, F, V)                                                     \
    do {                                                                           \
        if (wu == 16u)                                                             \
            hipLaunchKernelGGL((k_run_wide<C, P, F, V, 16>), dim3(grid), dim3(64), 0, st, b); \
        else                                                                       \
            hipLaunchKernelGGL((k_run_wide<C, P, F, V, 8>), dim3(grid), dim3(64), 0, st, b);  \
    } while (0)
#define CMHIP_WIDE_C(C)                                                            \
    do {                                                                           \
        if (pcm && !f32 && vu) CMHIP_WIDE(C, true, false, true);                   \
        else if (!pcm && !f32 && vu) CMHIP_WIDE(C, false, false, true);            \
        else if (pcm && !f32 && !vu) CMHIP_WIDE(C, true, false, false);            \
        else if (pcm && f32 && vu) CMHIP_WIDE(C, true, true, true);                \
        else if (!pcm && f32 && vu) CMHIP_WIDE(C, false, true, true);              \
        else if (pcm && f32 && !vu) CMHIP_WIDE(C, true, true, false);              \
        else if (!pcm && f32 && !vu) CMHIP_WIDE(C, false, true, false);            \
    } while (0)
        if (a.channels == 4)
            CMHIP_WIDE_C(4);
        else
            CMHIP_WIDE_C(8);
#undef CMHIP_WIDE_C
#undef CMHIP_WIDE
    } else {
        // any other channel count, or channel maps on more than two channels: rows of W vectors
        // so that every lane position keeps its channel
        RunArgs b = a;
        u32 g = a.channels, e = 8;
        while (e) {                              // gcd(C, 8)
            const u32 t = g % e;
            g = e;
            e = t;
        }
        const u32 P = a.channels / g;
        const u32 W = 64u - 64u % P;
        // rows per tile (~1 KiB each); with 4 or 8 channels (here only when they carry channel
        // maps) every lane adds to the same few LDS words at the end: bigger tiles, fewer merges
        const u32 rpt = P == 1 ? 32u : (!pcm && !f32) ? 16u : 8u;
        const u64 nvec = ((u64)a.frames * a.channels + 7) / 8;
        const u64 rows = (nvec + W - 1) / W;
        b.chunks = (u32)((rows + rpt - 1) / rpt);
        if (b.chunks == 0)
            b.chunks = 1;
        if ((u64)b.chunks * a.streams >= (1ull << 31))
            return hipErrorInvalidValue;
        const u32 grid = a.streams * b.chunks;
#define CMHIP_ROWS(P_, F_, V_)                                                                      \
    do {                                                                                            \
        if (a.identity_maps)                                                                        \
            hipLaunchKernelGGL((k_run_rows<P_, F_, V_, false>), dim3(grid), dim3(64), 0, st, b, W, rpt); \
        else                                                                                        \
            hipLaunchKernelGGL((k_run_rows<P_, F_, V_, true>), dim3(grid), dim3(64), 0, st, b, W, rpt);  \
    } while (0)
        if (pcm && !f32 && vu) CMHIP_ROWS(true, false, true);
        else if (!pcm && !f32 && vu) CMHIP_ROWS(false, false, true);
        else if (pcm && !f32 && !vu) CMHIP_ROWS(true, false, false);
        else if (pcm && f32 && vu) CMHIP_ROWS(true, true, true);
        else if (!pcm && f32 && vu) CMHIP_ROWS(false, true, true);
        else if (pcm && f32 && !vu) CMHIP_ROWS(true, true, false);
        else if (!pcm && f32 && !vu) CMHIP_ROWS(false, true, false);
#undef CMHIP_ROWS
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// EQ path: int16 -> channel map -> gain -> x/32768.f -> NSEC biquads (Direct Form I with the
// fmaf order the oracle fixes) -> float planes and/or int16 (+VU of the int16 result).

// float -> int16 of the EQ result: round to nearest even, saturate, NaN -> 0 (oracle_f32_to_i16)
__device__ __forceinline__ int f32_to_i16(float y)
{
    float v = y * 32768.0f;
    if (v != v)
        return 0;
    v = __builtin_rintf(v);                           // v_rndne_f32: nearest even
    v = v >= 32767.0f ? 32767.0f : v;
    v = v <= -32768.0f ? -32768.0f : v;
    return (int)v;
}

// ---------------------------------------------------------------------------
// The pipelined EQ kernel (1..4 sections; config 3 is its mono, float-planes case).
//
// The recurrence allows no parallelism along time, so the per-sample work is cut in two:
//
//   feed-forward  f[t] = fma(b2, x[t-2], fma(b1, x[t-1], b0*x[t]))      no loop-carried dependence
//   recurrence    y[t] = fma(-a1, y[t-1], fma(-a2, y[t-2], f[t]))       two dependent FMAs
//
// which is the oracle's Direct Form I in the same operation order.  A workgroup owns G
// rows (a row = one channel of one stream) and walks them in 64-frame blocks, one
// __syncthreads() per block, with waves of four roles that hand 64-frame rows to each other
// through double-buffered LDS tiles:
//
//   T-in waves (lane = row x 8-frame chunk; time-parallel, G/8 of them):
//     global load (two blocks ahead) -> gain -> float -> feed-forward of section 0 -> F_0
//   T-ff waves (same lane shape, G/2 rows each in passes of 8 rows):
//     Y_k-1 -> feed-forward of section k -> F_k               (k = 1 .. NSEC-1)
//     x[t-1], x[t-2] of a chunk come from the neighbouring lane by DPP (row_shr:1), those
//     of a block's first chunk from the last chunk of the previous step (row_shl:7).
//   R waves (lane = section x row; 64/G sections side by side, sequential in time):
//     F_k (loaded into registers a step ahead) -> the two dependent FMAs per sample -> Y_k
//   S work: Y_last -> coalesced non-temporal global stores (float planes), and for an int16
//     result / VU window the conversion and the window of it; dealt out over R and T-ff waves
//     (float planes only) or done by S waves of their own -- see eq_role() below.
//
// Measured on MI355X (tools/ubench_chain.hip, ubench_lds*.hip): a wave alone issues one
// VALU op per ~4.3 clk, a dependent one after ~8; ds_read_b128 costs a wave ~5-10 clk to
// issue, ds_write_b128 ~24 (50 when four waves write at once).  So the only waves that are
// long per step by themselves are the R waves (16 reads, 128 FMAs, 16 writes); everything
// without a recurrence is spread over T lanes, where a 64-frame row costs 2 reads + 2 writes.
// Rows are 68 floats (16-byte aligned, lane-per-row b128 access without bank conflicts).

#ifndef CMHIP_EQ_RLAG
#define CMHIP_EQ_RLAG 1           // R waves load the next row into registers a step ahead (0: same step)
#endif
#ifndef CMHIP_EQ_RSLOTS
#define CMHIP_EQ_RSLOTS 1         // row slots (of 8) whose stores each R wave takes when the output is float planes only
#endif
#ifndef CMHIP_EQ_ABL
#define CMHIP_EQ_ABL 0            // `make abl`: timing-only builds with one part of the pipeline cut out
#endif

// Channels: a "row" is one channel of one stream -- every channel runs its stream's filter
// with state of its own -- and a workgroup takes G / C whole streams.  CH = 1: mono, the 16-byte
// vector loads and packed stores of config 3; CH = 2: stereo, two vector loads per chunk and one
// v_perm_b32 per sample pair pick the row's channel (through the stream's channel map); CH = 0:
// any count, the T lanes gather their channel's samples with 16-bit loads.  For CH != 1 the S
// lanes scatter the int16 result into the interleaved frames.
//
// Wave order.  Waves w, w+4 and w+8 of a workgroup share a SIMD, and a step lasts as long as
// the most loaded SIMD needs (issue slots plus the time its waves are blocked on LDS writes):
// per step an R wave costs ~1100 clk of that, a T-in wave (load, gain, section 0) ~600, a
// T-ff wave (feed-forward of the later sections, 16 rows) ~850, the S wave ~600.  The order
// below pairs them so that no SIMD carries much more than a quarter of the total.
enum : u32 { EQ_R = 0x00, EQ_TIN = 0x10, EQ_TFF = 0x20, EQ_S = 0x30, EQ_TS = 0x40 };   // TS: T-ff and S in one wave

// waves of a workgroup: float planes only (NSW == 1) -> the store work rides on the T-ff waves
// (8 waves: two per SIMD, 256 VGPRs each); int16 / VU outputs (NSW == 4) -> S waves of their own
template <int NSEC, int G, int NSW>
constexpr u32 eq_waves()
{
    constexpr u32 nrw = (NSEC + 64 / G - 1) / (64 / G);
    return NSW == 1 ? (NSEC > 1 ? nrw + G / 8 + 2 : nrw + G / 8 + 1) : nrw + G / 8 + (NSEC > 1 ? 2 : 0) + NSW;
}

template <int NRW, int NTF, int NSW>
__device__ __forceinline__ u32 eq_role(u32 wave)
{
    if constexpr (NTF == 2 && NSW == 1) {
        if constexpr (NRW == 2) {                // SIMDs: {R0 Tin0} {R1 Tin1} {TS0 Tin2} {TS1 Tin3}
#if defined(CMHIP_EQ_ORDER) && CMHIP_EQ_ORDER == 1   // {R0 TS0} {R1 TS1} {Tin0 Tin2} {Tin1 Tin3}
            constexpr unsigned char t[8] = {EQ_R | 0, EQ_R | 1, EQ_TIN | 0, EQ_TIN | 1, EQ_TS | 0, EQ_TS | 1,
                                            EQ_TIN | 2, EQ_TIN | 3};
#elif defined(CMHIP_EQ_ORDER) && CMHIP_EQ_ORDER == 2 // {R0 Tin0} {R1 TS0} {Tin1 TS1} {Tin2 Tin3}
            constexpr unsigned char t[8] = {EQ_R | 0, EQ_R | 1, EQ_TIN | 1, EQ_TIN | 2, EQ_TIN | 0, EQ_TS | 0,
                                            EQ_TS | 1, EQ_TIN | 3};
#elif defined(CMHIP_EQ_ORDER) && CMHIP_EQ_ORDER == 3 // {R0 R1} {TS0 TS1} {Tin0 Tin1} {Tin2 Tin3}
            constexpr unsigned char t[8] = {EQ_R | 0, EQ_TS | 0, EQ_TIN | 0, EQ_TIN | 2, EQ_R | 1, EQ_TS | 1,
                                            EQ_TIN | 1, EQ_TIN | 3};
#else
            constexpr unsigned char t[8] = {EQ_R | 0, EQ_R | 1, EQ_TS | 0, EQ_TS | 1, EQ_TIN | 0, EQ_TIN | 1,
                                            EQ_TIN | 2, EQ_TIN | 3};
#endif
            return t[wave];
        } else {                                 // {R0 TS0} {Tin0 TS1} {Tin1 Tin3} {Tin2}
            constexpr unsigned char t[7] = {EQ_R | 0, EQ_TIN | 0, EQ_TIN | 1, EQ_TIN | 2, EQ_TS | 0, EQ_TS | 1,
                                            EQ_TIN | 3};
            return t[wave];
        }
    } else if constexpr (NTF == 2) {
        if constexpr (NRW == 2) {                // {Tff1 Tin0 S0} {R0 Tin1 S1} {R1 Tin2 S2} {Tff0 Tin3 S3}
            constexpr unsigned char t[12] = {EQ_TFF | 1, EQ_R | 0, EQ_R | 1, EQ_TFF | 0, EQ_TIN | 0, EQ_TIN | 1,
                                             EQ_TIN | 2, EQ_TIN | 3, EQ_S | 0, EQ_S | 1, EQ_S | 2, EQ_S | 3};
            return t[wave];
        } else {
            constexpr unsigned char t[11] = {EQ_R | 0, EQ_TIN | 0, EQ_TIN | 2, EQ_TIN | 3, EQ_S | 0, EQ_TIN | 1,
                                             EQ_TFF | 0, EQ_TFF | 1, EQ_S | 1, EQ_S | 2, EQ_S | 3};
            return t[wave];
        }
    } else if constexpr (NSW == 1) {             // one section: no T-ff waves
        constexpr unsigned char t[6] = {EQ_R | 0, EQ_TIN | 0, EQ_TIN | 1, EQ_TIN | 2, EQ_S | 0, EQ_TIN | 3};
        return t[wave];
    } else {
        constexpr unsigned char t[9] = {EQ_R | 0, EQ_TIN | 0, EQ_TIN | 1, EQ_TIN | 2, EQ_S | 0, EQ_TIN | 3,
                                        EQ_S | 1, EQ_S | 2, EQ_S | 3};
        return t[wave];
    }
}

template <int NSEC, int G, int NSW, int CH>
__global__ __launch_bounds__((eq_waves<NSEC, G, NSW>() * 64))
void k_eq_pipe(EqArgs a)
{
    constexpr bool MONO = CH == 1, STEREO = CH == 2;
    static_assert(G == 32, "the wave order below is laid out for 32 rows per workgroup");
    constexpr u32 EP_TB = 64;                     // frames per block
    constexpr u32 EP_ROW = EP_TB + 4;             // floats per LDS row
    constexpr u32 EP_TILE = G * EP_ROW;           // floats per buffer slot
    constexpr u32 SPW = 64 / G;                   // sections per R wave
    constexpr u32 NRW = (NSEC + SPW - 1) / SPW;   // R waves
    constexpr u32 NTF = NSEC > 1 ? 2 : 0;         // T-ff waves: G / 2 rows each, in PASSES of 8 rows
    constexpr u32 PASSES = 2;
    constexpr u32 NBUF = 2 * NSEC;                // F_0, Y_0, F_1, Y_1, ...
    constexpr u32 SPR = EP_TB / 4;                // store lanes per row (4 frames each)
    constexpr u32 RPI = 64 / SPR;                 // rows per store instruction
    constexpr int DPP_SHR1 = 0x111, DPP_SHL7 = 0x107;
    // the register prefetch of the R waves needs 128 VGPRs for two rows: only where the workgroup
    // has at most two waves per SIMD (256 VGPRs each); with S waves of their own it would spill
    constexpr bool RLAG = CMHIP_EQ_RLAG && eq_waves<NSEC, G, NSW>() <= 8;
    constexpr u32 HOP = RLAG ? 3 : 2;             // steps from F_k to F_k+1
    extern __shared__ float lds[];                // NBUF buffers x 2 slots x EP_TILE floats, then G counts
    u32 *nfr_lds = reinterpret_cast<u32 *>(lds + NBUF * 2 * EP_TILE);
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const u32 C = MONO ? 1u : STEREO ? 2u : a.channels;
    const u32 SPG = G / C;                                // whole streams of this workgroup
    const u32 s0 = blockIdx.x * SPG;
    // row r of the workgroup: channel r % C of stream s0 + r / C
    auto row_stream = [&](u32 r, u32 &ch) -> u32 {
        const u32 q = MONO ? r : r / C;
        ch = MONO ? 0u : r - q * C;
        return q < SPG ? s0 + q : 0xffffffffu;            // rows past the last whole stream idle
    };

    const u32 role = eq_role<(int)NRW, (int)NTF, NSW>(wave);
    const bool is_rec = (role & 0xf0u) == EQ_R;
    const bool is_tin = (role & 0xf0u) == EQ_TIN;
    const bool is_tff = (role & 0xf0u) == EQ_TFF || (role & 0xf0u) == EQ_TS;
    // store work: the G / 4 row slots (four rows each) of Y_last are dealt out by role.  Float planes
    // only, three or four sections: CMHIP_EQ_RSLOTS slots to each R wave, the rest to the T-ff waves
    // (evens the SIMDs out); two sections: half to each T-ff wave; otherwise the S waves share them.
    constexpr bool S_ON_R = NSW == 1 && NRW == 2 && NTF == 2;
    constexpr u32 NSLOT = G / 4;
    constexpr u32 RSL = CMHIP_EQ_RSLOTS, TSL = (NSLOT - 2 * RSL) / 2;    // slots of an R / a T-ff wave (S_ON_R)
    constexpr u32 NSL = S_ON_R ? (RSL > TSL ? RSL : TSL) : (NSW == 1 && NTF == 2) ? NSLOT / 2 : NSLOT / NSW;
    u32 s_first = 0, s_cnt = 0;                           // (NSL: most slots of a wave)
    if (S_ON_R) {
        if ((role & 0xf0u) == EQ_R) { s_first = RSL * (role & 15u); s_cnt = RSL; }
        if ((role & 0xf0u) == EQ_TS) { s_first = 2u * RSL + TSL * (role & 15u); s_cnt = TSL; }
    } else if ((role & 0xf0u) == EQ_S || (role & 0xf0u) == EQ_TS) {
        s_first = NSL * (role & 15u);
        s_cnt = NSL;
    }
    const bool is_store = s_cnt != 0;
    const u32 tw = role & 15u;                            // T-in / T-ff wave index

    // R lanes: section and stream row
    const u32 sec = (role & 15u) * SPW + lane / G;
    const bool has_sec = is_rec && sec < (u32)NSEC;
    const u32 row = lane % G;
    u32 my_ch;
    const u32 sl = row_stream(row, my_ch);
    const bool live = sl < a.streams;
    const u32 sidx = live ? sl * C + my_ch : 0u;           // EqState index of this row
    const u32 my_nfr = live ? (a.nframes ? a.nframes[sl] : a.frames) : 0u;
    if (wave == 0 && lane < G)
        nfr_lds[lane] = my_nfr;
    u32 nmax = my_nfr;
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1)
        nmax = max(nmax, (u32)__shfl_xor((int)nmax, o, 64));
    const u32 nblocks = (nmax + EP_TB - 1) / EP_TB;
    const u32 nsteps = nblocks + HOP * NSEC;
    __syncthreads();

    // R lanes own y1, y2 of their section (EqState: x1 x2 y1 y2)
    float d1 = 0, d2 = 0, h1 = 0, h2 = 0;
    if (has_sec && live) {
        const float *c = a.eq[sl].coef[sec];
        const float *st = a.state[sidx].s[sec];
        d1 = -c[3]; d2 = -c[4];
        h1 = st[2]; h2 = st[3];
    }

    // T lanes: stream row l_r, frames l_t8 .. l_t8+7 of every block
    const u32 l_r = 8u * tw + lane / 8u;
    const u32 l_c = lane % 8u;
    const u32 l_t8 = l_c * 8u;
    u32 l_ch;
    const u32 l_stream = row_stream(l_r, l_ch);
    const u32 l_s = min(l_stream, a.streams - 1);
    const bool l_live = is_tin && l_stream < a.streams;
    const u32 l_sidx = l_s * C + l_ch;
    u32 l_magic = 0, l_shift = 0, l_g2 = 0, l_n = 0, l_m = 0;
    // feed-forward registers: b0 b1 b2 and x[t-1], x[t-2] before the next block.  A T-in lane
    // uses [0][0] for section 0 of its row; a T-ff lane [p][k] for section k of the row of pass p.
    float fc[PASSES][NSEC][3];
    float sx1[PASSES][NSEC], sx2[PASSES][NSEC];
#pragma unroll
    for (u32 p = 0; p < PASSES; p++)
#pragma unroll
        for (int k = 0; k < NSEC; k++) {
            fc[p][k][0] = fc[p][k][1] = fc[p][k][2] = 0.f;
            sx1[p][k] = sx2[p][k] = 0.f;
        }
    if (is_tin) {
        l_magic = a.param[l_s].magic;
        l_shift = a.param[l_s].shift;
        l_g2 = a.param[l_s].gain2[l_ch];
        l_m = MONO ? 0u : a.param[l_s].chmap[l_ch];       // the input channel this row reads
        l_n = nfr_lds[l_r];
        if (l_live) {
            // section 0 sees integer-valued samples (the 2^-15 of "x / 32768.f" is not applied
            // by the conversion): it is folded into the coefficients instead, which is
            // bit-identical because power-of-two scaling commutes with every rounding of the
            // chain.  Its history is kept in the same unscaled form inside the kernel and
            // converted at the EqState boundary.
            const float *c = a.eq[l_s].coef[0];
            const float *st = a.state[l_sidx].s[0];
            fc[0][0][0] = c[0] * (1.0f / 32768.0f);
            fc[0][0][1] = c[1] * (1.0f / 32768.0f);
            fc[0][0][2] = c[2] * (1.0f / 32768.0f);
            sx1[0][0] = st[0] * 32768.0f;
            sx2[0][0] = st[1] * 32768.0f;
        }
    }
    u32 f_r[PASSES];                                      // T-ff: row of pass p
#pragma unroll
    for (u32 p = 0; p < PASSES; p++) {
        f_r[p] = (G / 2u) * tw + 8u * p + lane / 8u;
        if (is_tff) {
            u32 fch;
            const u32 fs = row_stream(f_r[p], fch);
            if (fs < a.streams) {
#pragma unroll
                for (int k = 1; k < NSEC; k++) {
                    const float *c = a.eq[fs].coef[k];
                    const float *st = a.state[fs * C + fch].s[k];
                    fc[p][k][0] = c[0]; fc[p][k][1] = c[1]; fc[p][k][2] = c[2];
                    sx1[p][k] = st[0]; sx2[p][k] = st[1];
                }
            }
        }
    }

    // gain disabled (scale 0, or every gain equal to the scale) is stored as 1/1: x -> x
    const bool gain_off = __all(l_g2 == 2u && l_shift == 0u);

    // The T waves keep two blocks of PCM in flight: a block's HBM latency is hidden behind
    // two pipeline steps.  The load is unconditional (address clamped into the stream's own
    // row, which is a multiple of 8 samples long) and nothing else in a T wave touches
    // global memory inside the loop, so the compiler can wait with vmcnt(1) for the older
    // block instead of draining the queue; the stores have a wave of their own.
    const int16_t *l_src = a.in + (u64)l_s * a.stride;
    // (EQ batches have rows of whole 8-frame chunks, so a chunk with a valid frame is never clamped)
    const u32 l_last = (u32)a.stride - (MONO ? 8u : STEREO ? 16u : 1u);
    const u32 l_sel = l_m ? 0x07060302u : 0x05040100u;    // STEREO: which halves of two frames make a pair
    struct Pcm { u32x4 a, b; };                           // a chunk in flight (b: second half, STEREO only)
    auto fetch = [&](u32 b) -> Pcm {
        const u32 f0 = b * EP_TB + l_t8;
        Pcm r = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        if (CMHIP_EQ_ABL & 2) {
            r.a = u32x4{f0, f0 * 3u, f0 * 5u, f0 * 7u};
        } else if constexpr (MONO) {
            r.a = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(l_src + min(f0, l_last)));
        } else if constexpr (STEREO) {                    // 8 frames x (L, R): 32 bytes
            const u32x4 *p = reinterpret_cast<const u32x4 *>(l_src + min(2u * f0, l_last));
            r.a = __builtin_nontemporal_load(p);
            r.b = __builtin_nontemporal_load(p + 1);
        } else {
            u32 h[8];                                     // frames f0..f0+7 of input channel l_m
#pragma unroll
            for (u32 k = 0; k < 8; k++)
                h[k] = *reinterpret_cast<const uint16_t *>(l_src + min((f0 + k) * C + l_m, l_last));
            r.a = u32x4{h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
        }
        return r;
    };
    Pcm wa = {{0, 0, 0, 0}, {0, 0, 0, 0}}, wb = wa;       // blocks of even / odd steps
    if (is_tin) {
        wa = fetch(0);
        wb = fetch(1);
    }

    // feed-forward half of section k on the 8 samples of this lane; the two samples before
    // them come from the lane to the left, or (first chunk) from the end of the last block
    auto feed_forward = [&](const u32 p, const int k, const float (&x)[8], float (&f)[8]) {
        const float p1 = dpp_f32<DPP_SHR1>(x[7]), p2 = dpp_f32<DPP_SHR1>(x[6]);
        const float q1 = dpp_f32<DPP_SHL7>(sx1[p][k]), q2 = dpp_f32<DPP_SHL7>(sx2[p][k]);
        const float xm1 = l_c == 0 ? q1 : p1;
        const float xm2 = l_c == 0 ? q2 : p2;
        sx1[p][k] = x[7];
        sx2[p][k] = x[6];
        const float c0 = fc[p][k][0], c1 = fc[p][k][1], c2 = fc[p][k][2];
        f[0] = __builtin_fmaf(c2, xm2, __builtin_fmaf(c1, xm1, c0 * x[0]));
        f[1] = __builtin_fmaf(c2, xm1, __builtin_fmaf(c1, x[0], c0 * x[1]));
#pragma unroll
        for (int j = 2; j < 8; j++)
            f[j] = __builtin_fmaf(c2, x[j - 2], __builtin_fmaf(c1, x[j - 1], c0 * x[j]));
    };

#ifdef CMHIP_EQ_STAMPS
    u64 st_busy = 0, st_p[3] = {0, 0, 0};
    const u64 st_begin = __builtin_readcyclecounter();
#endif
    // Schedule (HOP = 3): F_k of block b is written in step b+3k, loaded into the R lanes'
    // registers in step b+3k+1 while they still work on block b-1, turned into Y_k in step
    // b+3k+2, and the block leaves in step b+3*NSEC.  Every buffer is read one step after it
    // was written, so two slots per buffer are enough (the third copy is in VGPRs).
    auto rec_step = [&](float4 (&v)[EP_TB / 4], float4 (&nxt)[EP_TB / 4], const u32 step) {
        if (!(CMHIP_EQ_ABL & 32)) {
            const u32 first = HOP * sec + HOP - 1u;       // step in which block 0 is worked on
            const u32 b = step - first;
            if (RLAG) {                                   // next block's row: into registers now
                const u32 bp = b + 1u;
                if (has_sec && step + 1u >= first && bp < nblocks) {
                    const float4 *in = reinterpret_cast<const float4 *>(
                        lds + ((2u * sec) * 2u + (bp & 1u)) * EP_TILE + row * EP_ROW);
#pragma unroll
                    for (u32 t = 0; t < EP_TB / 4; t++)
                        nxt[t] = in[t];
                }
            }
            if (has_sec && step >= first && b < nblocks) {
                float4 *out = reinterpret_cast<float4 *>(
                    lds + ((2u * sec + 1u) * 2u + (b & 1u)) * EP_TILE + row * EP_ROW);
                if (!RLAG) {
                    const float4 *in = reinterpret_cast<const float4 *>(
                        lds + ((2u * sec) * 2u + (b & 1u)) * EP_TILE + row * EP_ROW);
#pragma unroll
                    for (u32 t = 0; t < EP_TB / 4; t++)  // whole row first: 16 LDS reads in flight
                        v[t] = in[t];
                }
                const u32 done = b * EP_TB;
                const u32 cnt = my_nfr > done ? min(my_nfr - done, EP_TB) : 0u;
                if (CMHIP_EQ_ABL & 4) {
#pragma unroll
                    for (u32 t = 0; t < EP_TB / 4; t++)
                        out[t] = v[t];
                } else if (__all(cnt == EP_TB)) {
#pragma unroll
                    for (u32 t = 0; t < EP_TB / 4; t++) {
                        float4 y;
                        y.x = __builtin_fmaf(d1, h1, __builtin_fmaf(d2, h2, v[t].x));
                        y.y = __builtin_fmaf(d1, y.x, __builtin_fmaf(d2, h1, v[t].y));
                        y.z = __builtin_fmaf(d1, y.y, __builtin_fmaf(d2, y.x, v[t].z));
                        y.w = __builtin_fmaf(d1, y.z, __builtin_fmaf(d2, y.y, v[t].w));
                        h2 = y.z;
                        h1 = y.w;
                        if (!(CMHIP_EQ_ABL & 8) || t == 0)
                            out[t] = y;
                    }
                } else {
                    // some stream ends inside this block: same arithmetic, but the history of a
                    // lane moves only on its real samples (what lies beyond is never stored)
#pragma unroll
                    for (u32 t = 0; t < EP_TB / 4; t++) {
                        const float xs[4] = {v[t].x, v[t].y, v[t].z, v[t].w};
                        float rs[4];
#pragma unroll
                        for (u32 j = 0; j < 4; j++) {
                            const float r = __builtin_fmaf(d1, h1, __builtin_fmaf(d2, h2, xs[j]));
                            const bool real = 4u * t + j < cnt;
                            h2 = real ? h1 : h2;
                            h1 = real ? r : h1;
                            rs[j] = r;
                        }
                        out[t] = make_float4(rs[0], rs[1], rs[2], rs[3]);
                    }
                }
            }
        }
    };
    float keep1 = 0.f, keep2 = 0.f;                       // section 0's new x1 / x2, if seen
    bool has1 = false, has2 = false;
    auto tin_step = [&](Pcm &wcur, const u32 step) {
        if (!(CMHIP_EQ_ABL & 128)) {
#ifdef CMHIP_EQ_STAMPS
            const u64 st_tt = __builtin_readcyclecounter();
#endif
            // --- input block `step`: PCM -> gain -> float -> feed-forward of section 0 -> F_0
            // (also in the drain steps at the end, where it works on zeros: keeping the load
            // unconditional is what lets the wait above be counted)
            if (!(CMHIP_EQ_ABL & 64)) {
                const u32 b = step;
                const u32 f0 = b * EP_TB + l_t8;
                const bool have = f0 < l_n;               // beyond the end of the stream: zeros
                u32 w[4] = {wcur.a.x, wcur.a.y, wcur.a.z, wcur.a.w};
                if constexpr (STEREO) {                   // this row's channel of the eight frames
                    w[0] = __builtin_amdgcn_perm(wcur.a.y, wcur.a.x, l_sel);
                    w[1] = __builtin_amdgcn_perm(wcur.a.w, wcur.a.z, l_sel);
                    w[2] = __builtin_amdgcn_perm(wcur.b.y, wcur.b.x, l_sel);
                    w[3] = __builtin_amdgcn_perm(wcur.b.w, wcur.b.z, l_sel);
                }
#pragma unroll
                for (u32 q = 0; q < 4; q++)
                    w[q] = have ? w[q] : 0u;              // (a chunk the stream ends in keeps what
                wcur = fetch(b + 2);                      // follows in the row: never stored)
#ifdef CMHIP_EQ_STAMPS
                u32 stw = w[0];
                asm volatile("" : "+v"(stw));
                st_p[0] += __builtin_readcyclecounter() - st_tt;          // PCM of this block has arrived
#endif
                // gain in integers (exact), then straight to float: the magnitude is converted,
                // the sign bit of the sample is copied in, and one med3 is the int16 saturation
                float x[8], f[8];
                if (gain_off) {                           // no master gain on any row of this wave
#pragma unroll
                    for (u32 q = 0; q < 4; q++) {
                        x[2 * q] = (float)(int)(int16_t)(w[q] & 0xffffu);
                        x[2 * q + 1] = (float)((int)w[q] >> 16);
                    }
                } else {
#pragma unroll
                    for (u32 q = 0; q < 4; q++) {
                        const u32 sg = pk_sign(w[q]);
                        const u32 aw = pk_sub(w[q] ^ sg, sg);
                        const u32 n0 = __umul24(aw & 0xffffu, l_g2);
                        const u32 n1 = __umul24(aw >> 16, l_g2);
                        const float m0 = (float)(__umulhi(n0, l_magic) >> l_shift);
                        const float m1 = (float)(__umulhi(n1, l_magic) >> l_shift);
                        const u32 b0 = (__builtin_bit_cast(u32, m0) & 0x7fffffffu) | ((w[q] << 16) & 0x80000000u);
                        const u32 b1 = (__builtin_bit_cast(u32, m1) & 0x7fffffffu) | (w[q] & 0x80000000u);
                        x[2 * q] = __builtin_amdgcn_fmed3f(__builtin_bit_cast(float, b0), -32768.0f, 32767.0f);
                        x[2 * q + 1] = __builtin_amdgcn_fmed3f(__builtin_bit_cast(float, b1), -32768.0f, 32767.0f);
                    }
                }
                // The last real samples of a stream are section 0's x1/x2 for the next launch
                // (unscaled form here, see above); they pass through exactly one lane each and
                // are written after the loop.
                const u32 bf = b * EP_TB;
                if (l_live && l_n > bf && l_n <= bf + EP_TB) {
                    const u32 e1 = l_n - 1u - bf;                     // last sample, block relative
                    if ((e1 >> 3) == l_c) {
                        float val = x[0];
#pragma unroll
                        for (u32 j = 1; j < 8; j++)
                            val = (e1 & 7u) == j ? x[j] : val;
                        keep1 = val;
                        has1 = true;
                    }
                    if (e1 >= 1u) {
                        const u32 e2 = e1 - 1u;
                        if ((e2 >> 3) == l_c) {
                            float val = x[0];
#pragma unroll
                            for (u32 j = 1; j < 8; j++)
                                val = (e2 & 7u) == j ? x[j] : val;
                            keep2 = val;
                            has2 = true;
                        }
                    } else if (l_c == 7u) {
                        keep2 = sx1[0][0];                            // the sample before this block
                        has2 = true;
                    }
                }
#ifdef CMHIP_EQ_STAMPS
                asm volatile("" : "+v"(x[7]));
                st_p[1] += __builtin_readcyclecounter() - st_tt;          // converted
#endif
                feed_forward(0, 0, x, f);
                float4 *dst = reinterpret_cast<float4 *>(lds + (b & 1u) * EP_TILE + l_r * EP_ROW + l_t8);
                dst[0] = make_float4(f[0], f[1], f[2], f[3]);
                dst[1] = make_float4(f[4], f[5], f[6], f[7]);
#ifdef CMHIP_EQ_STAMPS
                asm volatile("" ::: "memory");
                st_p[2] += __builtin_readcyclecounter() - st_tt;          // F_0 handed to the LDS queue
#endif
            }
        }
    };
    // T-ff waves: feed-forward of the later sections, Y_k-1 -> F_k, for G / 2 rows in PASSES of
    // eight (each with the history registers of its own rows)
    auto tff_step = [&](const u32 step) {
        if (!(CMHIP_EQ_ABL & (128 | 16))) {
            // reads first, then arithmetic: one LDS latency per step where the registers allow
            // it (two waves per SIMD), one per pass otherwise
            constexpr u32 PG = RLAG ? PASSES : 1;         // passes whose rows are loaded together
#pragma unroll
            for (u32 p0 = 0; p0 < PASSES; p0 += PG) {
                float4 yin[PG][NSEC][2];
#pragma unroll
                for (int k = 1; k < NSEC; k++) {
                    const u32 b = step - HOP * (u32)k;
                    if (step >= HOP * (u32)k && b < nblocks) {
#pragma unroll
                        for (u32 p = 0; p < PG; p++) {
                            const float4 *src = reinterpret_cast<const float4 *>(
                                lds + ((2u * k - 1u) * 2u + (b & 1u)) * EP_TILE + f_r[p0 + p] * EP_ROW + l_t8);
                            yin[p][k][0] = src[0];
                            yin[p][k][1] = src[1];
                        }
                    }
                }
#pragma unroll
                for (int k = 1; k < NSEC; k++) {
                    const u32 b = step - HOP * (u32)k;
                    if (step >= HOP * (u32)k && b < nblocks) {
#pragma unroll
                        for (u32 p = 0; p < PG; p++) {
                            const float4 v0 = yin[p][k][0], v1 = yin[p][k][1];
                            const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                            float f[8];
                            feed_forward(p0 + p, k, x, f);
                            float4 *dst = reinterpret_cast<float4 *>(
                                lds + ((2u * k) * 2u + (b & 1u)) * EP_TILE + f_r[p0 + p] * EP_ROW + l_t8);
                            dst[0] = make_float4(f[0], f[1], f[2], f[3]);
                            dst[1] = make_float4(f[4], f[5], f[6], f[7]);
                        }
                    }
                }
            }
        }
    };
    // S wave: per row slot i (rows RPI*i + lane/16) the VU window of the int16 result
    static_assert(RPI == 4, "a row slot is four rows");
    u64 vpw[NSL], vky[NSL], vbase[NSL];
    u32 v_stream[NSL], v_ch[NSL];                         // stream (or none) and channel of the slot's row
#pragma unroll
    for (u32 i = 0; i < NSL; i++) {
        vpw[i] = vky[i] = vbase[i] = 0;
        const u32 r = RPI * (s_first + i) + lane / SPR;
        v_stream[i] = row_stream(r, v_ch[i]);
        if (v_stream[i] >= a.streams || i >= s_cnt)
            v_stream[i] = 0xffffffffu;
        if (is_store && a.vu && v_stream[i] != 0xffffffffu)
            vbase[i] = a.vu[v_stream[i]].samples[a.parity];
    }
    auto s_step = [&](const u32 step) {
        if (!(CMHIP_EQ_ABL & 1)) {
            // --- the finished block of the last section leaves: 256 B (float) / 128 B (int16)
            // per stream row and instruction, fire and forget (this wave never waits for
            // global memory); the int16 form is what the VU meter sees
            const u32 b = step - HOP * NSEC;
            if (step >= HOP * NSEC && b < nblocks) {
                const float *Y = lds + ((2u * NSEC - 1u) * 2u + (b & 1u)) * EP_TILE;
                const u32 t4 = (lane % SPR) * 4u;
                const u32 f0 = b * EP_TB + t4;
                float4 vin[NSL];                              // every LDS read of the step up front:
                u32 nin[NSL];                                 // one latency, not one per row slot
#pragma unroll
                for (u32 i = 0; i < NSL; i++) {
                    const u32 r = min(RPI * (s_first + i) + lane / SPR, (u32)G - 1u);   // (slots past s_cnt are skipped below)
                    nin[i] = nfr_lds[r];
                    vin[i] = *reinterpret_cast<const float4 *>(Y + r * EP_ROW + t4);
                }
#pragma unroll
                for (u32 i = 0; i < NSL; i++) {
                    const u32 n = nin[i];
                    const float4 v = vin[i];
                    const float e[4] = {v.x, v.y, v.z, v.w};
                    const u32 vs_ = v_stream[i], vc_ = v_ch[i];
                    if (vs_ == 0xffffffffu)
                        continue;
                    if (a.f32) {
                        float *dstf = a.f32 + ((u64)vs_ * C + vc_) * a.plane + f0;
                        if (f0 + 4u <= n) {
                            typedef float f32x4 __attribute__((ext_vector_type(4)));
                            const f32x4 vv = {v.x, v.y, v.z, v.w};
                            __builtin_nontemporal_store(vv, reinterpret_cast<f32x4 *>(dstf));
                        } else if (f0 < n) {
                            for (u32 j = 0; j < n - f0; j++)
                                dstf[j] = e[j];
                        }
                    }
                    if (a.out || a.vu) {
                        int q[4];
#pragma unroll
                        for (u32 j = 0; j < 4; j++)
                            q[j] = f32_to_i16(e[j]);
                        const bool whole = __all(f0 + 4u <= n);     // no stream ends inside these
                        if (a.out) {
                            if constexpr (MONO) {
                                int16_t *d16 = a.out + (u64)vs_ * a.stride + f0;
                                if (f0 + 4u <= n) {
                                    typedef u32 u32x2 __attribute__((ext_vector_type(2)));
                                    const u32x2 pk = {((u32)q[0] & 0xffffu) | ((u32)q[1] << 16),
                                                      ((u32)q[2] & 0xffffu) | ((u32)q[3] << 16)};
                                    __builtin_nontemporal_store(pk, reinterpret_cast<u32x2 *>(d16));
                                } else if (f0 < n) {
                                    for (u32 j = 0; j < n - f0; j++)
                                        d16[j] = (int16_t)q[j];
                                }
                            } else {                      // interleaved result: this row's channel
                                int16_t *d16 = a.out + (u64)vs_ * a.stride + (u64)f0 * C + vc_;
#pragma unroll
                                for (u32 j = 0; j < 4; j++)
                                    if (f0 + j < n)
                                        d16[j * C] = (int16_t)q[j];
                            }
                        }
                        if (a.vu) {
                            u32 am[4];
#pragma unroll
                            for (u32 j = 0; j < 4; j++) {
                                am[j] = (u32)(q[j] < 0 ? -q[j] : q[j]);
                                if (!whole)
                                    am[j] = f0 + j < n ? am[j] : 0u;
                            }
#pragma unroll
                            for (u32 j = 0; j < 4; j++)
                                vpw[i] += (u64)am[j] * am[j];
                            u32 m = am[0], jm = 0;                     // first of the largest
#pragma unroll
                            for (u32 j = 1; j < 4; j++) {
                                const bool gt = am[j] > m;
                                m = gt ? am[j] : m;
                                jm = gt ? j : jm;
                            }
                            int qm = q[0];
#pragma unroll
                            for (u32 j = 1; j < 4; j++)
                                qm = jm == j ? q[j] : qm;
                            const u64 kk = make_key(m, vbase[i] + (u64)(f0 + jm) * C + vc_, qm < 0 ? 1u : 0u);
                            vky[i] = kk > vky[i] ? kk : vky[i];
                        }
                    }
                }
            }
        }
    };
#ifdef CMHIP_EQ_STAMPS
#define EQ_STEP(call)                                                   \
    do {                                                                \
        const u64 st_t0 = __builtin_readcyclecounter();                 \
        call;                                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              \
        st_busy += __builtin_readcyclecounter() - st_t0;                \
        __syncthreads();                                                \
    } while (0)
#else
#define EQ_STEP(call) do { call; __syncthreads(); } while (0)
#endif
    // One loop per role (the role never changes, and a loop of its own lets the compiler
    // count a T wave's outstanding loads); every wave passes the same number of barriers.
    const u32 nst2 = (nsteps + 1u) & ~1u;                 // an odd tail step finds nothing to do
    if (is_rec) {
        float4 ra[EP_TB / 4], rb[EP_TB / 4];
#pragma unroll
        for (u32 t = 0; t < EP_TB / 4; t++)
            ra[t] = rb[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (S_ON_R) {                          // the recurrence first, then this wave's share of the stores
            for (u32 step = 0; step < nst2; step += 2) {
#if defined(CMHIP_EQ_V) && CMHIP_EQ_V == 1
                EQ_STEP((s_step(step), rec_step(ra, rb, step)));
                EQ_STEP((s_step(step + 1), rec_step(rb, ra, step + 1)));
#else
                EQ_STEP((rec_step(ra, rb, step), s_step(step)));
                EQ_STEP((rec_step(rb, ra, step + 1), s_step(step + 1)));
#endif
            }
        } else {
            for (u32 step = 0; step < nst2; step += 2) {
                EQ_STEP(rec_step(ra, rb, step));
                EQ_STEP(rec_step(rb, ra, step + 1));
            }
        }
    } else if (is_store) {
        if (is_tff) {                                     // float planes only: T-ff and S in one wave
            for (u32 step = 0; step < nst2; step++)
                EQ_STEP((tff_step(step), s_step(step)));
        } else {
            for (u32 step = 0; step < nst2; step++)
                EQ_STEP(s_step(step));
        }
        if (a.vu) {
            // the 16 lanes of a row hold parts of its window; lane 0 of them is the row's only writer
#pragma unroll
            for (u32 i = 0; i < NSL; i++) {
                u64 pw = vpw[i], ky = vky[i];
#pragma unroll
                for (int o = SPR / 2; o > 0; o >>= 1) {
                    pw += (u64)__shfl_xor((long long)pw, o, 64);
                    const u64 ok = (u64)__shfl_xor((long long)ky, o, 64);
                    ky = ok > ky ? ok : ky;
                }
                const u32 r = RPI * (s_first + i) + lane / SPR;
                if (lane % SPR == 0 && v_stream[i] != 0xffffffffu) {
                    VuState *vs = a.vu + v_stream[i];
                    if (v_ch[i] == 0)
                        vs->samples[a.parity ^ 1u] = vbase[i] + (u64)nfr_lds[r] * C;
                    vs->power[v_ch[i]] += pw;
                    if (ky > vs->key[v_ch[i]])
                        vs->key[v_ch[i]] = ky;
                }
            }
        }
    } else if (is_tff) {
        for (u32 step = 0; step < nst2; step++)
            EQ_STEP(tff_step(step));
    } else {
        for (u32 step = 0; step < nst2; step += 2) {
            EQ_STEP(tin_step(wa, step));
            EQ_STEP(tin_step(wb, step + 1));
        }
        if (has1)
            a.state[l_sidx].s[0][0] = keep1 * (1.0f / 32768.0f);
        if (has2)
            a.state[l_sidx].s[0][1] = keep2 * (1.0f / 32768.0f);
    }
#undef EQ_STEP
#ifdef CMHIP_EQ_STAMPS
    if (blockIdx.x == 7 && lane == 0 && a.dbg) {     // per-role busy cycles (tools/eq_stamps.py)
        a.dbg[2 * wave] = st_busy;
        a.dbg[2 * wave + 1] = __builtin_readcyclecounter() - st_begin;
        a.dbg[41 + wave] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_ID
        a.dbg[40] = nsteps;
        if (is_tin && tw < 2) {                       // two T-in waves: phases inside a step
            for (int i = 0; i < 3; i++)
                a.dbg[50 + 3 * tw + i] = st_p[i];
        }
        a.dbg[20 + wave] = role;
    }
#endif

    // state for the next launch.  y1/y2 of section k are also the x1/x2 of section k+1
    // (its input is this section's output); section 0's x1/x2 were written by the T lanes.
    if (has_sec && live) {
        float *st = a.state[sidx].s[sec];
        st[2] = h1;
        st[3] = h2;
        if (sec + 1u < (u32)NSEC) {
            float *sn = a.state[sidx].s[sec + 1u];
            if (my_nfr >= 2u) {
                sn[0] = h1;
                sn[1] = h2;
            } else if (my_nfr == 1u) {
                sn[1] = sn[0];
                sn[0] = h1;
            }
        }
    }
}

template <int NSEC, int G>
static constexpr size_t eq_pipe_lds_bytes()
{
    return ((size_t)(2 * NSEC) * 2 * G * (64 + 4)) * sizeof(float) + G * sizeof(u32);
}

template <int NSEC, int G, int NSW, int CH>
static hipError_t launch_eq_pipe(const EqArgs &a, hipStream_t st)
{
    constexpr size_t lds_bytes = eq_pipe_lds_bytes<NSEC, G>();
    static_assert(lds_bytes <= 160 * 1024, "tiles of a workgroup must fit the LDS");
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_eq_pipe<NSEC, G, NSW, CH>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess)
            return e;
        configured = true;
    }
    const u32 spg = CH == 1 ? G : G / a.channels;         // whole streams per workgroup
    hipLaunchKernelGGL((k_eq_pipe<NSEC, G, NSW, CH>), dim3((a.streams + spg - 1) / spg),
                       dim3(eq_waves<NSEC, G, NSW>() * 64), lds_bytes, st, a);
    return hipGetLastError();
}

// 32 rows per workgroup: all 256 CUs at 8192 mono streams, and what the LDS holds for four
// sections (8 and 16 rows with several workgroups per CU measured the same or slower)
template <int NSEC>
static hipError_t launch_eq_pipe_g(const EqArgs &a, hipStream_t st)
{
    // the int16 conversion and the VU window are per-sample work of the S waves: more of them
    const bool heavy = a.out || a.vu;
    if (a.channels == 1)
        return heavy ? launch_eq_pipe<NSEC, 32, 4, 1>(a, st) : launch_eq_pipe<NSEC, 32, 1, 1>(a, st);
    if (a.channels == 2 && a.stride >= 16 && a.stride % 16 == 0)
        return heavy ? launch_eq_pipe<NSEC, 32, 4, 2>(a, st) : launch_eq_pipe<NSEC, 32, 1, 2>(a, st);
    return heavy ? launch_eq_pipe<NSEC, 32, 4, 0>(a, st) : launch_eq_pipe<NSEC, 32, 1, 0>(a, st);
}

hipError_t launch_eq(const EqArgs &a, hipStream_t st)
{
    if (a.streams == 0 || a.frames == 0 || !(a.f32 || a.out || a.vu))
        return hipSuccess;
    if (a.channels == 0 || a.channels > MAX_CH)
        return hipErrorInvalidValue;
    switch (a.nsec) {                                     // (0 sections: the caller uses launch_run)
    case 1: return launch_eq_pipe_g<1>(a, st);            // the pipelined kernel, whatever is asked
    case 2: return launch_eq_pipe_g<2>(a, st);            // for (float planes, int16, VU of it)
    case 3: return launch_eq_pipe_g<3>(a, st);
    case 4: return launch_eq_pipe_g<4>(a, st);
    default: return hipErrorInvalidValue;
    }
}

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

__global__ __launch_bounds__(256) void k_node_partial(const VuState *vu, u32 streams, u32 channels,
                                                      u32 parity, u64 first_global, u64 global_step,
                                                      long long *dst)
{
    __shared__ u64 lsum[MAX_CH + 1];
    __shared__ u64 lkey[MAX_CH + 1];
    if (threadIdx.x <= MAX_CH) {
        lsum[threadIdx.x] = 0;
        lkey[threadIdx.x] = 0;
    }
    __syncthreads();
    u64 sum[MAX_CH + 1], key[MAX_CH + 1];
#pragma unroll
    for (u32 c = 0; c <= MAX_CH; c++) {
        sum[c] = 0;
        key[c] = 0;
    }
    for (u32 s = blockIdx.x * 256u + threadIdx.x; s < streams; s += gridDim.x * 256u) {
        const u64 gs = first_global + (u64)s * global_step;
        sum[MAX_CH] += vu[s].samples[parity] / channels;
#pragma unroll
        for (u32 c = 0; c < MAX_CH; c++) {
            if (c < channels) {
                sum[c] += vu[s].power[c];
                const u64 k0 = vu[s].key[c];
                if (k0) {
                    const u64 mag = k0 >> KEY_ABS_SHIFT;
                    const u64 idx = ~(k0 >> 1) & KEY_IDX_MASK;
                    u64 fr = idx / channels;
                    fr = fr > 0x1fffffffull ? 0x1fffffffull : fr;
                    const u64 nk = (mag << 46) | ((0x1fffffffull - fr) << 17) |
                                   ((65535ull - (gs & 65535ull)) << 1) | (k0 & 1ull);
                    key[c] = nk > key[c] ? nk : key[c];
                    key[MAX_CH] = nk > key[MAX_CH] ? nk : key[MAX_CH];
                }
            }
        }
    }
#pragma unroll
    for (u32 c = 0; c <= MAX_CH; c++) {
        const u64 ws = wave_sum(sum[c]);
        const u64 wk = wave_max(key[c]);
        if ((threadIdx.x & 63u) == 0) {
            if (ws)
                atomicAdd(&lsum[c], ws);
            if (wk)
                atomicMax(&lkey[c], wk);
        }
    }
    __syncthreads();
    if (threadIdx.x <= MAX_CH) {
        u64 *d = reinterpret_cast<u64 *>(dst);
        if (lsum[threadIdx.x])
            atomicAdd(&d[threadIdx.x], lsum[threadIdx.x]);
        if (lkey[threadIdx.x])
            atomicMax(&d[MAX_CH + 1 + threadIdx.x], lkey[threadIdx.x]);
    }
}

hipError_t launch_node_partial(const VuState *vu, u32 streams, u32 channels, u32 parity,
                               uint64_t first_global, uint64_t global_step, long long *dst,
                               hipStream_t st)
{
    hipError_t e = hipMemsetAsync(dst, 0, sizeof(long long) * (2 * MAX_CH + 2), st);
    if (e != hipSuccess)
        return e;
    u32 grid = (streams + 255) / 256;
    if (grid > 256)
        grid = 256;
    hipLaunchKernelGGL(k_node_partial, dim3(grid), dim3(256), 0, st, vu, streams, channels,
                       parity, first_global, global_step, dst);
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
