// k_block.hip -- gfx950 (MI355X, wave64) block kernels of the transform -> vumeter path:
// k_run_fast (mono, stereo), k_run_wide (4, 8 channels), k_run_rows (any count, channel maps).
//
// Everything here is pointwise + reduction over packed int16, so the bound is HBM
// bandwidth (no MFMA).  Design rules followed (cdna_hip_programming.md G11-G13,
// Appendix B "Reduction"):
//   * 16 bytes per lane per load/store (global_load_dwordx4), 1 KiB per wave
//     instruction, four loads in flight per lane before the first use;
//   * a wave owns a contiguous chunk of ONE stream, so stream parameters live in
//     SGPRs and the waves share nothing while they work (the mono / stereo runs that write
//     PCM and keep a window put four of them into a workgroup that merges their window
//     sums in LDS at the very end: a quarter of the global atomics);
//   * exact integer arithmetic only: one mul_hi and one 24-bit multiply-add for the division,
//     64-bit integer atomics for the VU window (order independent => bit exact).
//
// Reference semantics restated (never copied):
//   gain     ref: src/transform.c:101-124   q = trunc(x*g/scale) saturated
//   VU       ref: src/vumeter.c:161-177     first max-|x| peak, sum of squares
//   float    ref: src/enc_vorbis.c:108-115  x / 32768.f, planar
#include "cmhip_device.h"
#include <type_traits>

namespace cmhip {

// read-only runs take bigger tiles to amortise the epilogue (picked per channel count from
// interleaved A/B runs, tools/ab_tiles.py); RunTune::vu_tile (4, 8, 16) overrides for tuning
constexpr u32 TILE_U_VUONLY_MONO = 16;         // (8 until the A/B was repeated at sustained clocks: 6.03 -> 6.51 TB/s)
constexpr u32 TILE_U_VUONLY_STEREO = 16;

// One wave = one 4 KiB tile of one stream, one pass: four non-temporal 16-byte loads per
// lane, arithmetic, four non-temporal stores, then a short epilogue.  Short-lived waves
// over small tiles keep the chip-wide access window compact; on MI355X that is worth
// ~15 % of HBM bandwidth over waves that each stream through tens of KiB
// (tools/ubench_copy*.hip: 6.3-6.5 TB/s against 5.0-5.4 TB/s for read+write).
// FULL: the tile lies completely inside the stream's whole vectors -- no bounds tests, no
// zero padding, no ragged tail; this is the case for all but the last tile of a stream.
// Window sums of a workgroup of several waves (FastShare, below): the waves add theirs up in LDS and the
// last one to finish hands the workgroup's to the stream's window.
struct FastShare {
    u64 sum[2], key[2];
    u32 arrived;
};

// window position: read from one slot of VuState::samples, the stream's first tile writes the other
__device__ __forceinline__ u64 window_base(const RunArgs &a, VuState *vs, u32 k, u32 nsamp)
{
    const u64 base = vs->samples[a.parity];
    if (k == 0 && (threadIdx.x & 63u) == 0)
        vs->samples[a.parity ^ 1u] = base + nsamp;
    return base;
}

template <int C, bool WRITE_PCM, bool WRITE_F32, bool DO_VU, int U, bool FULL>
__device__ __forceinline__ void fast_tile(const RunArgs &a, u32 s, u32 k, u32 nsamp, u32 nfull, u32 ntail,
                                          u64 base, VuState *vs, FastShare *share, const int16_t *a_in,
                                          int16_t *a_out, u64 a_stride)
{
    constexpr u32 TILE_U = U;
    constexpr u32 TILE_VEC = 64 * TILE_U;
    const u32 lane = threadIdx.x & 63u;
    const u32 v0 = k * TILE_VEC;

    const int16_t *ins = a_in + (u64)s * a_stride;
    const u32x4 *src = reinterpret_cast<const u32x4 *>(ins);
    int16_t *outs = WRITE_PCM ? a_out + (u64)s * a_stride : nullptr;
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

    // ---- the stream's parameters, read only now: the tile's loads above depend on kernel arguments (and the
    // window position, which run_fast reads first) alone and are on their way.  (Scalar loads return out of order, so a wait for the kernel arguments is a wait for
    // every scalar load issued by then: with the parameter reads ahead of the tile's loads each of these
    // short-lived waves paid the parameter line's miss BEFORE its loads went out -- 8 % of a config-4 launch,
    // round 4.  The barrier keeps the scheduler from moving them back up.)
    __builtin_amdgcn_sched_barrier(0);
    (void)nsamp;
    const StreamParam *p = a.param + s;
    const u32 perm2 = p->perm2;
    // Stereo, nothing but the VU window asked for, and the stream's map the identity or the swap:
    // the samples stay where they are and the two halves of a dword change roles instead (sw = 1:
    // the low half is output channel 1) -- one v_perm_b32 per dword less.  A map that repeats a
    // channel, and every run that writes PCM or floats, permutes as before.
    constexpr bool ROLES = C == 2 && DO_VU && !WRITE_PCM && !WRITE_F32;
    const bool keep = ROLES && (perm2 == 0x03020100u || perm2 == 0x01000302u);
    const u32 sw = keep && perm2 == 0x01000302u ? 1u : 0u;
    // Gains of the two dword halves: integer part and fraction (StreamParam).  Both channels' constants are
    // loaded at FIXED offsets and the roles picked with scalar selects: indexed by `sw` the compiler forms
    // s_load_dword with a register offset AND an immediate one, and on gfx950 (ROCm 7.2) that load came back
    // from base + immediate alone -- both halves got channel 0's fraction (round 4, NOTES_r04).
    const u32 mi01 = p->mi01;                                           // mi[0] | mi[1] << 16
    const u32 mf0 = p->mf[0], mf1 = p->mf[C - 1];
    const u32 mipk = C == 1 ? (mi01 & 0xffffu) * 0x10001u : (sw ? (mi01 >> 16) | (mi01 << 16) : mi01);
    const u32 mflo = sw ? mf1 : mf0, mfhi = sw ? mf0 : mf1;
    // StreamParam::mode: the general form serves every stream; the shorter ones are taken where the
    // VALU binds (a VU window, no PCM result, whole tiles).  The branch comes after the loads are out.
    constexpr bool MODES = DO_VU && !WRITE_PCM && FULL;
    const u32 mode = MODES ? uniform(p->mode) : GAIN_GENERAL;

    if constexpr (ROLES) {
        if (!keep) {                             // (uniform: a wave works on one stream)
#pragma unroll
            for (u32 u = 0; u < TILE_U; u++)
#pragma unroll
                for (u32 i = 0; i < 4; i++)
                    x[u][i] = __builtin_amdgcn_perm(x[u][i], x[u][i], perm2);
        }
    }

    // ---- arithmetic
    u32 qw[TILE_U][4];                           // packed magnitudes, kept for the epilogue
    PowAcc pw[2] = {{0, 0, 0}, {0, 0, 0}};
    u32 best[2] = {0, 0};                        // (magnitude << 16) | (U-1-u) << 6 | (63-lane)
    __shared__ __attribute__((aligned(16))) float fst[WRITE_F32 && C == 1 && FULL ? 64 * 8 : 4];   // mono float planes
    auto arithmetic = [&](auto mode_c) {
    constexpr u32 MODE = decltype(mode_c)::value;
#pragma unroll
    for (u32 u = 0; u < TILE_U; u++) {
        const u32 v = v0 + 64u * u + lane;
        u32 o[4], vmax = 0;
#pragma unroll
        for (u32 i = 0; i < 4; i++) {
            if constexpr (C == 2 && !ROLES)
                x[u][i] = __builtin_amdgcn_perm(x[u][i], x[u][i], perm2);   // stereo channel map
            if constexpr (MODE == GAIN_IDENTITY)
                qw[u][i] = gain2_identity(x[u][i], o[i]);
            else if constexpr (MODE == GAIN_BELOW_SCALE)
                qw[u][i] = gain2_below(x[u][i], mflo, mfhi, o[i]);
            else
                qw[u][i] = gain2<true>(x[u][i], mipk, mflo, mfhi, o[i]);
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
            if constexpr (WRITE_F32 && C == 1 && FULL) {
                // Mono plane, every lane of the wave holding a whole vector: a lane's eight floats
                // are a 32-byte run, so storing them lane by lane would fill half of every line per
                // instruction.  The 512 floats of this step trade places through 2 KiB of LDS
                // instead and leave as two whole-line 16-byte stores per lane.
                constexpr float kf = 1.0f / 32768.0f;
                float f[8];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    f[2 * i] = (float)(int)(short)(o[i] & 0xffffu) * kf;
                    f[2 * i + 1] = (float)((int)o[i] >> 16) * kf;
                }
                __syncthreads();                 // (one wave: the reads of the previous step are done)
                float4 *wr = reinterpret_cast<float4 *>(fst) + 2u * lane;
                wr[0] = make_float4(f[0], f[1], f[2], f[3]);
                wr[1] = make_float4(f[4], f[5], f[6], f[7]);
                __syncthreads();
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                const float4 r0 = reinterpret_cast<const float4 *>(fst)[lane];
                const float4 r1 = reinterpret_cast<const float4 *>(fst)[64u + lane];
                f32x4 *pd = reinterpret_cast<f32x4 *>(f32s + (u64)(v0 + 64u * u) * 8) + lane;
                __builtin_nontemporal_store(f32x4{r0.x, r0.y, r0.z, r0.w}, pd);
                __builtin_nontemporal_store(f32x4{r1.x, r1.y, r1.z, r1.w}, pd + 64);
            } else if constexpr (WRITE_F32) {
                store_f32<C>(f32s, a.plane, v, o);
            }
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
    };
    if constexpr (MODES) {
        if (mode == GAIN_IDENTITY)
            arithmetic(std::integral_constant<u32, GAIN_IDENTITY>{});
        else if (mode == GAIN_BELOW_SCALE)
            arithmetic(std::integral_constant<u32, GAIN_BELOW_SCALE>{});
        else
            arithmetic(std::integral_constant<u32, GAIN_GENERAL>{});
    } else {
        arithmetic(std::integral_constant<u32, GAIN_GENERAL>{});
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
            // (c counts dword halves here; with sw the half's output position is the other one of its frame)
            gkey[c] = make_key(mag, base + 8ull * (v0 + 64u * uw + lw) + (first ^ sw), neg);
        }
        if (lane < (u32)C) {                     // lane = output channel, fed by half lane ^ sw
            const u64 ssum = (lane ^ sw) == 0 ? sum[0] : sum[1];
            const u64 skey = (lane ^ sw) == 0 ? gkey[0] : gkey[1];
            if (share) {                         // (LDS atomics; the workgroup's last wave does the global ones)
                if (ssum)
                    atomicAdd(&share->sum[lane], ssum);
                if (skey)
                    atomicMax(&share->key[lane], skey);
            } else {
                if (ssum)
                    atomicAdd(&vs->power[lane], ssum);
                if (skey)
                    atomicMax(&vs->key[lane], skey);
            }
        }
    }
}

// NW waves per workgroup, each with a tile of its own (NW consecutive tiles of one stream).  The waves
// share nothing while they work; with a VU window they merge their sums in LDS and whichever finishes
// last adds the workgroup's to the window -- 1/NW of the global atomics, which are carried out far from
// the CU and cost the one-wave form 2-8 % of a launch (tools/placement_forms.py).
template <int C, bool WRITE_PCM, bool WRITE_F32, bool DO_VU, int U, int NW = 1>
__device__ __forceinline__ void run_fast(const RunArgs &a)
{
    constexpr u32 TILE_VEC = 64 * U;
    static_assert(NW == 1 || !WRITE_F32, "the float-plane forms stage through LDS with one-wave barriers");
    const u32 lane = threadIdx.x & 63u;
    // Every kernel argument a tile's loads need is read HERE, with the first batch of scalar loads: fetched where
    // they are used they come behind two more waits -- and a scalar wait is a wait for everything issued so far.
    // (No asm volatile to pin them: it makes every later load of the kernel a vector load.)
    const int16_t *a_in = a.in;
    int16_t *a_out = a.out;
    const u64 a_stride = a.stride;
    const u32 a_frames = a.frames;
    u32 s, k;
    if constexpr (NW == 1) {
        s = blockIdx.x / a.chunks;               // stream
        k = blockIdx.x - s * a.chunks;           // tile inside the stream
    } else {
        const u32 cw = (a.chunks + (u32)NW - 1u) / (u32)NW;      // workgroups per stream
        s = blockIdx.x / cw;
        k = (blockIdx.x - s * cw) * (u32)NW + (threadIdx.x >> 6);
    }
    __shared__ FastShare shared_;
    FastShare *share = NW > 1 && DO_VU ? &shared_ : nullptr;
    if constexpr (NW > 1 && DO_VU) {
        if (threadIdx.x < 2u) {
            shared_.sum[threadIdx.x] = 0;
            shared_.key[threadIdx.x] = 0;
        }
        if (threadIdx.x == 0)
            shared_.arrived = 0;
        __syncthreads();                         // (the only barrier: all waves are at their start)
    }

    const u32 nfr = a.nframes ? a.nframes[s] : a_frames;
    const u32 nsamp = nfr * (u32)C;
    const u32 nfull = nsamp >> 3;                // whole 16-byte vectors
    const u32 ntail = nsamp & 7u;                // samples in the partial last vector
    const u32 v0 = k * TILE_VEC;

    VuState *vs = DO_VU ? a.vu + s : nullptr;
    u64 base = 0;
    if constexpr (DO_VU)
        base = window_base(a, vs, k, nsamp);
    if (v0 >= nfull + (ntail ? 1u : 0u)) {
        if constexpr (NW == 1)
            return;
    } else if (v0 + TILE_VEC <= nfull) {
        fast_tile<C, WRITE_PCM, WRITE_F32, DO_VU, U, true>(a, s, k, nsamp, nfull, ntail, base, vs, share, a_in, a_out, a_stride);
    } else {
        fast_tile<C, WRITE_PCM, WRITE_F32, DO_VU, U, false>(a, s, k, nsamp, nfull, ntail, base, vs, share, a_in, a_out, a_stride);
    }
    if constexpr (NW > 1 && DO_VU) {
        // LDS serves a wave's operations in order, and the waves' one after the other: the wave that counts
        // itself last finds every other wave's sums in place
        // (the hardware keeps that order; the waits and the "memory" clobbers keep the compiler from moving
        // this wave's sums behind its count, or the last wave's reads ahead of it)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        u32 n = 0;
        if (lane == 0)
            n = atomicAdd(&shared_.arrived, 1u);
        n = (u32)__builtin_amdgcn_readfirstlane((int)n);
        asm volatile("" ::: "memory");
        if (n == (u32)NW - 1u && lane < (u32)C) {
            const u64 ssum = shared_.sum[lane], skey = shared_.key[lane];
            if (ssum)
                atomicAdd(&vs->power[lane], ssum);
            if (skey)
                atomicMax(&vs->key[lane], skey);
        }
    }
}

// Waves per SIMD.  The form that writes PCM and keeps a window (NW waves per workgroup: configs 2, 4, 5) is held to
// FOUR waves per SIMD: on the same 16 array pairs (PF_SNAP=1 tools/placement_forms.py, round 3) 0.3294 ms against
// 0.3447 ms at the seven waves its 72 registers would allow -- 5 waves 0.3364, 6 0.3489, 3 0.3438, 2 0.427, and 8
// (64 registers, spills) 0.363.  Fewer waves in flight keep the chip-wide access window compact, as the short tiles
// do; with sixteen waves a CU still has 64 KiB of loads under way.  $CMHIP_FAST_WPE at build time overrides (A/B).
#ifndef CMHIP_FAST_WPE
#define CMHIP_FAST_WPE 4
#endif
// (The other families were swept the same way and gain nothing: the one-wave PCM-only form is within noise for 3-8
// waves, k_run_wide at its best as it is, and k_run_rows / the read-only entry already run at the 3-4 waves their
// registers leave -- forcing more makes them spill: profiles/r03_occupancy_forms.txt.)
template <int C, bool WRITE_PCM, bool WRITE_F32, bool DO_VU, int U, int NW = 1>
__global__ __launch_bounds__(64 * NW)
__attribute__((amdgpu_waves_per_eu(NW > 1 ? CMHIP_FAST_WPE : 1, NW > 1 ? CMHIP_FAST_WPE : 8))) void k_run_fast(RunArgs a)
{
    run_fast<C, WRITE_PCM, WRITE_F32, DO_VU, U, NW>(a);
    done_epilogue(a.done_flag, a.done_seq);
}
// The read-only entry (VU window, no PCM, no floats): at least three waves per SIMD -- the 16 KiB
// tile holds 128 VGPRs of samples and magnitudes, and with the three arithmetic forms in one
// function the allocator would otherwise take 170 and leave two.
template <int C, int U>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3))) void k_run_fast_ro(RunArgs a)
{
    run_fast<C, false, false, true, U>(a);
    done_epilogue(a.done_flag, a.done_seq);
}

// ---------------------------------------------------------------------------
// Wide path: 4 or 8 channels with the identity channel map (the template also covers 16,
// which k_run_rows now serves faster).  Same tile scheme and the
// same packed arithmetic as k_run_fast; a 16-byte vector holds 8/C frames, so every vector
// position has a fixed channel (for 16 channels: fixed per lane parity) and the per-channel
// accumulators live in registers.  NS = min(C, 8) accumulator slots per lane: the half h of
// dword i feeds slot 2*(i % (NS/2)) + h.

// IDENT: no stream of the batch has a gain (the transform as the reference creates it) -- the
// magnitudes are the samples' own; instantiated for the read-only runs, which the VALU binds.
template <int C, bool WRITE_PCM, bool WRITE_F32, bool DO_VU, int U, bool IDENT = false>
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
    if (v0 >= nfull + (ntail ? 1u : 0u)) {
        done_epilogue(a.done_flag, a.done_seq);
        return;
    }

    const StreamParam *p = a.param + s;
    const u32 cls = C == 16 ? (lane & 1u) : 0u;  // v0 and 64*u are even: vector parity = lane parity
    u32 mipk[NG], mf[NS];                        // (per lane class for 16 channels: not uniform)
#pragma unroll
    for (u32 i = 0; i < NS; i++)
        mf[i] = p->mf[i + 8u * cls];
#pragma unroll
    for (u32 g = 0; g < NG; g++)
        mipk[g] = (u32)p->mi[2u * g + 8u * cls] | ((u32)p->mi[2u * g + 1u + 8u * cls] << 16);

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
            if constexpr (IDENT)
                qw[u][i] = gain2_identity(x[u][i], o[i]);
            else
                qw[u][i] = gain2<C != 16>(x[u][i], mipk[g], mf[2 * g], mf[2 * g + 1], o[i]);
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
    done_epilogue(a.done_flag, a.done_seq);
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

// STAGE: planar floats go through LDS so that every plane leaves in whole runs (any count but 16,
// where lanes of equal parity already hold consecutive frames of the same eight channels).
//
// The VU window costs ~3.5 VALU instructions per sample here (it was ~7.5): a lane's eight positions
// have eight different channels, so nothing can be folded across positions, but the work is done per
// STEP of UR = 4 rows instead of per row:
//   * sum of squares: three squares fit a u32 (3 * 2^30), so rows 0-2 of a step are one chain of three
//     v_mad_u32_u16 and row 3 a fourth; two 64-bit additions per position and step, nothing to decide
//     at run time (a counter "flush after three" that lives across the row loop costs a compare, two
//     selects and a 64-bit add per SAMPLE);
//   * peak: one packed maximum over the step's four rows, one 32-bit key (magnitude << 16 | step) per
//     position; a lane whose key improves keeps the step's four signed results of that position
//     (v_bfi_b32 under a mask), and which of the four came first, and its sign, is looked up once
//     per tile in the epilogue.
struct RowsVu {
    u64 pw[8];                                   // sum of squares per position
    u32 best[8];                                 // |peak| << 16 | (0x7fff - step in tile)
    u32 sv[4][4];                                // [dword i][row of the step]: the results of the winning step
};

// one dword column (positions 2i, 2i+1) of one step: q = packed magnitudes, o = packed signed results
__device__ __forceinline__ void rows_vu_column(RowsVu &v, const u32 i, const u32 (&q)[4], const u32 (&o)[4],
                                               const u32 steptag)
{
    const u32 m = pk_max(pk_max(q[0], q[1]), pk_max(q[2], q[3]));
    const u32 klo = (m << 16) | steptag, khi = (m & 0xffff0000u) | steptag;
    const bool blo = klo > v.best[2 * i], bhi = khi > v.best[2 * i + 1];   // (a later step never wins a tie)
    v.best[2 * i] = blo ? klo : v.best[2 * i];
    v.best[2 * i + 1] = bhi ? khi : v.best[2 * i + 1];
    const u32 mask = (blo ? 0xffffu : 0u) | (bhi ? 0xffff0000u : 0u);
#pragma unroll
    for (u32 u = 0; u < 4; u++)
        v.sv[i][u] = (o[u] & mask) | (v.sv[i][u] & ~mask);
    v.pw[2 * i] += sq_lo(q[2], sq_lo(q[1], sq_lo0(q[0])));
    v.pw[2 * i] += sq_lo0(q[3]);
    v.pw[2 * i + 1] += sq_hi(q[2], sq_hi(q[1], sq_hi0(q[0])));
    v.pw[2 * i + 1] += sq_hi0(q[3]);
}

template <bool WRITE_PCM, bool WRITE_F32, bool DO_VU, bool MAP, bool STAGE>
__global__ __launch_bounds__(64) void k_run_rows(RunArgs a, u32 W, u32 rows_per_tile)
{
    constexpr u32 UR = 4;                        // rows per step
    __shared__ u64 lsum[MAX_CH];
    __shared__ u64 lkey[MAX_CH];
    __shared__ u32x4 raw[MAP ? UR * 64 : 1];     // the rows as loaded (MAP only)
    // planar float output: the UR rows of a step, plane-major, so that every plane leaves in
    // whole 16-byte stores of consecutive frames (UR * 8 * 64 floats at most)
    __shared__ __attribute__((aligned(16))) float fstage[WRITE_F32 && STAGE ? UR * 8 * 64 : 4];
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
    if ((u64)row0 * W >= nvec) {
        done_epilogue(a.done_flag, a.done_seq);
        return;
    }
    if constexpr (DO_VU) {
        if (lane < MAX_CH) {
            lsum[lane] = 0;
            lkey[lane] = 0;
        }
    }

    const StreamParam *p = a.param + s;
    const bool active = lane < W;
    const u32 lane_fr = 8u * lane / C;           // whole frames before this lane's vector in a row
    const u32 phase = 8u * lane - lane_fr * C;   // channel of its position 0
    const u32 FW = 8u * W / C;                   // frames per row (8W is a multiple of C)
    // The shorter forms of the gain (StreamParam::mode, chosen per stream by the host) where the VALU counts
    // most: a VU window and nothing written.  The branch is uniform: a wave works on one stream.
    constexpr bool MODES = DO_VU && !WRITE_PCM && !WRITE_F32;
    const u32 smode = uniform(p->mode);          // the stream's gain class (StreamParam::mode)
    const u32 mode = MODES ? smode : GAIN_GENERAL;
    u32 ch[8], df[8], mipk[4], mf[8];
    u32 so[8];                                   // MAP: byte offset of position j's source in its row
#pragma unroll
    for (u32 j = 0; j < 8; j++) {
        const u32 t = phase + j;
        df[j] = t / C;
        ch[j] = t - df[j] * C;
        mf[j] = p->mf[ch[j]];
        so[j] = MAP ? 2u * ((lane_fr + df[j]) * C + p->chmap[ch[j]]) : 0u;
    }
    // The integer parts of the gains are per-lane loads (a lane's positions have channels of their own): eight
    // more in front of every tile -- unless the stream's class says what they are (every gain below the scale:
    // 0; gain disabled or unity: 1), which is the common case.  (Uniform branch: a wave works on one stream.)
    if (smode == GAIN_GENERAL) {
#pragma unroll
        for (u32 i = 0; i < 4; i++)
            mipk[i] = (u32)p->mi[ch[2 * i]] | ((u32)p->mi[ch[2 * i + 1]] << 16);
    } else {
#pragma unroll
        for (u32 i = 0; i < 4; i++)
            mipk[i] = smode == GAIN_IDENTITY ? 0x00010001u : 0u;
    }

    const int16_t *ins = a.in + (u64)s * a.stride;
    const u32x4 *src = reinterpret_cast<const u32x4 *>(ins);
    int16_t *outs = WRITE_PCM ? a.out + (u64)s * a.stride : nullptr;
    u32x4 *dst = reinterpret_cast<u32x4 *>(outs);
    float *f32s = WRITE_F32 ? a.f32 + (u64)s * a.plane * C : nullptr;

    RowsVu vu;
#pragma unroll
    for (u32 j = 0; j < 8; j++) {
        vu.pw[j] = 0;
        vu.best[j] = 0;
    }
#pragma unroll
    for (u32 i = 0; i < 4; i++)
#pragma unroll
        for (u32 u = 0; u < UR; u++)
            vu.sv[i][u] = 0;

    auto tile = [&](auto mode_c) {
    constexpr u32 MODE = decltype(mode_c)::value;
    // gain + VU of one step: x[u][i] in, the signed results back in o[u][i]
    auto arithmetic = [&](const u32 step, const u32 (&x)[UR][4], u32 (&o)[UR][4]) {
        const u32 steptag = 0x7fffu - step;
#pragma unroll
        for (u32 i = 0; i < 4; i++) {
            u32 q[UR], oc[UR];
#pragma unroll
            for (u32 u = 0; u < UR; u++) {
                if constexpr (MODE == GAIN_IDENTITY)
                    q[u] = gain2_identity(x[u][i], oc[u]);
                else if constexpr (MODE == GAIN_BELOW_SCALE)
                    q[u] = gain2_below(x[u][i], mf[2 * i], mf[2 * i + 1], oc[u]);
                else
                    q[u] = gain2<false>(x[u][i], mipk[i], mf[2 * i], mf[2 * i + 1], oc[u]);
                o[u][i] = oc[u];
            }
            if constexpr (DO_VU)
                rows_vu_column(vu, i, q, oc, steptag);
        }
    };

    // everything after the loads of a step: channel map (through LDS), gain + VU, PCM and float stores
    auto finish_step = [&](const u32 r0, u32 (&x)[UR][4], const bool (&full)[UR], const bool (&tail)[UR]) {
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
        u32 o[UR][4];
        arithmetic(r0 / UR, x, o);
#pragma unroll
        for (u32 u = 0; u < UR; u++) {
            const u32 row = row0 + r0 + u;
            const u32 v = row * W + lane;
            if (full[u]) {
                if constexpr (WRITE_PCM) {
                    const u32x4 ov = {o[u][0], o[u][1], o[u][2], o[u][3]};
                    __builtin_nontemporal_store(ov, dst + v);
                }
            } else if (tail[u]) {
                if constexpr (WRITE_PCM) {
                    for (u32 j = 0; j < ntail; j++) {
                        u32 ow = 0;
#pragma unroll
                        for (u32 i = 0; i < 4; i++)
                            if (i == (j >> 1))
                                ow = o[u][i];
                        outs[(u64)v * 8 + j] = (int16_t)((ow >> (16u * (j & 1u))) & 0xffffu);
                    }
                }
            }
            if constexpr (WRITE_F32) {
                if constexpr (!STAGE) {
                    // 16 channels: lanes of equal parity hold consecutive frames of the same eight
                    // channels, so these stores are already runs of a plane
                    const u32 cnt = full[u] ? 8u : (tail[u] ? ntail : 0u);
                    const u32 fr = row * FW + lane_fr;
#pragma unroll
                    for (u32 j = 0; j < 8; j++) {
                        if (j < cnt) {
                            const int q = (int)(short)((o[u][j >> 1] >> (16u * (j & 1u))) & 0xffffu);
                            f32s[(u64)ch[j] * a.plane + fr + df[j]] = q * (1.0f / 32768.0f);
                        }
                    }
                } else if (active) {
                    // this lane's eight samples to [channel][frame of the UR-row step]
#pragma unroll
                    for (u32 j = 0; j < 8; j++) {
                        const int q = (int)(short)((o[u][j >> 1] >> (16u * (j & 1u))) & 0xffffu);
                        fstage[ch[j] * (UR * FW) + u * FW + lane_fr + df[j]] = q * (1.0f / 32768.0f);
                    }
                }
            }
        }
        if constexpr (WRITE_F32 && STAGE) {
            // every plane's UR*FW consecutive frames of this step leave as 16-byte stores
            __syncthreads();
            const u32 fbase = (row0 + r0) * FW;          // first frame of the step (a multiple of 4)
            const u32 fcnt = UR * FW;                     // frames of the step
            for (u32 c = 0; c < C; c++) {
                float *plane_dst = f32s + (u64)c * a.plane + fbase;
                for (u32 i4 = lane * 4u; i4 < fcnt; i4 += 256u) {
                    const float4 v4 = *reinterpret_cast<const float4 *>(&fstage[c * fcnt + i4]);
                    if (fbase + i4 + 4u <= nfr) {
                        typedef float f32x4 __attribute__((ext_vector_type(4)));
                        const f32x4 vv = {v4.x, v4.y, v4.z, v4.w};
                        __builtin_nontemporal_store(vv, reinterpret_cast<f32x4 *>(plane_dst + i4));
                    } else if (fbase + i4 < nfr) {
                        const float e[4] = {v4.x, v4.y, v4.z, v4.w};
                        for (u32 k2 = 0; fbase + i4 + k2 < nfr; k2++)
                            plane_dst[i4 + k2] = e[k2];
                    }
                }
            }
            __syncthreads();                              // before the next step overwrites the stage
        }
    };
    // Steps of whole rows (every vector inside the stream's whole vectors -- everything but a stream's
    // ragged end) take a software-pipelined loop: the loads of the next step are in flight while this
    // step is worked on, all of them unconditional (lanes beyond W repeat lane W-1's vector; what they
    // compute is dropped at the merge and never stored) so that the compiler can count its waits.
    // Without it these runs are latency bound: a wave that loads, waits and then computes leaves the
    // memory idle for as long as it computes.
    // (a stream's last tile takes it for its whole steps and the ragged loop below for the rest)
    u32 piped_rows = 0;
    {
        const u64 whole = (u64)nfull / W;                        // rows of whole vectors in the stream
        const u64 here = whole > row0 ? whole - row0 : 0;
        piped_rows = (u32)(here < rows_per_tile ? here : rows_per_tile) & ~(UR - 1u);
    }
    if (piped_rows) {
        const u32 lw = active ? lane : W - 1u;
        const u32x4 *srow = src + (u64)row0 * W + lw;
        auto load4 = [&](u32 r0, u32 (&xx)[UR][4]) {
#pragma unroll
            for (u32 u = 0; u < UR; u++) {
                const u32x4 w = __builtin_nontemporal_load(srow + (u64)(r0 + u) * W);
                xx[u][0] = w.x; xx[u][1] = w.y; xx[u][2] = w.z; xx[u][3] = w.w;
            }
        };
        bool all_rows[UR], no_tail[UR];
#pragma unroll
        for (u32 u = 0; u < UR; u++) {
            all_rows[u] = active;
            no_tail[u] = false;
        }
        // (every pass of the loop issues its loads unconditionally -- the last step is peeled off)
        u32 xa[UR][4], xb[UR][4];
        load4(0, xa);
        u32 r0 = 0;
        for (; r0 + UR < piped_rows; r0 += UR) {
            load4(r0 + UR, xb);
            finish_step(r0, xa, all_rows, no_tail);
#pragma unroll
            for (u32 u = 0; u < UR; u++)
#pragma unroll
                for (u32 i = 0; i < 4; i++)
                    xa[u][i] = xb[u][i];
        }
        finish_step(r0, xa, all_rows, no_tail);
    }
    for (u32 r0 = piped_rows; r0 < rows_per_tile; r0 += UR) {
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
        finish_step(r0, x, full, tail);
    }
    };
    if constexpr (MODES) {
        if (mode == GAIN_IDENTITY)
            tile(std::integral_constant<u32, GAIN_IDENTITY>{});
        else if (mode == GAIN_BELOW_SCALE)
            tile(std::integral_constant<u32, GAIN_BELOW_SCALE>{});
        else
            tile(std::integral_constant<u32, GAIN_GENERAL>{});
    } else {
        tile(std::integral_constant<u32, GAIN_GENERAL>{});
    }

    if constexpr (DO_VU) {
        __syncthreads();                         // accumulators cleared (one wave: cheap)
#pragma unroll
        for (u32 j = 0; j < 8; j++) {
            const u32 i = j >> 1, sh = 16u * (j & 1u);
            const u32 mag = vu.best[j] >> 16;
            if (!active)                         // (lanes beyond W held copies in the pipelined loop)
                continue;
            if (vu.pw[j])
                atomicAdd(reinterpret_cast<unsigned long long *>(&lsum[ch[j]]), (unsigned long long)vu.pw[j]);
            if (mag) {
                // the first of the winning step's four results with that magnitude, and its sign
                const u32 step = 0x7fffu - (vu.best[j] & 0xffffu);
                u32 first = UR, neg = 0;
#pragma unroll
                for (u32 u = 0; u < UR; u++) {
                    const int sv = (int)(short)((vu.sv[i][u] >> sh) & 0xffffu);
                    const u32 am = (u32)(sv < 0 ? -sv : sv);
                    if (am == mag && first == UR) {
                        first = u;
                        neg = sv < 0 ? 1u : 0u;
                    }
                }
                const u64 v = (u64)(row0 + step * UR + first) * W + lane;
                const u64 key = make_key(mag, base + 8ull * v + j, neg);
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
    done_epilogue(a.done_flag, a.done_seq);
}

// ---------------------------------------------------------------------------
// Launcher of the block kernels: by channel count and by what the batch asks for.

constexpr u32 FAST_NW = 4;

hipError_t launch_run(const RunArgs &a, const RunTune &tune, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop,
                      bool *flagged)
{
    const bool pcm = a.out != nullptr, f32 = a.f32 != nullptr, vu = a.vu != nullptr;
    if (flagged)
        *flagged = false;
    if (a.streams == 0 || a.frames == 0)
        return hipSuccess;
    // completion by flag only for a launch of one workgroup (RunArgs::done_flag)
    auto flag_if_single = [&](RunArgs &b, u32 workgroups) {
        if (workgroups != 1u)
            b.done_flag = nullptr;
        if (flagged)
            *flagged = b.done_flag != nullptr;
    };
    if (a.channels <= 2) {
        // one wave per tile: 4 KiB when PCM or float is written, larger read-only
        RunArgs b = a;
        u32 tile_u = TILE_U;
        if (!pcm && !f32) {
            tile_u = a.channels == 1 ? TILE_U_VUONLY_MONO : TILE_U_VUONLY_STEREO;
            if (tune.vu_tile)                    // tuning knob, validated when the batch was made
                tile_u = tune.vu_tile;
        }
        const u64 nvec = ((u64)a.frames * a.channels + 7) / 8;
        b.chunks = (u32)((nvec + 64ull * tile_u - 1) / (64ull * tile_u));
        if (b.chunks == 0)
            b.chunks = 1;
        if ((u64)b.chunks * a.streams >= (1ull << 31))
            return hipErrorInvalidValue;
        const u32 grid = a.streams * b.chunks;
        // Waves per workgroup of the runs that write PCM and keep a window (see run_fast).  Over 36 pairs of
        // input and output arrays in one process, three boxes (tools/placement_forms.py): 1 wave 0.356-0.361 ms,
        // 2 0.348-0.357, 4 0.346-0.351, 8 0.344-0.349 on config 2 (no window: 0.327-0.332) -- but only the
        // four-wave form also gains from arrays that lie apart (place_arrays_apart in cmhip_batch.hip:
        // 0.333 ms, against 0.345-0.354 for eight waves and 0.340-0.351 for one).  The read-only runs
        // (16 KiB tiles, a quarter of the atomics per byte) lose with more than one wave: 0.171 / 0.175 /
        // 0.181 / 0.207 ms for 1 / 2 / 4 / 8.
        // (streams of fewer than three tiles -- blocks of a few hundred frames -- take the one-wave form: a
        // workgroup of four waves would leave most of its waves idle there, and it is held to four per SIMD)
        const u32 nw = tune.fast_nw ? tune.fast_nw : (b.chunks >= 3u ? FAST_NW : 1u);
        const u32 gridw = a.streams * ((b.chunks + nw - 1u) / nw);
        if ((u64)a.streams * ((b.chunks + nw - 1u) / nw) >= (1ull << 31))
            return hipErrorInvalidValue;
        flag_if_single(b, (pcm && !f32 && vu && nw > 1u) ? gridw : grid);
#define CMHIP_FAST(C, P, F, V, U)                                                  \
    hipExtLaunchKernelGGL((k_run_fast<C, P, F, V, U>), dim3(grid), dim3(64), 0, st, ev_start, ev_stop, 0, b)
#define CMHIP_FAST_W(C, P, F, V, U)                                                \
    do {                                                                           \
        if (nw == 4) hipExtLaunchKernelGGL((k_run_fast<C, P, F, V, U, 4>), dim3(gridw), dim3(256), 0, st, ev_start, ev_stop, 0, b);        \
        else if (nw == 8) hipExtLaunchKernelGGL((k_run_fast<C, P, F, V, U, 8>), dim3(gridw), dim3(512), 0, st, ev_start, ev_stop, 0, b);   \
        else CMHIP_FAST(C, P, F, V, U);                                            \
    } while (0)
#define CMHIP_FAST_RO(C, U)                                                        \
    hipExtLaunchKernelGGL((k_run_fast_ro<C, U>), dim3(grid), dim3(64), 0, st, ev_start, ev_stop, 0, b)
#define CMHIP_FAST_C(C)                                                            \
    do {                                                                           \
        if (pcm && !f32 && vu) CMHIP_FAST_W(C, true, false, true, 4);              \
        else if (!pcm && !f32 && vu && tile_u == 4) CMHIP_FAST_RO(C, 4);           \
        else if (!pcm && !f32 && vu && tile_u == 8) CMHIP_FAST_RO(C, 8);           \
        else if (!pcm && !f32 && vu) CMHIP_FAST_RO(C, 16);                         \
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
#undef CMHIP_FAST_RO
#undef CMHIP_FAST_W
#undef CMHIP_FAST
    } else if ((a.channels == 4 || a.channels == 8) && a.identity_maps && (pcm || f32) &&
               !(a.channels == 4 && f32 && !tune.wide4_f32)) {
        // (4-channel float planes: k_run_wide writes every other float of a line per store;
        // k_run_rows stages the planes through LDS and runs 25 % faster there.  Read-only runs:
        // k_run_rows 5.9-6.0 TB/s on 8 channels against 4.0-4.7 for k_run_wide, round 2.)
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
        flag_if_single(b, grid);
#define CMHIP_WIDE(C, P, F, V)                                                     \
    do {                                                                           \
        if (wu == 16u)                                                             \
            hipExtLaunchKernelGGL((k_run_wide<C, P, F, V, 16>), dim3(grid), dim3(64), 0, st, ev_start, ev_stop, 0, b); \
        else                                                                       \
            hipExtLaunchKernelGGL((k_run_wide<C, P, F, V, 8>), dim3(grid), dim3(64), 0, st, ev_start, ev_stop, 0, b);  \
    } while (0)
#define CMHIP_WIDE_C(C)                                                            \
    do {                                                                           \
        if (pcm && !f32 && vu) CMHIP_WIDE(C, true, false, true);                   \
        else if (!pcm && !f32 && vu && a.identity_gains)                           \
            hipExtLaunchKernelGGL((k_run_wide<C, false, false, true, 16, true>), dim3(grid), dim3(64), 0, st, ev_start, ev_stop, 0, b); \
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
        // (at sustained clocks, tools/bench_generic.py: runs with channel maps gain 3-5 % from 16 rows
        // instead of 8, 16 channels writing PCM 4-16 % from 16 instead of 32)
        const bool ro = !pcm && !f32;
        // (read-only runs: 64 rows -- the merge at the end of a tile costs as much as two steps of the
        // loop; 5.2-6.0 TB/s against 4.8-5.6 with 16 or 32 rows, round 2)
        u32 rpt = ro ? 64u : P == 1 ? (a.channels == 16 ? 16u : 32u) : (a.identity_maps ? 8u : 16u);
        if (tune.rows_rpt)                                     // tuning knob (tools/bench_generic.py)
            rpt = tune.rows_rpt;
        const u64 nvec = ((u64)a.frames * a.channels + 7) / 8;
        const u64 rows = (nvec + W - 1) / W;
        if (ro && !tune.rows_rpt) {
            // equal tiles: a stream of 98 rows is two tiles of 52, not one of 64 and a ragged one of 34
            // whose merge costs as much as the whole tile's (3 and 6 channels at 16 384 frames: +8 %)
            const u64 nt = (rows + rpt - 1) / rpt;
            rpt = (u32)(((rows + nt - 1) / nt + 3u) & ~3ull);
        }
        b.chunks = (u32)((rows + rpt - 1) / rpt);
        if (b.chunks == 0)
            b.chunks = 1;
        if ((u64)b.chunks * a.streams >= (1ull << 31))
            return hipErrorInvalidValue;
        const u32 grid = a.streams * b.chunks;
        flag_if_single(b, grid);
#define CMHIP_ROWS(P_, F_, V_)                                                                      \
    do {                                                                                            \
        constexpr bool S_ = F_;                        /* staged float planes unless 16 channels */ \
        if (a.identity_maps && (!F_ || a.channels != 16))                                           \
            hipExtLaunchKernelGGL((k_run_rows<P_, F_, V_, false, S_>), dim3(grid), dim3(64), 0, st, ev_start, ev_stop, 0, b, W, rpt); \
        else if (a.identity_maps)                                                                   \
            hipExtLaunchKernelGGL((k_run_rows<P_, F_, V_, false, false>), dim3(grid), dim3(64), 0, st, ev_start, ev_stop, 0, b, W, rpt); \
        else if (!F_ || a.channels != 16)                                                           \
            hipExtLaunchKernelGGL((k_run_rows<P_, F_, V_, true, S_>), dim3(grid), dim3(64), 0, st, ev_start, ev_stop, 0, b, W, rpt);  \
        else                                                                                        \
            hipExtLaunchKernelGGL((k_run_rows<P_, F_, V_, true, false>), dim3(grid), dim3(64), 0, st, ev_start, ev_stop, 0, b, W, rpt);  \
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

}  // namespace cmhip
