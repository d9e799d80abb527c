// cmhip_device.h -- device-side helpers shared by the kernel files: exact gain arithmetic on packed
// int16, the VU window key, DPP wave reductions.  (Included by k_block.hip, k_eq.hip, k_misc.hip.)
#ifndef CMHIP_DEVICE_H
#define CMHIP_DEVICE_H

#include "cmhip_internal.h"
#include <stdlib.h>

namespace cmhip {

using u32 = uint32_t;
using u64 = unsigned long long;


__device__ __forceinline__ u32 uniform(u32 v) { return __builtin_amdgcn_readfirstlane(v); }

// Completion by flag (RunArgs::done_flag; launches of one workgroup only): every wave's stores are made
// visible at system scope, the workgroup meets, one lane stores the sequence number for the host.
__device__ __forceinline__ void done_epilogue(uint32_t *flag, uint32_t seq)
{
    if (flag) {                                  // (uniform: a kernel argument)
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
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
// `out` receives the packed signed result.  floor(|x| * gain / scale) = |x| * mi + mulhi(|x|, mf)
// (StreamParam): per sample one v_mul_hi_u32 and one v_mad_u32_u16, which takes the magnitude and the
// integer part straight from the halves of their packed dwords (mipk = mi of the low half | mi of the
// high half << 16).  UNIFORM: the gains are the same in every lane (a wave works on one stream and the
// dword halves have fixed channels: mono, stereo) and sit in SGPRs.
template <bool UNIFORM>
__device__ __forceinline__ u32 gain2(u32 w, u32 mipk, u32 mflo, u32 mfhi, u32 &out)
{
    const u32 sg = pk_sign(w);
    const u32 aw = pk_sub(w ^ sg, sg);                       // |x| per half (u16, 32768 ok)
    const u32 h0 = __umulhi(aw & 0xffffu, mflo);
    const u32 h1 = __umulhi(aw >> 16, mfhi);
    u32 q0, q1;                                              // < 2^31: 32768 * 65535 + 32767
    if constexpr (UNIFORM) {
        asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[0,0,0,0]" : "=v"(q0) : "v"(aw), "s"(mipk), "v"(h0));
        asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,1,0,0]" : "=v"(q1) : "v"(aw), "s"(mipk), "v"(h1));
    } else {
        asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[0,0,0,0]" : "=v"(q0) : "v"(aw), "v"(mipk), "v"(h0));
        asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,1,0,0]" : "=v"(q1) : "v"(aw), "v"(mipk), "v"(h1));
    }
    u32 qw = __builtin_bit_cast(u32, __builtin_amdgcn_cvt_pk_u16(q0, q1));   // saturates at 65535
    qw = pk_min(qw, pk_sub(0x7fff7fffu, sg));                // 32767, or 32768 for negatives
    out = pk_sub(qw ^ sg, sg);
    return qw;
}

// the shorter forms (StreamParam::mode): every gain below the scale -- one mulhi per sample, no
// saturation (the quotient is below |x|) ...
__device__ __forceinline__ u32 gain2_below(u32 w, u32 klo, u32 khi, u32 &out)
{
    const u32 sg = pk_sign(w);
    const u32 aw = pk_sub(w ^ sg, sg);
    const u32 q0 = __umulhi(aw & 0xffffu, klo);              // floor(|x|*gain/scale): the mi = 0 case (StreamParam)
    const u32 q1 = __umulhi(aw >> 16, khi);
    const u32 qw = __builtin_amdgcn_perm(q1, q0, 0x05040100u);   // both below 2^15: low halves side by side
    out = pk_sub(qw ^ sg, sg);
    return qw;
}
// ... and no gain at all: the magnitudes are the samples' own (32768 for -32768, as abs() in int gives)
__device__ __forceinline__ u32 gain2_identity(u32 w, u32 &out)
{
    const u32 sg = pk_sign(w);
    out = w;
    return pk_sub(w ^ sg, sg);
}

// sum of squares with as few 64-bit additions as exactness allows: three squares
// (each <= 2^30) fit a u32
struct PowAcc {
    u64 total;
    u32 part;
    u32 n;
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

// squares of the 16-bit halves of packed magnitudes, added into a u32 (three of them fit: 3 * 2^30)
__device__ __forceinline__ u32 sq_lo0(u32 pair)
{
    u32 r;
    asm("v_mad_u32_u16 %0, %1, %1, 0 op_sel:[0,0,0,0]" : "=v"(r) : "v"(pair));
    return r;
}
__device__ __forceinline__ u32 sq_hi0(u32 pair)
{
    u32 r;
    asm("v_mad_u32_u16 %0, %1, %1, 0 op_sel:[1,1,0,0]" : "=v"(r) : "v"(pair));
    return r;
}
__device__ __forceinline__ u32 sq_lo(u32 pair, u32 acc)
{
    asm("v_mad_u32_u16 %0, %1, %1, %0 op_sel:[0,0,0,0]" : "+v"(acc) : "v"(pair));
    return acc;
}
__device__ __forceinline__ u32 sq_hi(u32 pair, u32 acc)
{
    asm("v_mad_u32_u16 %0, %1, %1, %0 op_sel:[1,1,0,0]" : "+v"(acc) : "v"(pair));
    return acc;
}

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

// the tile of the hot kernel (and of the plain-copy ceilings that mirror its access shape)
constexpr u32 TILE_U = 4;                        // 16-byte vectors per lane when PCM is written
constexpr u32 TILE_VEC = 64 * TILE_U;            // vectors per wave: 4 KiB of PCM

}  // namespace cmhip
#endif
