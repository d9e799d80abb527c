// k_eq.hip -- the pipelined equaliser kernel (BASELINE config 3) and its launcher.
#include "cmhip_device.h"
#include <type_traits>
// 0 never, 1 always, 2 (the product) when the launch has an int16 result: see rec_step
#ifndef CMHIP_EQ_R_INTERLEAVE
#define CMHIP_EQ_R_INTERLEAVE 2
#endif

#include <mutex>

namespace cmhip {

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
// through double-buffered LDS tiles (F_k of block b is written in step b+2k, Y_k in step b+2k+1,
// the block leaves in step b+2*NSEC):
//
//   T-in waves (lane = row x 8-frame chunk; time-parallel, G/8 of them):
//     global load (two blocks ahead) -> gain -> float -> feed-forward of section 0 -> F_0
//   T-ff waves (same lane shape, G/2 rows each in passes of 8 rows -- which rows: see f_r below):
//     Y_k-1 -> feed-forward of section k -> F_k               (k = 1 .. NSEC-1)
//     x[t-1], x[t-2] of a chunk come from the neighbouring lane by DPP (row_shr:1), those
//     of a block's first chunk from the last chunk of the previous step (row_shl:7).
//   R waves (lane = section x row; 64/G sections side by side, sequential in time):
//     F_k (16 ds_read_b128) -> the two dependent FMAs per sample -> Y_k (16 ds_write_b128)
//   S waves (four; lane = row x 4-frame piece): Y_last -> coalesced non-temporal global stores
//     (float planes), and for an int16 result / VU window the conversion and the window of it.
//
// Measured on MI355X (tools/ubench_chain.hip, ubench_lds*.hip): a wave alone issues one
// VALU op per ~4.3 clk, a dependent one after ~8; ds_read_b128 costs a wave ~5-10 clk to
// issue, ds_write_b128 ~24 (50 when four waves write at once).  So the only waves that are
// long per step by themselves are the R waves (16 reads, 128 FMAs, 16 writes); everything
// without a recurrence is spread over T lanes, where a 64-frame row costs 2 reads + 2 writes.
// Rows are 68 floats (16-byte aligned, lane-per-row b128 access without bank conflicts).

#ifndef CMHIP_EQ_ABL
#define CMHIP_EQ_ABL 0            // `make abl`: timing-only builds with one part of the pipeline cut out
#endif
// Sensitivity builds (`make variant NAME=pad_r DEFS=-DCMHIP_EQ_PAD_R=32`, tools/ab_two_libs.py): N extra
// VALU instructions per step in the waves of one role.  What the launch time gains per padded
// instruction says which role the step waits for.  Never defined in the product.
#ifndef CMHIP_EQ_PAD_R
#define CMHIP_EQ_PAD_R 0
#endif
#ifndef CMHIP_EQ_PAD_TIN
#define CMHIP_EQ_PAD_TIN 0
#endif
#ifndef CMHIP_EQ_PAD_TFF
#define CMHIP_EQ_PAD_TFF 0
#endif
#ifndef CMHIP_EQ_PAD_S
#define CMHIP_EQ_PAD_S 0
#endif
template <int N>
__device__ __forceinline__ void eq_pad()
{
    if constexpr (N > 0) {
        u32 t = 0;
#pragma unroll
        for (int i = 0; i < N; i++)
            asm volatile("v_add_u32 %0, %0, %0" : "+v"(t));
    }
}

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
// T-ff wave (feed-forward of the later sections, 16 rows) ~850, an S wave 100-350.  The order
// below gives every SIMD one long wave, one T-in wave and one S wave:
//   {T-ff1, T-in0, S0}  {R0, T-in1, S1}  {R1, T-in2, S2}  {T-ff0, T-in3, S3}
// (measured against eight-wave layouts with the store work on the R / T-ff waves: 0.91 ms
// against 0.97-1.03 ms on config 3, 0.96 against 1.49 ms with int16 + VU outputs).
enum : u32 { EQ_R = 0x00, EQ_TIN = 0x10, EQ_TFF = 0x20, EQ_S = 0x30 };
constexpr u32 EQ_NSW = 4;                        // S waves

template <int NSEC, int G>
constexpr u32 eq_waves()
{
    constexpr u32 nrw = (NSEC + 64 / G - 1) / (64 / G);
    return nrw + G / 8 + (NSEC > 1 ? 2 : 0) + EQ_NSW;
}

template <int NRW, int NTF>
__device__ __forceinline__ u32 eq_role(u32 wave)
{
    if constexpr (NTF == 2 && NRW == 2) {        // three or four sections
        constexpr unsigned char t[12] = {EQ_TFF | 1, EQ_R | 0, EQ_R | 1, EQ_TFF | 0, EQ_TIN | 0, EQ_TIN | 1,
                                         EQ_TIN | 2, EQ_TIN | 3, EQ_S | 0, EQ_S | 1, EQ_S | 2, EQ_S | 3};
        return t[wave];
    } else if constexpr (NTF == 2) {             // two sections: one R wave
        constexpr unsigned char t[11] = {EQ_R | 0, EQ_TIN | 0, EQ_TIN | 2, EQ_TIN | 3, EQ_S | 0, EQ_TIN | 1,
                                         EQ_TFF | 0, EQ_TFF | 1, EQ_S | 1, EQ_S | 2, EQ_S | 3};
        return t[wave];
    } else {                                     // one section: no T-ff waves
        constexpr unsigned char t[9] = {EQ_R | 0, EQ_TIN | 0, EQ_TIN | 1, EQ_TIN | 2, EQ_S | 0, EQ_TIN | 3,
                                        EQ_S | 1, EQ_S | 2, EQ_S | 3};
        return t[wave];
    }
}

// CH = 0: LDS behind the tiles.  A T-in wave's eight rows touch at most 8 / C + 2 streams; a block of a stream is
// 64 frames x C channels = 8 C vectors of 16 bytes: at most 256 vectors (C = 16: two streams) per wave.
constexpr u32 EQ_RAWIN = 256 * 16;                // bytes of one T-in wave's raw block
constexpr u32 EQ_STAGE_OUT = 32 * 64 * 2;         // bytes of one slot of the staged int16 result (32 rows x 64 frames)

template <int NSEC, int G, int CH>
__global__ __launch_bounds__((eq_waves<NSEC, G>() * 64))
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
    constexpr u32 HOP = 2;                        // steps from F_k to F_k+1
    constexpr u32 NSW = EQ_NSW;
    // NBUF buffers x 2 slots x EP_TILE floats; for CH = 0 behind them the T-in waves' raw blocks (EQ_RAWIN bytes
    // each) and two slots of EQ_STAGE_OUT bytes for the interleaved int16 result (see below).  The rows' frame
    // counts sit in the padding of the first tile's rows (a row is 64 + 4 floats; nobody writes the four).
    extern __shared__ float lds[];
    auto nfr_at = [&](u32 r) -> u32 & { return reinterpret_cast<u32 *>(lds)[r * EP_ROW + EP_TB]; };
    constexpr bool STAGE_IN = CH == 0;            // T-in: the block's interleaved PCM by 16-byte loads, through LDS
    unsigned char *rawin = reinterpret_cast<unsigned char *>(lds + NBUF * 2 * EP_TILE);
    unsigned char *outstage = rawin + 4u * EQ_RAWIN;
    // The interleaved int16 result of a many-channel stream leaves through LDS as well: the S lanes hold four
    // frames of ONE channel each, and stored as they are that is a 2-byte store every 2C bytes -- 46 % of a
    // 5.1 launch were the S waves' stores (round 4, profiles/r04_eq_multichannel_stamps.txt).  They put their
    // samples where they belong in a copy of the block's frames in LDS, and one step later the same waves send
    // that copy out in whole 16-byte vectors.  Only streams whose rows ALL lie in this workgroup go that way (a
    // vector holds samples of every channel); the one or two streams a workgroup shares with its neighbours keep
    // the 2-byte stores -- rows are dealt out over the whole batch, 32 to a workgroup, so that 1365 x 6 channels
    // are 256 workgroups and not 273, which would be a second round on 256 CUs.
    const bool stage_out = CH == 0 && a.out != nullptr;
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const u32 C = MONO ? 1u : STEREO ? 2u : a.channels;
    // Rows are numbered stream * C + channel over the whole batch and a workgroup takes G
    // consecutive ones (rows are independent, so a stream's channels may sit in two workgroups:
    // no idle rows, and e.g. 1365 x 6 channels are 256 workgroups, not 273).
    // (In place with channel maps a row may read what another row of its stream writes: then a
    // stream's rows must stay in one workgroup, where the barriers order them -- whole_streams.)
    const u32 SPG = G / C;
    const u32 row0_global = a.whole_streams ? blockIdx.x * SPG * C : blockIdx.x * G;
    const u32 rows_here = a.whole_streams ? SPG * C : (u32)G;
    auto row_stream = [&](u32 r, u32 &ch) -> u32 {
        const u32 gr = row0_global + r;
        const u32 q = MONO ? gr : STEREO ? gr >> 1 : gr / C;
        ch = MONO ? 0u : gr - q * C;
        return r < rows_here ? q : 0xffffffffu;           // (q >= streams: past the end of the batch)
    };

    const u32 role = eq_role<(int)NRW, (int)NTF>(wave);
    const bool is_rec = (role & 0xf0u) == EQ_R;
    const bool is_tin = (role & 0xf0u) == EQ_TIN;
    const bool is_tff = (role & 0xf0u) == EQ_TFF;
    // store work: the G / 4 row slots (four rows each) of Y_last, shared out among the S waves
    constexpr u32 NSLOT = G / 4;
    constexpr u32 NSL = NSLOT / NSW;                      // slots of one S wave
    u32 s_first = 0, s_cnt = 0;
    if ((role & 0xf0u) == EQ_S) {
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
        nfr_at(lane) = my_nfr;
    u32 nmax = my_nfr;
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1)
        nmax = max(nmax, (u32)__shfl_xor((int)nmax, o, 64));
    const u32 nblocks = (nmax + EP_TB - 1) / EP_TB;
    const u32 nsteps = nblocks + HOP * NSEC + (stage_out ? 1u : 0u);     // (the staged result leaves a step later)
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
    u32 l_mi = 0, l_mf = 0, l_n = 0, l_m = 0;                // (gain: integer part and fraction, StreamParam)
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
        l_mi = a.param[l_s].mi[l_ch];
        l_mf = a.param[l_s].mf[l_ch];
        l_m = MONO ? 0u : a.param[l_s].chmap[l_ch];       // the input channel this row reads
        l_n = nfr_at(l_r);
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
        // Which rows the eight lane groups of a pass take decides the bank conflicts of their b128 reads:
        // the hardware serves lanes {0-3,12-15,20-27}, {4-11,16-19,28-31} (and the same + 32) together,
        // i.e. chunks 0-3 of the first and fourth group with chunks 4-7 of the second and third.  With
        // rows r, r+16, r+17, r+1 in those four groups the sixteen 16-byte columns are all different
        // (rows 68 floats apart: a row shifts the columns by one); with r, r+1, r+2, r+3 two collide.
        {
            const u32 q = lane / 8u, blk = 2u * tw + p, sub = q & 3u;
            f_r[p] = 4u * blk + 2u * (q >> 2) + (sub >> 1) + ((sub == 1u || sub == 2u) ? 16u : 0u);
        }
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

    // gain disabled (scale 0, or every gain equal to the scale) is stored as integer part 1, fraction 0: x -> x
    const bool gain_off = __all(l_mi == 1u && l_mf == 0u);

    // The T waves keep two blocks of PCM in flight: a block's HBM latency is hidden behind
    // two pipeline steps.  The load is unconditional (address clamped into the stream's own
    // row, which is a multiple of 8 samples long) and nothing else in a T wave touches
    // global memory inside the loop, so the compiler can wait with vmcnt(1) for the older
    // block instead of draining the queue; the stores have a wave of their own.
    const int16_t *l_src = a.in + (u64)l_s * a.stride;
    // (EQ batches have rows of whole 8-frame chunks, so a chunk with a valid frame is never clamped)
    const u32 l_last = (u32)a.stride - (MONO ? 8u : STEREO ? 16u : 1u);
    const u32 l_sel = l_m ? 0x07060302u : 0x05040100u;    // STEREO: which halves of two frames make a pair
    // CH = 0 (STAGE_IN): a T-in wave loads the block's interleaved PCM of the streams its eight rows belong to
    // -- tv_n vectors of 16 bytes, lane l takes vectors l, l + 64, l + 128, l + 192 (those past tv_n repeat the
    // last one: the loads stay unconditional) -- and when the block's turn comes passes them through its own 4 KiB
    // of LDS, where every lane picks the eight samples of its row's channel.  (Round 3 gathered them with eight
    // 2-byte global loads per lane: 5.1 float planes waited 350 clk of every step for them and the T-in waves
    // beside the R waves were last at the barrier in every step, profiles/r04_eq_multichannel_stamps.txt.)
    u32 tv_n = 0;                                         // vectors of a block of this wave's streams
    const int16_t *tv_src[4] = {a.in, a.in, a.in, a.in};  // per lane and vector: its stream's slot ...
    u32 tv_off[4] = {0, 0, 0, 0};                         // ... and the vector's place in a block, in samples
    u32 l_raw = 0;                                        // byte offset of this lane's first sample in the raw block
    if constexpr (STAGE_IN) {
        if (is_tin) {
            const u32 gr_first = row0_global + 8u * tw;
            const u32 gr_end = min(min(gr_first + 8u, row0_global + rows_here), a.streams * C);    // one past the last row
            if (gr_first < gr_end) {
                const u32 ts_lo = gr_first / C, tnst = (gr_end - 1u) / C - ts_lo + 1u;
                tv_n = tnst * 8u * C;                     // <= 256 (EQ_RAWIN)
#pragma unroll
                for (u32 u = 0; u < 4; u++) {
                    const u32 v = min(64u * u + lane, tv_n - 1u);
                    const u32 si = v / (8u * C), vv = v - si * 8u * C;
                    tv_src[u] = a.in + (u64)(ts_lo + si) * a.stride;
                    tv_off[u] = vv * 8u;
                }
                if (l_live)
                    l_raw = (((l_s - ts_lo) * EP_TB + l_t8) * C + l_m) * 2u;
            }
        }
    }
    struct Pcm { u32x4 a, b, c, d; };                     // a chunk in flight (b: second half, STEREO; c, d: STAGE_IN)
    auto fetch = [&](u32 b) -> Pcm {
        const u32 f0 = b * EP_TB + l_t8;
        Pcm r = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        if (CMHIP_EQ_ABL & 2) {
            r.a = u32x4{f0, f0 * 3u, f0 * 5u, f0 * 7u};
        } else if constexpr (MONO) {
            r.a = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(l_src + min(f0, l_last)));
        } else if constexpr (STEREO) {                    // 8 frames x (L, R): 32 bytes
            const u32x4 *p = reinterpret_cast<const u32x4 *>(l_src + min(2u * f0, l_last));
            r.a = __builtin_nontemporal_load(p);
            r.b = __builtin_nontemporal_load(p + 1);
        } else {
            // (clamped into the stream's own slot, which is a whole number of vectors long: the blocks fetched
            // ahead of a stream's end load its last vector again)
            const u32 blk = b * EP_TB * C, last8 = (u32)a.stride - 8u;
            r.a = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(tv_src[0] + min(blk + tv_off[0], last8)));
            r.b = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(tv_src[1] + min(blk + tv_off[1], last8)));
            r.c = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(tv_src[2] + min(blk + tv_off[2], last8)));
            r.d = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(tv_src[3] + min(blk + tv_off[3], last8)));
        }
        return r;
    };
    // Section 0's x1 / x2 for the next launch are the last two samples of the row after the gain.
    // On mono rows one lane per row fetches them here, before the loop, straight from the input (still intact also
    // when the result is written in place: the first store is many barriers away), instead of every
    // T lane watching for the end of its stream in every step.
    float keep1 = 0.f, keep2 = 0.f;
    u32 keep_n = 0;                                       // bit 0: keep1 is to be written, bit 1: keep2
    if (MONO && l_live && l_c == 0u && l_n >= 1u) {
        keep_n = 3u;
        auto gain_one = [&](const u32 frame) -> float {
            const int xs = (int)l_src[MONO ? frame : frame * C + l_m];
            const u32 ax = (u32)(xs < 0 ? -xs : xs);
            const u32 qq = __umul24(ax, l_mi) + __umulhi(ax, l_mf);
            const float m = fminf((float)qq, xs < 0 ? 32768.0f : 32767.0f);
            return (xs < 0 ? -m : m) * (1.0f / 32768.0f);
        };
        keep1 = gain_one(l_n - 1u);
        keep2 = l_n >= 2u ? gain_one(l_n - 2u) : a.state[l_sidx].s[0][0];
    }
    Pcm wa = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}}, wb = wa;       // blocks of even / odd steps
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
    // which wave reaches the barrier last (sampled every 16th step: the look itself delays wave 0's next
    // step), and how long the first one waits for it.  One 256-byte block: the tiles in dynamic LDS behind
    // it keep their 16-byte alignment.
    struct alignas(256) StampLds { u32 arr[2][16]; u32 last[16]; unsigned long long skew; u32 pad[14]; };
    static_assert(sizeof(StampLds) == 256, "");
    __shared__ StampLds st_lds;
    u32 st_n = 0;
    if (threadIdx.x < 16)
        st_lds.last[threadIdx.x] = 0;
    if (threadIdx.x == 0)
        st_lds.skew = 0;
    __syncthreads();
#endif
    // Schedule: F_k of block b is written in step b+2k, Y_k in step b+2k+1, the block leaves in
    // step b+2*NSEC.  Every buffer is read one step after it was written, so two slots do.
    // (Loading an R wave's next row into registers a step ahead was worth 7 % with eight waves
    // per workgroup; with twelve it needs more registers than three waves per SIMD leave.)
    auto rec_step = [&](const u32 step) {
        eq_pad<CMHIP_EQ_PAD_R>();
        if (!(CMHIP_EQ_ABL & 32)) {
            const u32 first = HOP * sec + HOP - 1u;       // step in which block 0 is worked on
            const u32 b = step - first;
            if (has_sec && step >= first && b < nblocks) {
                float4 *out = reinterpret_cast<float4 *>(
                    lds + ((2u * sec + 1u) * 2u + (b & 1u)) * EP_TILE + row * EP_ROW);
                const float4 *in = reinterpret_cast<const float4 *>(
                    lds + ((2u * sec) * 2u + (b & 1u)) * EP_TILE + row * EP_ROW);
                float4 v[EP_TB / 4];
#pragma unroll
                for (u32 t = 0; t < EP_TB / 4; t++)      // whole row first: 16 LDS reads in flight
                    v[t] = in[t];
                const u32 done = b * EP_TB;
                const u32 cnt = my_nfr > done ? min(my_nfr - done, EP_TB) : 0u;
                if (CMHIP_EQ_ABL & 4) {
#pragma unroll
                    for (u32 t = 0; t < EP_TB / 4; t++)
                        out[t] = v[t];
                } else if (__all(cnt == EP_TB || cnt == 0u)) {
                    // (every row of the wave has the whole block or nothing of it: rows past the end of the batch,
                    // streams that ended in an earlier block.  Those run the same instructions on whatever their tile
                    // holds -- nothing of it is ever stored -- and get their history back afterwards.  Round 3 sent
                    // the whole wave down the sample-by-sample path below as soon as ONE row was idle: a batch whose
                    // rows are not a multiple of 32 -- 1365 x 6 -- had one workgroup that took 1.4 x as long as the
                    // others for the whole launch, and the launch waited for it.)
                    const float keep_h1 = h1, keep_h2 = h2;
                    // Where the stores go: left alone the scheduler moves all sixteen to the end of the row.  Kept
                    // behind their four samples each (a scheduling barrier per store) the runs that produce an
                    // int16 result take 2.3-3.6 % less and config 3's float planes 0.8 % more (A/B in one process,
                    // profiles/r03_eq_store_order_ab.txt) -- so the order follows the outputs.  Either way a store
                    // holds the wave for its 25-50 clk: spacing them does not hide that (NOTES_r03).
                    auto row = [&](auto spaced) {
#pragma unroll
                        for (u32 t = 0; t < EP_TB / 4; t++) {
                            float4 y = v[t];
                            if (!(CMHIP_EQ_ABL & 256) || (t & 1u) == 0u) {      // (256: timing only, half the FMAs)
                                y.x = __builtin_fmaf(d1, h1, __builtin_fmaf(d2, h2, v[t].x));
                                y.y = __builtin_fmaf(d1, y.x, __builtin_fmaf(d2, h1, v[t].y));
                                y.z = __builtin_fmaf(d1, y.y, __builtin_fmaf(d2, y.x, v[t].z));
                                y.w = __builtin_fmaf(d1, y.z, __builtin_fmaf(d2, y.y, v[t].w));
                                h2 = y.z;
                                h1 = y.w;
                            }
                            if (!(CMHIP_EQ_ABL & 8) || t == 0)
                                out[t] = y;
                            if constexpr (decltype(spaced)::value)
                                __builtin_amdgcn_sched_barrier(0);
                        }
                    };
                    if (CMHIP_EQ_R_INTERLEAVE == 1 || (CMHIP_EQ_R_INTERLEAVE == 2 && a.out != nullptr))
                        row(std::true_type{});
                    else
                        row(std::false_type{});
                    h1 = cnt ? h1 : keep_h1;
                    h2 = cnt ? h2 : keep_h2;
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
    auto tin_step = [&](Pcm &wcur, const u32 step) {
        eq_pad<CMHIP_EQ_PAD_TIN>();
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
                if constexpr (STAGE_IN) {
                    // the raw block into this wave's own LDS area, then this lane's eight samples out of it
                    // (one wave: its LDS operations execute in the order issued, no barrier)
                    unsigned char *rw = rawin + tw * EQ_RAWIN;
                    if (lane < tv_n)
                        *reinterpret_cast<u32x4 *>(rw + lane * 16u) = wcur.a;
                    if (64u + lane < tv_n)
                        *reinterpret_cast<u32x4 *>(rw + (64u + lane) * 16u) = wcur.b;
                    if (128u + lane < tv_n)
                        *reinterpret_cast<u32x4 *>(rw + (128u + lane) * 16u) = wcur.c;
                    if (192u + lane < tv_n)
                        *reinterpret_cast<u32x4 *>(rw + (192u + lane) * 16u) = wcur.d;
                    const unsigned char *mine = rw + l_raw;
#pragma unroll
                    for (u32 q = 0; q < 4; q++) {
                        const u32 lo = *reinterpret_cast<const uint16_t *>(mine + (2u * q) * 2u * C);
                        const u32 hi = *reinterpret_cast<const uint16_t *>(mine + (2u * q + 1u) * 2u * C);
                        w[q] = lo | (hi << 16);
                    }
                }
                // (These selects are not needed for the results -- nothing beyond a stream's end is ever
                // stored -- and the next block's load could be issued before the wait for this one's data or
                // after the stores below.  Measured, A/B in one process, round 2: without the selects the
                // compiler issues the load first and the launch takes 6-9 % longer; with the load after the
                // stores 2-5 % longer.  They stay where round 1 left them.)
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
                        const u32 x0 = aw & 0xffffu, x1 = aw >> 16;
                        const float m0 = (float)(__umul24(x0, l_mi) + __umulhi(x0, l_mf));
                        const float m1 = (float)(__umul24(x1, l_mi) + __umulhi(x1, l_mf));
                        const u32 b0 = (__builtin_bit_cast(u32, m0) & 0x7fffffffu) | ((w[q] << 16) & 0x80000000u);
                        const u32 b1 = (__builtin_bit_cast(u32, m1) & 0x7fffffffu) | (w[q] & 0x80000000u);
                        x[2 * q] = __builtin_amdgcn_fmed3f(__builtin_bit_cast(float, b0), -32768.0f, 32767.0f);
                        x[2 * q + 1] = __builtin_amdgcn_fmed3f(__builtin_bit_cast(float, b1), -32768.0f, 32767.0f);
                    }
                }
                // (Stereo and many-channel rows watch for the end of their stream inside the loop, as in round
                // 1: the last real samples pass through exactly one lane each.  The mono form fetches them
                // before the loop -- the same loop without this block ran 1.7 % faster on mono rows and
                // 4.7 % slower on stereo ones, A/B in one process.)
                if constexpr (!MONO) {
                    const u32 bf = b * EP_TB;
                    if (l_live && l_n > bf && l_n <= bf + EP_TB) {
                        const u32 e1 = l_n - 1u - bf;                     // last sample, block relative
                        if ((e1 >> 3) == l_c) {
                            float val = x[0];
#pragma unroll
                            for (u32 j = 1; j < 8; j++)
                                val = (e1 & 7u) == j ? x[j] : val;
                            keep1 = val * (1.0f / 32768.0f);
                            keep_n |= 1u;
                        }
                        if (e1 >= 1u) {
                            const u32 e2 = e1 - 1u;
                            if ((e2 >> 3) == l_c) {
                                float val = x[0];
#pragma unroll
                                for (u32 j = 1; j < 8; j++)
                                    val = (e2 & 7u) == j ? x[j] : val;
                                keep2 = val * (1.0f / 32768.0f);
                                keep_n |= 2u;
                            }
                        } else if (l_c == 7u) {
                            keep2 = sx1[0][0] * (1.0f / 32768.0f);        // the sample before this block
                            keep_n |= 2u;
                        }
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
        eq_pad<CMHIP_EQ_PAD_TFF>();
        if (!(CMHIP_EQ_ABL & (128 | 16))) {
            // reads of every pass first, then arithmetic: one LDS latency per step
            constexpr u32 PG = PASSES;
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
    // (rows of a slot: r, r+16, r+1, r+17 for the four lane groups of 16 -- the two groups a b128
    // read serves together then sit 16 rows apart and their columns do not collide)
    auto s_row = [&](u32 i) -> u32 {
        const u32 g4 = lane / SPR;                        // lane group of 16
        return 2u * (s_first + i) + (g4 >> 1) + ((g4 & 1u) ? 16u : 0u);
    };
    u64 vpw[NSL], vky[NSL], vbase[NSL];
    u32 vmag[NSL], vq01[NSL], vq23[NSL], vfr[NSL];        // running peak: magnitude, the four results it was among, their first frame
    u32 v_stream[NSL], v_ch[NSL];                         // stream (or none) and channel of the slot's row
#pragma unroll
    for (u32 i = 0; i < NSL; i++) {
        vpw[i] = vky[i] = vbase[i] = 0;
        vmag[i] = vq01[i] = vq23[i] = vfr[i] = 0;
        const u32 r = s_row(i);
        v_stream[i] = row_stream(r, v_ch[i]);
        if (v_stream[i] >= a.streams || i >= s_cnt)
            v_stream[i] = 0xffffffffu;
        if (is_store && a.vu && v_stream[i] != 0xffffffffu)
            vbase[i] = a.vu[v_stream[i]].samples[a.parity];
    }
    // stage_out: where this lane's four frames of row slot i go in a staged block ([stream of the workgroup][frame]
    // [channel] int16, as the frames lie in global memory), and which 16-byte vector of a staged block this lane
    // sends out a step later (the 4 x 64 S lanes cover the 256 vectors a workgroup's 32 rows x 64 frames make)
    u32 so_base[NSL];                                     // 0xffffffff: the row's stream is shared with a neighbour
    int16_t *co_ptr = nullptr;                            // nullptr: no vector for this lane
    u32 co_n = 0, co_f0 = 0, co_f1 = 0, co_vec = 0, co_s8 = 0;
#pragma unroll
    for (u32 i = 0; i < NSL; i++)
        so_base[i] = 0xffffffffu;
    if (stage_out && is_store) {
        // whole streams of this workgroup: s_fw .. s_we - 1
        const u32 s_fw = (row0_global + C - 1u) / C;
        const u32 s_we = min((row0_global + rows_here) / C, a.streams);
        const u32 t4 = (lane % SPR) * 4u;
#pragma unroll
        for (u32 i = 0; i < NSL; i++)
            if (v_stream[i] != 0xffffffffu && v_stream[i] >= s_fw && v_stream[i] < s_we)
                so_base[i] = (((v_stream[i] - s_fw) * EP_TB + t4) * C + v_ch[i]) * 2u;
        co_vec = (role & 15u) * 64u + lane;
        const u32 si = co_vec / (8u * C), vv = co_vec - si * 8u * C;
        if (s_fw + si < s_we) {
            co_ptr = a.out + (u64)(s_fw + si) * a.stride + vv * 8u;
            co_n = a.nframes ? a.nframes[s_fw + si] : a.frames;
            co_s8 = vv * 8u;                              // the vector's first sample inside a block
            co_f0 = co_s8 / C;                            // first and last frame (inside a block) the vector touches
            co_f1 = (co_s8 + 7u) / C;
        }
    }
    auto s_step = [&](const u32 step) {
        eq_pad<CMHIP_EQ_PAD_S>();
        if (stage_out && !(CMHIP_EQ_ABL & 2048)) {                       // (2048: timing only, nothing copied out)
            // the block the S waves staged in the step before leaves in whole vectors (the barrier between the two
            // steps has made every wave's samples visible; the other slot takes this step's block meanwhile)
            const u32 bo = step - (HOP * NSEC + 1u);
            if (step >= HOP * NSEC + 1u && bo < nblocks && co_ptr) {
                const u32x4 v = *reinterpret_cast<const u32x4 *>(outstage + (bo & 1u) * EQ_STAGE_OUT + co_vec * 16u);
                int16_t *dst = co_ptr + (u64)bo * EP_TB * C;
                const u32 fb = bo * EP_TB;
                if (fb + co_f1 < co_n) {
                    __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(dst));
                } else if (fb + co_f0 < co_n) {           // the stream ends inside this vector: sample by sample
                    const u32 e[4] = {v.x, v.y, v.z, v.w};
                    for (u32 j = 0; j < 8; j++)
                        if (fb + (co_s8 + j) / C < co_n)
                            dst[j] = (int16_t)(e[j >> 1] >> (16u * (j & 1u)));
                }
            }
        }

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
                    const u32 r = min(s_row(i), (u32)G - 1u);   // (slots past s_cnt are skipped below)
                    nin[i] = nfr_at(r);
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
                        // float -> int16 as oracle_f32_to_i16: y * 32768, round to nearest even, then the
                        // hardware's saturating conversions do the rest (v_cvt_i32_f32: NaN -> 0, out of
                        // range -> INT_MIN / INT_MAX; the packing below saturates to int16)
                        int q[4];
#pragma unroll
                        for (u32 j = 0; j < 4; j++) {
                            const float r = __builtin_rintf(e[j] * 32768.0f);
                            asm("v_cvt_i32_f32 %0, %1" : "=v"(q[j]) : "v"(r));
                        }
                        // (v_cvt_pk_i16_i32 saturates: the clamp and the packing of a sample pair in one)
                        const u32 p01 = __builtin_bit_cast(u32, __builtin_amdgcn_cvt_pk_i16(q[0], q[1]));
                        const u32 p23 = __builtin_bit_cast(u32, __builtin_amdgcn_cvt_pk_i16(q[2], q[3]));
                        const bool whole = __all(f0 + 4u <= n);     // no stream ends inside these
                        if (a.out) {
                            if constexpr (MONO) {
                                int16_t *d16 = a.out + (u64)vs_ * a.stride + f0;
                                if (f0 + 4u <= n) {
                                    typedef u32 u32x2 __attribute__((ext_vector_type(2)));
                                    const u32x2 pk = {p01, p23};
                                    __builtin_nontemporal_store(pk, reinterpret_cast<u32x2 *>(d16));
                                } else if (f0 < n) {
                                    for (u32 j = 0; j < n - f0; j++)
                                        d16[j] = (int16_t)((j < 2u ? p01 : p23) >> (16u * (j & 1u)));
                                }
                            } else if constexpr (STEREO) {
                                // The two rows of a stream are 32 lanes apart (s_row): swap halves with the
                                // partner (v_permlane32_swap) so that the left row's lanes hold
                                // frames f0, f0+1 of both channels and the right row's lanes frames
                                // f0+2, f0+3 -- whole interleaved frames, 8 bytes per lane.
                                typedef u32 u32x2 __attribute__((ext_vector_type(2)));
                                const u32x2 sw2 = __builtin_amdgcn_permlane32_swap(p01, p23, false, false);
                                const u32 left = sw2.x, right = sw2.y;   // this frame pair: channel 0, channel 1
                                const u32 d0 = __builtin_amdgcn_perm(right, left, 0x05040100u);
                                const u32 d1 = __builtin_amdgcn_perm(right, left, 0x07060302u);
                                const u32 ff = f0 + 2u * vc_;            // first of this lane's two frames
                                u32 *d32 = reinterpret_cast<u32 *>(a.out + (u64)vs_ * a.stride) + ff;
                                if (ff + 2u <= n) {
                                    const u32x2 pk = {d0, d1};
                                    __builtin_nontemporal_store(pk, reinterpret_cast<u32x2 *>(d32));
                                } else if (ff < n) {
                                    d32[0] = d0;
                                }
                            } else if ((C & 1u) == 0u) {
                                // An even channel count: the rows of a slot's lanes L and L + 32 are channels 2k and
                                // 2k + 1 of one stream (s_row; rows are dealt out from even numbers), so the two swap
                                // halves as the stereo form does -- L keeps frames f0, f0 + 1 of both channels, L + 32
                                // frames f0 + 2, f0 + 3 -- and every sample pair is one aligned dword of a frame:
                                // two 4-byte stores per lane instead of four 2-byte ones, staged or not.
                                typedef u32 u32x2 __attribute__((ext_vector_type(2)));
                                const u32x2 sw2 = __builtin_amdgcn_permlane32_swap(p01, p23, false, false);
                                const u32 d0 = __builtin_amdgcn_perm(sw2.y, sw2.x, 0x05040100u);
                                const u32 d1 = __builtin_amdgcn_perm(sw2.y, sw2.x, 0x07060302u);
                                const u32 hi = lane >> 5;                 // this lane's row is the odd channel of the pair
                                if (CMHIP_EQ_ABL & 1024) {                 // (1024: timing only, the result goes nowhere)
                                } else if (so_base[i] != 0xffffffffu) {
                                    unsigned char *st32 = outstage + (b & 1u) * EQ_STAGE_OUT + so_base[i] - 2u * hi + hi * 4u * C;
                                    *reinterpret_cast<u32 *>(st32) = d0;
                                    *reinterpret_cast<u32 *>(st32 + 2u * C) = d1;
                                } else {
                                    const u32 ff = f0 + 2u * hi;
                                    int16_t *d16 = a.out + (u64)vs_ * a.stride + (u64)ff * C + (vc_ - hi);
                                    if (ff < n)
                                        *reinterpret_cast<u32 *>(d16) = d0;
                                    if (ff + 1u < n)
                                        *reinterpret_cast<u32 *>(d16 + C) = d1;
                                }
                            } else if (so_base[i] != 0xffffffffu) {       // interleaved result: into the staged block
                                unsigned char *st16 = outstage + (b & 1u) * EQ_STAGE_OUT + so_base[i];
#pragma unroll
                                for (u32 j = 0; j < 4; j++)
                                    *reinterpret_cast<int16_t *>(st16 + j * 2u * C) =
                                        (int16_t)((j < 2u ? p01 : p23) >> (16u * (j & 1u)));
                            } else {                      // ... or, a stream shared with a neighbour, this row's channel
                                int16_t *d16 = a.out + (u64)vs_ * a.stride + (u64)f0 * C + vc_;
#pragma unroll
                                for (u32 j = 0; j < 4; j++)
                                    if (f0 + j < n)
                                        d16[j * C] = (int16_t)((j < 2u ? p01 : p23) >> (16u * (j & 1u)));
                            }
                        }
                        if (a.vu && !(CMHIP_EQ_ABL & 512)) {            // (512: timing only, no window)
                            // The window of the int16 result on packed pairs (as the block kernels do): the
                            // magnitudes of four samples in six instructions, their squares in one chain of
                            // three v_mad_u32_u16 and a fourth, and one comparison of the four's maximum
                            // with the lane's running peak -- a lane whose peak improves keeps the four
                            // results and their first frame; which of them came first, and its sign, is
                            // looked up once, after the loop.  (The samples come in time order, so "strictly
                            // greater" keeps the first of equals.)
                            u32 v01 = p01, v23 = p23;     // the results that count: all four, unless a stream ends here
                            if (!whole) {                 // (uniform branch; what lies beyond a stream's end counts nothing)
                                v01 &= (f0 + 0u < n ? 0xffffu : 0u) | (f0 + 1u < n ? 0xffff0000u : 0u);
                                v23 &= (f0 + 2u < n ? 0xffffu : 0u) | (f0 + 3u < n ? 0xffff0000u : 0u);
                            }
                            const u32 s01 = pk_sign(v01), s23 = pk_sign(v23);
                            const u32 m01 = pk_sub(v01 ^ s01, s01), m23 = pk_sub(v23 ^ s23, s23);
                            vpw[i] += sq_lo(m23, sq_hi(m01, sq_lo0(m01)));
                            vpw[i] += sq_hi0(m23);
                            const u32 mm = pk_max(m01, m23);
                            const u32 m4 = max(mm & 0xffffu, mm >> 16);
                            const bool gt = m4 > vmag[i];
                            vmag[i] = gt ? m4 : vmag[i];
                            vq01[i] = gt ? v01 : vq01[i];
                            vq23[i] = gt ? v23 : vq23[i];
                            vfr[i] = gt ? f0 : vfr[i];
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
        const u64 st_t1 = __builtin_readcyclecounter();                 \
        st_busy += st_t1 - st_t0;                                       \
        if (lane == 0)                                                  \
            st_lds.arr[st_n & 1u][wave] = (u32)st_t1;                   \
        __syncthreads();                                                \
        if (threadIdx.x == 0 && (st_n & 15u) == 0) {                    \
            const u32 nw_ = blockDim.x >> 6, ref_ = st_lds.arr[st_n & 1u][0]; \
            int lo_ = 0, hi_ = 0;                                       \
            u32 who_ = 0;                                               \
            for (u32 w_ = 1; w_ < nw_; w_++) {                          \
                const int d_ = (int)(st_lds.arr[st_n & 1u][w_] - ref_); \
                if (d_ > hi_) { hi_ = d_; who_ = w_; }                  \
                if (d_ < lo_) lo_ = d_;                                 \
            }                                                           \
            st_lds.last[who_]++;                                        \
            st_lds.skew += (unsigned long long)(hi_ - lo_);             \
        }                                                               \
        st_n++;                                                         \
    } while (0)
#else
#define EQ_STEP(call) do { call; __syncthreads(); } while (0)
#endif
    // One loop per role (the role never changes, and a loop of its own lets the compiler
    // count a T wave's outstanding loads); every wave passes the same number of barriers.
    const u32 nst2 = (nsteps + 1u) & ~1u;                 // an odd tail step finds nothing to do
    if (is_rec) {
        for (u32 step = 0; step < nst2; step++)
            EQ_STEP(rec_step(step));
    } else if (is_store) {
        for (u32 step = 0; step < nst2; step++)
            EQ_STEP(s_step(step));
    } else if (is_tff) {
        for (u32 step = 0; step < nst2; step++)
            EQ_STEP(tff_step(step));
    } else {
        for (u32 step = 0; step < nst2; step += 2) {
            EQ_STEP(tin_step(wa, step));
            EQ_STEP(tin_step(wb, step + 1));
        }
        if (keep_n & 1u)
            a.state[l_sidx].s[0][0] = keep1;
        if (keep_n & 2u)
            a.state[l_sidx].s[0][1] = keep2;
    }
#undef EQ_STEP

    // VU windows of the int16 result: every wave that did store work holds parts of them
    if (is_store && a.vu) {
        // the 16 lanes of a row hold parts of its window; lane 0 of them is the row's only writer
#pragma unroll
        for (u32 i = 0; i < NSL; i++) {
            {
                u32 first = 0, neg = 0;                   // the first of the four with that magnitude, and its sign
#pragma unroll
                for (u32 j = 4; j-- > 0;) {
                    const int sv = (int)(short)(((j < 2u ? vq01[i] : vq23[i]) >> (16u * (j & 1u))) & 0xffffu);
                    const u32 am = (u32)(sv < 0 ? -sv : sv);
                    if (am == vmag[i]) {
                        first = j;
                        neg = sv < 0 ? 1u : 0u;
                    }
                }
                vky[i] = make_key(vmag[i], vbase[i] + (u64)(vfr[i] + first) * C + v_ch[i], neg);
            }
            u64 pw = vpw[i], ky = vky[i];
#pragma unroll
            for (int o = SPR / 2; o > 0; o >>= 1) {
                pw += (u64)__shfl_xor((long long)pw, o, 64);
                const u64 ok = (u64)__shfl_xor((long long)ky, o, 64);
                ky = ok > ky ? ok : ky;
            }
            const u32 r = s_row(i);
            if (lane % SPR == 0 && v_stream[i] != 0xffffffffu) {
                VuState *vs = a.vu + v_stream[i];
                if (v_ch[i] == 0)
                    vs->samples[a.parity ^ 1u] = vbase[i] + (u64)nfr_at(r) * C;
                vs->power[v_ch[i]] += pw;
                if (ky > vs->key[v_ch[i]])
                    vs->key[v_ch[i]] = ky;
            }
        }
    }
#ifdef CMHIP_EQ_STAMPS
    if (blockIdx.x == 7 && lane == 0 && a.dbg) {     // per-role busy cycles (tools/eq_stamps.py)
        a.dbg[2 * wave] = st_busy;
        a.dbg[2 * wave + 1] = __builtin_readcyclecounter() - st_begin;
        a.dbg[36 + wave] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_ID
        a.dbg[48] = nsteps;
        if (is_tin && tw < 2) {                       // two T-in waves: phases inside a step
            for (int i = 0; i < 3; i++)
                a.dbg[50 + 3 * tw + i] = st_p[i];
        }
        a.dbg[24 + wave] = role;
    }
    __syncthreads();
    if (blockIdx.x == 7 && threadIdx.x == 0 && a.dbg) {
        for (u32 w_ = 0; w_ < 12; w_ += 2)
            a.dbg[56 + w_ / 2] = (u64)st_lds.last[w_] | ((u64)st_lds.last[w_ + 1] << 32);
        a.dbg[62] = st_lds.skew;
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
    done_epilogue(a.done_flag, a.done_seq);
}

template <int NSEC, int G, int CH>
static constexpr size_t eq_pipe_lds_bytes()
{
    return ((size_t)(2 * NSEC) * 2 * G * (64 + 4)) * sizeof(float) + (CH == 0 ? 4u * EQ_RAWIN + 2u * EQ_STAGE_OUT : 0u);
}

template <int NSEC, int G, int CH>
static hipError_t launch_eq_pipe(const EqArgs &a0, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop, bool *flagged)
{
    EqArgs a = a0;
    constexpr size_t lds_bytes = eq_pipe_lds_bytes<NSEC, G, CH>();
    static_assert(lds_bytes <= 160 * 1024, "tiles of a workgroup must fit the LDS");

    const u64 rows = (u64)a.streams * (CH == 1 ? 1u : a.channels);      // one row per stream and channel
    const u32 spg = CH == 1 ? G : G / a.channels;
    const u32 grid = a.whole_streams ? (a.streams + spg - 1) / spg : (u32)((rows + G - 1) / G);
    if (grid != 1u)                                      // completion by flag: one workgroup only (EqArgs::done_flag)
        a.done_flag = nullptr;
    if (flagged)
        *flagged = a.done_flag != nullptr;
    hipExtLaunchKernelGGL((k_eq_pipe<NSEC, G, CH>), dim3(grid), dim3(eq_waves<NSEC, G>() * 64), lds_bytes, st, ev_start, ev_stop, 0, a);
    return hipGetLastError();
}

// 32 rows per workgroup: all 256 CUs at 8192 mono streams, and what the LDS holds for four
// sections (8 and 16 rows with several workgroups per CU measured the same or slower)
template <int NSEC>
static hipError_t launch_eq_pipe_g(const EqArgs &a, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop, bool *flagged)
{
    if (a.channels == 1)
        return launch_eq_pipe<NSEC, 32, 1>(a, st, ev_start, ev_stop, flagged);
    if (a.channels == 2 && a.stride >= 16 && a.stride % 16 == 0)
        return launch_eq_pipe<NSEC, 32, 2>(a, st, ev_start, ev_stop, flagged);
    return launch_eq_pipe<NSEC, 32, 0>(a, st, ev_start, ev_stop, flagged);
}

// The kernels ask for more dynamic LDS than the default limit; the limit is raised per function
// AND per device, so every device a batch lives on gets its own pass (a C host may drive all
// GPUs of a node from one process).  Guarded by a mutex: two threads may create their first
// batches at the same time.
template <int NSEC>
static hipError_t raise_lds_limit()
{
    // (a -DCMHIP_EQ_STAMPS build keeps 256 bytes of static LDS for its stamps: the one variant that fills the
    // whole 160 KiB, four sections on many channels, cannot be launched in that diagnostic build)
#ifdef CMHIP_EQ_STAMPS
    constexpr size_t cap = 160 * 1024 - 256;
#else
    constexpr size_t cap = 160 * 1024;
#endif
    auto bytes = [](size_t want) { return (int)(want < cap ? want : cap); };
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_eq_pipe<NSEC, 32, 1>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, bytes(eq_pipe_lds_bytes<NSEC, 32, 1>()));
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_eq_pipe<NSEC, 32, 2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes(eq_pipe_lds_bytes<NSEC, 32, 2>()));
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_eq_pipe<NSEC, 32, 0>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes(eq_pipe_lds_bytes<NSEC, 32, 0>()));
    return e;
}

hipError_t prepare_eq(int device)
{
    constexpr int MAX_DEV = 64;
    static std::mutex mu;
    static bool done[MAX_DEV];
    if (device < 0 || device >= MAX_DEV)
        return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> g(mu);
    if (done[device])
        return hipSuccess;
    int cur = -1;
    hipError_t e = hipGetDevice(&cur);
    if (e == hipSuccess && cur != device)
        e = hipSetDevice(device);
    if (e == hipSuccess) e = raise_lds_limit<1>();
    if (e == hipSuccess) e = raise_lds_limit<2>();
    if (e == hipSuccess) e = raise_lds_limit<3>();
    if (e == hipSuccess) e = raise_lds_limit<4>();
    if (cur >= 0 && cur != device)
        (void)hipSetDevice(cur);
    if (e == hipSuccess)
        done[device] = true;
    return e;
}

hipError_t launch_eq(const EqArgs &a, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop, bool *flagged)
{
    if (flagged)
        *flagged = false;
    if (a.streams == 0 || a.frames == 0 || !(a.f32 || a.out || a.vu))
        return hipSuccess;
    {
        int dev = -1;                                     // (the engine has made the batch's device current)
        hipError_t e = hipGetDevice(&dev);
        if (e == hipSuccess)
            e = prepare_eq(dev);
        if (e != hipSuccess)
            return e;
    }
    if (a.channels == 0 || a.channels > MAX_CH)
        return hipErrorInvalidValue;
    switch (a.nsec) {                                     // (0 sections: the caller uses launch_run)
    case 1: return launch_eq_pipe_g<1>(a, st, ev_start, ev_stop, flagged);            // the pipelined kernel, whatever is asked
    case 2: return launch_eq_pipe_g<2>(a, st, ev_start, ev_stop, flagged);            // for (float planes, int16, VU of it)
    case 3: return launch_eq_pipe_g<3>(a, st, ev_start, ev_stop, flagged);
    case 4: return launch_eq_pipe_g<4>(a, st, ev_start, ev_stop, flagged);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace cmhip
