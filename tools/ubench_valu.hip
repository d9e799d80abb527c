// ubench_valu.hip -- issue cost of the integer VALU ops the hot kernel can choose from,
// measured on gfx950: one 256-thread block per CU slot, 8 independent chains per lane,
// wall time over all CUs.  Not part of the product.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CHAINS 8
#define REPS 64

#define BODY(NAME, ASM)                                                                   \
    __global__ __launch_bounds__(256) void k_##NAME(uint32_t *out, int iters, uint32_t a, \
                                                     uint32_t b)                          \
    {                                                                                     \
        uint32_t x[CHAINS];                                                               \
        for (int i = 0; i < CHAINS; i++) x[i] = threadIdx.x * 2654435761u + i + a;        \
        for (int it = 0; it < iters; it++) {                                              \
            _Pragma("unroll") for (int r = 0; r < REPS / CHAINS; r++) {                   \
                _Pragma("unroll") for (int i = 0; i < CHAINS; i++) {                      \
                    asm volatile(ASM : "+v"(x[i]) : "v"(b), "s"(a));                      \
                }                                                                         \
            }                                                                             \
        }                                                                                 \
        uint32_t s = 0;                                                                   \
        for (int i = 0; i < CHAINS; i++) s ^= x[i];                                       \
        if (s == 0x12345678u) out[0] = s;                                                 \
    }


#define BODY2(NAME, ASM)                                                                  \
    __global__ __launch_bounds__(256) void k_##NAME(uint32_t *out, int iters, uint32_t a, \
                                                     uint32_t b)                          \
    {                                                                                     \
        double x[CHAINS];                                                                 \
        double bb = __hiloint2double(b, b);                                               \
        for (int i = 0; i < CHAINS; i++) x[i] = __hiloint2double(threadIdx.x + i + a, threadIdx.x * 3 + i); \
        for (int it = 0; it < iters; it++) {                                              \
            _Pragma("unroll") for (int r = 0; r < REPS / CHAINS; r++) {                   \
                _Pragma("unroll") for (int i = 0; i < CHAINS; i++) {                      \
                    asm volatile(ASM : "+v"(x[i]) : "v"(bb));                             \
                }                                                                         \
            }                                                                             \
        }                                                                                 \
        double s = 0;                                                                     \
        for (int i = 0; i < CHAINS; i++) s += x[i];                                       \
        if (s == 0.12345) out[0] = 1;                                                     \
    }

BODY(add, "v_add_u32 %0, %0, %1")
BODY(mul_u24, "v_mul_u32_u24 %0, %0, %1")
BODY(mul_hi_u24, "v_mul_hi_u32_u24 %0, %0, %1")
BODY(mul_lo, "v_mul_lo_u32 %0, %0, %1")
BODY(mul_hi, "v_mul_hi_u32 %0, %0, %1")
BODY(mad_u32_u16, "v_mad_u32_u16 %0, %0, %1, %0")
BODY(mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %0")
BODY(pk_max_u16, "v_pk_max_u16 %0, %0, %1")
BODY(pk_sub_i16, "v_pk_sub_i16 %0, %0, %1")
BODY(pk_min_u16, "v_pk_min_u16 %0, %0, %1")
BODY(cvt_pk_u16, "v_cvt_pk_u16_u32 %0, %0, %1")
BODY(perm, "v_perm_b32 %0, %0, %1, %2")
BODY(lshl_or, "v_lshl_or_b32 %0, %0, 16, %1")
BODY(max3, "v_max3_u32 %0, %0, %1, %1")
BODY(alignbit, "v_alignbit_b32 %0, %0, %1, 7")
BODY(dot2_u16, "v_dot2_u32_u16 %0, %0, %1, %0")
BODY(xor_sdwa, "v_xor_b32_sdwa %0, %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1")
BODY(lshr_s, "v_lshrrev_b32 %0, %2, %0")
BODY(cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
BODY(mul_f32, "v_mul_f32 %0, %0, %1")
BODY(fmac_f32, "v_fmac_f32 %0, %1, %1")
BODY(fma_f32, "v_fma_f32 %0, %0, %1, %1")
BODY2(pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %1")
BODY2(pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
BODY2(pk_add_f32, "v_pk_add_f32 %0, %0, %1")

template <typename K>
static void run(const char *name, K kern, uint32_t *d)
{
    const int iters = 2000;
    const int blocks = 256 * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 10, 3u, 5u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters, 3u, 5u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * 256 * iters * REPS;     // lane-ops
    // cycles per wave-instruction per SIMD at 2.4 GHz: 1024 SIMDs
    const double wave_instr = ops / 64;
    const double cyc = ms * 1e-3 * 2.4e9 * 1024 / wave_instr;
    printf("%-14s %8.3f ms  %7.2f Tlane-op/s  ~%.2f cyc/wave-instr/SIMD@2.4GHz\n", name, ms,
           ops / (ms * 1e-3) / 1e12, cyc);
}

int main()
{
    uint32_t *d;
    hipMalloc(&d, 4096);
#define RUN(n) run(#n, k_##n, d)
    RUN(add); RUN(mul_u24); RUN(mul_hi_u24); RUN(mul_lo); RUN(mul_hi); RUN(mad_u32_u16);
    RUN(mad_u32_u24); RUN(pk_max_u16); RUN(pk_sub_i16); RUN(pk_min_u16); RUN(cvt_pk_u16);
    RUN(perm); RUN(lshl_or); RUN(max3); RUN(alignbit); RUN(dot2_u16); RUN(xor_sdwa); RUN(lshr_s);
    RUN(cvt_f32_u32); RUN(mul_f32); RUN(fmac_f32); RUN(fma_f32); RUN(pk_fma_f32); RUN(pk_mul_f32); RUN(pk_add_f32);
    return 0;
}
