// ubench_chain.hip -- what one wave can do per cycle when few waves share a SIMD: the
// situation of the pipelined EQ kernel (k_eq_pipe).  Dependent and independent FMA
// chains, and the recurrence row loop (16 ds_read_b128, 128 dependent FMAs,
// 16 ds_write_b128) with 1..8 waves per CU.  Cycles are s_memtime ticks measured inside
// the kernel by wave 0 of block 0.  Not part of the product.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// NCH independent chains of dependent v_fma_f32, 256 FMAs per iteration in total
template <int NCH>
__global__ void k_fma(float *out, uint64_t *cyc, int iters, float b, float c)
{
    float x[NCH];
    for (int i = 0; i < NCH; i++) x[i] = threadIdx.x * 0.001f + i;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 256 / NCH; r++)
#pragma unroll
            for (int i = 0; i < NCH; i++)
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(b), "v"(c));
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < NCH; i++) s += x[i];
    if (s == 0.12345f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

// recurrence row loop as in k_eq_pipe: lane = row, rows of 68 floats
template <bool LDS_R, bool LDS_W, bool BAR>
__global__ void k_rec(float *out, uint64_t *cyc, int iters, float c1, float c2)
{
    extern __shared__ float lds[];
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *rowp = lds + (wave * 64 + lane) * 68;
    for (int t = 0; t < 64; t++) rowp[t] = (float)(threadIdx.x & 15) * 1e-3f;
    __syncthreads();
    float h1 = 0.f, h2 = 0.f;
    float4 v[16];
    for (int t = 0; t < 16; t++) v[t] = make_float4(1e-3f, 2e-3f, 3e-3f, 1e-3f);
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if (LDS_R) {
#pragma unroll
            for (int t = 0; t < 16; t++) v[t] = reinterpret_cast<const float4 *>(rowp)[t];
        }
#pragma unroll
        for (int t = 0; t < 16; t++) {
            float4 y;
            y.x = __builtin_fmaf(c1, h1, __builtin_fmaf(c2, h2, v[t].x));
            y.y = __builtin_fmaf(c1, y.x, __builtin_fmaf(c2, h1, v[t].y));
            y.z = __builtin_fmaf(c1, y.y, __builtin_fmaf(c2, y.x, v[t].z));
            y.w = __builtin_fmaf(c1, y.z, __builtin_fmaf(c2, y.y, v[t].w));
            h2 = y.z;
            h1 = y.w;
            if (LDS_W)
                reinterpret_cast<float4 *>(rowp)[t] = y;
            else
                v[t] = y;
        }
        if (BAR) __syncthreads();
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    float s = h1 + h2;
    for (int t = 0; t < 16; t++) s += v[t].x;
    if (s == 0.12345f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

// feed-forward row loop: 3 ops per sample, no loop-carried dependence inside the row
template <bool LDS_RW>
__global__ void k_fir(float *out, uint64_t *cyc, int iters, float c0, float c1, float c2)
{
    extern __shared__ float lds[];
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *rowp = lds + (wave * 64 + lane) * 68;
    for (int t = 0; t < 64; t++) rowp[t] = (float)(threadIdx.x & 15) * 1e-3f;
    __syncthreads();
    float h1 = 0.f, h2 = 0.f;
    float4 v[16];
    for (int t = 0; t < 16; t++) v[t] = make_float4(1e-3f, 2e-3f, 3e-3f, 1e-3f);
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if (LDS_RW) {
#pragma unroll
            for (int t = 0; t < 16; t++) v[t] = reinterpret_cast<const float4 *>(rowp)[t];
        }
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const float4 x = v[t];
            float4 f;
            f.x = __builtin_fmaf(c2, h2, __builtin_fmaf(c1, h1, c0 * x.x));
            f.y = __builtin_fmaf(c2, h1, __builtin_fmaf(c1, x.x, c0 * x.y));
            f.z = __builtin_fmaf(c2, x.x, __builtin_fmaf(c1, x.y, c0 * x.z));
            f.w = __builtin_fmaf(c2, x.y, __builtin_fmaf(c1, x.z, c0 * x.w));
            h2 = x.z;
            h1 = x.w;
            if (LDS_RW)
                reinterpret_cast<float4 *>(rowp)[t] = f;
            else
                v[t] = f;
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    float s = h1 + h2;
    for (int t = 0; t < 16; t++) s += v[t].x;
    if (s == 0.12345f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}


// pure LDS traffic in the row pattern: 16 b128 reads (or writes) per lane and iteration
template <int MODE, int STRIDE>      // 0 read, 1 write, 2 both
__global__ void k_lds(float *out, uint64_t *cyc, int iters)
{
    extern __shared__ float lds[];
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *rowp = lds + (wave * 64 + lane) * STRIDE;
    for (int t = 0; t < 64; t++) rowp[t] = (float)(threadIdx.x & 15) * 1e-3f;
    __syncthreads();
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        asm volatile("" ::: "memory");
        if (MODE != 1) {
            float4 v[16];
#pragma unroll
            for (int t = 0; t < 16; t++) v[t] = reinterpret_cast<const float4 *>(rowp)[t];
#pragma unroll
            for (int t = 0; t < 16; t++) { acc.x += v[t].x; acc.y += v[t].w; }
        }
        if (MODE != 0) {
#pragma unroll
            for (int t = 0; t < 16; t++) reinterpret_cast<float4 *>(rowp)[t] = acc;
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    if (acc.x + acc.y == 0.12345f) out[0] = acc.x;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

// recurrence with the next row prefetched into registers while this one is worked on
// (two LDS regions so that reads and writes touch different rows), barrier per step
template <bool BAR>
__global__ void k_rec_pf(float *out, uint64_t *cyc, int iters, float c1, float c2)
{
    extern __shared__ float lds[];
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    float *rin = lds + (wave * 64 + lane) * 68;
    float *rout = lds + ((nw + wave) * 64 + lane) * 68;
    for (int t = 0; t < 64; t++) rin[t] = rout[t] = (float)(threadIdx.x & 15) * 1e-3f;
    __syncthreads();
    float h1 = 0.f, h2 = 0.f;
    float4 a[16], b[16];
    for (int t = 0; t < 16; t++) a[t] = b[t] = make_float4(1e-3f, 2e-3f, 3e-3f, 1e-3f);
    auto step = [&](float4 (&cur)[16], float4 (&nxt)[16]) {
#pragma unroll
        for (int t = 0; t < 16; t++) nxt[t] = reinterpret_cast<const float4 *>(rin)[t];
#pragma unroll
        for (int t = 0; t < 16; t++) {
            float4 y;
            y.x = __builtin_fmaf(c1, h1, __builtin_fmaf(c2, h2, cur[t].x));
            y.y = __builtin_fmaf(c1, y.x, __builtin_fmaf(c2, h1, cur[t].y));
            y.z = __builtin_fmaf(c1, y.y, __builtin_fmaf(c2, y.x, cur[t].z));
            y.w = __builtin_fmaf(c1, y.z, __builtin_fmaf(c2, y.y, cur[t].w));
            h2 = y.z;
            h1 = y.w;
            reinterpret_cast<float4 *>(rout)[t] = y;
        }
        if (BAR) __syncthreads();
    };
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it += 2) {
        step(a, b);
        step(b, a);
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    float s = h1 + h2;
    for (int t = 0; t < 16; t++) s += a[t].x + b[t].y;
    if (s == 0.12345f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

// the recurrence row loop with its results stored to GLOBAL memory instead of LDS (a scratch that stays in L2:
// rows of 68 floats per lane, as in LDS), reads from LDS as before -- what the issuing wave pays per store there
template <bool NT>
__global__ void k_rec_g(float *out, uint64_t *cyc, int iters, float c1, float c2, float *scratch)
{
    extern __shared__ float lds[];
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    float *rowp = lds + (wave * 64 + lane) * 68;
    float *growp = scratch + ((size_t)(blockIdx.x * nw + wave) * 64 + lane) * 68;
    for (int t = 0; t < 64; t++) rowp[t] = (float)(threadIdx.x & 15) * 1e-3f;
    __syncthreads();
    float h1 = 0.f, h2 = 0.f;
    float4 v[16];
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int t = 0; t < 16; t++) v[t] = reinterpret_cast<const float4 *>(rowp)[t];
#pragma unroll
        for (int t = 0; t < 16; t++) {
            float4 y;
            y.x = __builtin_fmaf(c1, h1, __builtin_fmaf(c2, h2, v[t].x));
            y.y = __builtin_fmaf(c1, y.x, __builtin_fmaf(c2, h1, v[t].y));
            y.z = __builtin_fmaf(c1, y.y, __builtin_fmaf(c2, y.x, v[t].z));
            y.w = __builtin_fmaf(c1, y.z, __builtin_fmaf(c2, y.y, v[t].w));
            h2 = y.z;
            h1 = y.w;
            const f32x4 yy = {y.x, y.y, y.z, y.w};
            if (NT)
                __builtin_nontemporal_store(yy, reinterpret_cast<f32x4 *>(growp) + t);
            else
                reinterpret_cast<f32x4 *>(growp)[t] = yy;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    float s = h1 + h2;
    if (s == 0.12345f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

// barrier cost alone: nothing between the barriers but a few FMAs
__global__ void k_bar(float *out, uint64_t *cyc, int iters, float c)
{
    float x = threadIdx.x;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        x = __builtin_fmaf(x, c, 1.0f);
        __syncthreads();
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    if (x == 0.12345f) out[0] = x;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <typename F>
static void run(const char *name, F launch, uint64_t *dcyc, int iters, double per_iter_ops)
{
    launch();
    CHECK(hipDeviceSynchronize());
    launch();
    CHECK(hipDeviceSynchronize());
    uint64_t c = 0;
    CHECK(hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost));
    printf("%-44s %9.1f clk/iter  %6.2f clk/op\n", name, (double)c / iters, (double)c / iters / per_iter_ops);
}

int main()
{
    float *out;
    uint64_t *cyc;
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMalloc(&cyc, 64));
    const int iters = 2000;
    const int grid = 256;
#define FMA(NCH, THREADS) run("fma chains=" #NCH " threads/block=" #THREADS, [&] { \
        hipLaunchKernelGGL(k_fma<NCH>, dim3(grid), dim3(THREADS), 0, 0, out, cyc, iters, 0.999f, 1e-3f); }, cyc, iters, 256)
    FMA(1, 64); FMA(2, 64); FMA(4, 64); FMA(8, 64);
    FMA(1, 256); FMA(2, 256); FMA(4, 256);
    FMA(1, 512); FMA(2, 512); FMA(4, 512);
#define REC(R, W, B, THREADS) run("rec ldsR=" #R " ldsW=" #W " bar=" #B " threads=" #THREADS, [&] { \
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rec<R, W, B>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        hipLaunchKernelGGL((k_rec<R, W, B>), dim3(grid), dim3(THREADS), (THREADS) * 68 * 4, 0, out, cyc, iters, -0.5f, 0.25f); }, cyc, iters, 128)
    REC(false, false, false, 64);
    REC(true, false, false, 64);
    REC(false, true, false, 64);
    REC(true, true, false, 64);
    REC(true, true, false, 256);
    REC(true, true, true, 256);
    REC(true, true, false, 512);
    REC(true, true, true, 512);
#define FIR(RW, THREADS) run("fir lds=" #RW " threads=" #THREADS, [&] { \
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fir<RW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        hipLaunchKernelGGL((k_fir<RW>), dim3(grid), dim3(THREADS), (THREADS) * 68 * 4, 0, out, cyc, iters, 0.5f, 0.25f, 0.125f); }, cyc, iters, 192)
    FIR(false, 64);
    FIR(true, 64);
    FIR(true, 256);
    FIR(true, 512);

#define LDSB(MODE, STRIDE, THREADS) run("lds mode=" #MODE " stride=" #STRIDE " threads=" #THREADS, [&] { \
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lds<MODE, STRIDE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        hipLaunchKernelGGL((k_lds<MODE, STRIDE>), dim3(grid), dim3(THREADS), (THREADS) * STRIDE * 4, 0, out, cyc, iters); }, cyc, iters, 16)
    LDSB(0, 68, 64); LDSB(0, 68, 256); LDSB(0, 68, 512);
    LDSB(1, 68, 64); LDSB(1, 68, 256); LDSB(1, 68, 512);
    LDSB(2, 68, 256); LDSB(2, 68, 512);
    LDSB(0, 72, 256); LDSB(1, 72, 256); LDSB(0, 76, 256); LDSB(1, 76, 256);
    LDSB(0, 66, 256); LDSB(1, 66, 256);
#define RECPF(B, THREADS) run("rec prefetch bar=" #B " threads=" #THREADS, [&] { \
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rec_pf<B>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        hipLaunchKernelGGL((k_rec_pf<B>), dim3(grid), dim3(THREADS), 2 * (THREADS) * 68 * 4, 0, out, cyc, iters, -0.5f, 0.25f); }, cyc, iters, 128)
    RECPF(false, 64); RECPF(true, 64); RECPF(false, 256); RECPF(true, 256);
    float *scratch;
    CHECK(hipMalloc(&scratch, (size_t)grid * 8 * 64 * 68 * 4));
#define RECG(NT, THREADS) run("rec ldsR=1 GLOBAL W nt=" #NT " threads=" #THREADS, [&] { \
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rec_g<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        hipLaunchKernelGGL((k_rec_g<NT>), dim3(grid), dim3(THREADS), (THREADS) * 68 * 4, 0, out, cyc, iters, -0.5f, 0.25f, scratch); }, cyc, iters, 128)
    RECG(false, 64); RECG(false, 128); RECG(false, 256); RECG(false, 512);
    RECG(true, 64); RECG(true, 256);
    run("barrier only threads=256", [&] { hipLaunchKernelGGL(k_bar, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 0.5f); }, cyc, iters, 1);
    run("barrier only threads=512", [&] { hipLaunchKernelGGL(k_bar, dim3(grid), dim3(512), 0, 0, out, cyc, iters, 0.5f); }, cyc, iters, 1);
    return 0;
}
