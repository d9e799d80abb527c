// ubench_addtid.hip -- the recurrence row loop of k_eq_pipe (16 ds_read_b128, 128 dependent
// FMAs) with its 64 results per lane stored three ways: 16 ds_write_b128 into the lane's row,
// 64 ds_write_b32 time-major, 64 ds_write_addtid_b32 time-major (address = M0 + offset + 4*lane,
// no address VGPR: 2 cycles of the SIMD->LDS path per store instead of 13 for a b128).
// Cycles per step for wave 0 of block 0; 1, 2 or 4 such waves per CU, barrier per step.  Not product code.
// MI355X: one wave 1220 / 1224 / 1540 clk per step, four waves (one per SIMD) 1904 / 1767 / 1717:
// a lone wave issues addtid stores at ~13 clk each, and with every SIMD storing the SIMD->LDS
// path is the limit whatever the opcode.  k_eq_pipe keeps the b128 rows.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int OFF>
__device__ __forceinline__ void st_addtid(float v)
{
    asm volatile("ds_write_addtid_b32 %0 offset:%1" :: "v"(v), "n"(OFF) : "memory");
}

template <int T, int MODE>
__device__ __forceinline__ void rec4(const float4 x, float &h1, float &h2, float c1, float c2, float *rowp, float *slab, unsigned lane)
{
    float4 y;
    y.x = __builtin_fmaf(c1, h1, __builtin_fmaf(c2, h2, x.x));
    y.y = __builtin_fmaf(c1, y.x, __builtin_fmaf(c2, h1, x.y));
    y.z = __builtin_fmaf(c1, y.y, __builtin_fmaf(c2, y.x, x.z));
    y.w = __builtin_fmaf(c1, y.z, __builtin_fmaf(c2, y.y, x.w));
    h2 = y.z;
    h1 = y.w;
    if (MODE == 0) {
        reinterpret_cast<float4 *>(rowp)[T] = y;
    } else if (MODE == 1) {
        slab[(4 * T + 0) * 65 + lane] = y.x;
        slab[(4 * T + 1) * 65 + lane] = y.y;
        slab[(4 * T + 2) * 65 + lane] = y.z;
        slab[(4 * T + 3) * 65 + lane] = y.w;
    } else {
        st_addtid<(4 * T + 0) * 260>(y.x);
        st_addtid<(4 * T + 1) * 260>(y.y);
        st_addtid<(4 * T + 2) * 260>(y.z);
        st_addtid<(4 * T + 3) * 260>(y.w);
    }
}

template <int MODE, bool BAR>
__global__ void k_rec(float *out, uint64_t *cyc, int iters, float c1, float c2)
{
    extern __shared__ float lds[];
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    float *rowp = lds + (wave * 64 + lane) * 68;                 // input rows (and MODE 0 output)
    float *slab = lds + nw * 64 * 68 + wave * (64 * 65);         // time-major output of this wave
    for (int t = 0; t < 64; t++) rowp[t] = (float)(threadIdx.x & 15) * 1e-3f;
    __syncthreads();
    float h1 = 0.f, h2 = 0.f;
    const unsigned m0v = (unsigned)((nw * 64 * 68 + wave * (64 * 65)) * 4);
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        float4 v[16];
#pragma unroll
        for (int t = 0; t < 16; t++) v[t] = reinterpret_cast<const float4 *>(rowp)[t];
        if (MODE == 2)
            asm volatile("s_mov_b32 m0, %0" :: "s"(__builtin_amdgcn_readfirstlane(m0v)) : "memory");
        rec4<0, MODE>(v[0], h1, h2, c1, c2, rowp, slab, lane);   rec4<1, MODE>(v[1], h1, h2, c1, c2, rowp, slab, lane);
        rec4<2, MODE>(v[2], h1, h2, c1, c2, rowp, slab, lane);   rec4<3, MODE>(v[3], h1, h2, c1, c2, rowp, slab, lane);
        rec4<4, MODE>(v[4], h1, h2, c1, c2, rowp, slab, lane);   rec4<5, MODE>(v[5], h1, h2, c1, c2, rowp, slab, lane);
        rec4<6, MODE>(v[6], h1, h2, c1, c2, rowp, slab, lane);   rec4<7, MODE>(v[7], h1, h2, c1, c2, rowp, slab, lane);
        rec4<8, MODE>(v[8], h1, h2, c1, c2, rowp, slab, lane);   rec4<9, MODE>(v[9], h1, h2, c1, c2, rowp, slab, lane);
        rec4<10, MODE>(v[10], h1, h2, c1, c2, rowp, slab, lane); rec4<11, MODE>(v[11], h1, h2, c1, c2, rowp, slab, lane);
        rec4<12, MODE>(v[12], h1, h2, c1, c2, rowp, slab, lane); rec4<13, MODE>(v[13], h1, h2, c1, c2, rowp, slab, lane);
        rec4<14, MODE>(v[14], h1, h2, c1, c2, rowp, slab, lane); rec4<15, MODE>(v[15], h1, h2, c1, c2, rowp, slab, lane);
        if (MODE == 2)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (BAR) __syncthreads();
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    float s = h1 + h2 + slab[lane] + slab[63 * 65 + lane];
    if (s == 0.12345f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
    if (blockIdx.x == 0 && threadIdx.x < 64 && MODE != 0)       // layout check: slab[t][lane] of wave 0
        out[1 + threadIdx.x] = lds[nw * 64 * 68 + 5 * 65 + threadIdx.x];
}

int main()
{
    float *out;
    uint64_t *cyc;
    CHECK(hipMalloc(&out, 1024));
    CHECK(hipMalloc(&cyc, 64));
    const int iters = 2000, grid = 256;
    float chk[3][65];
#define RUN(MODE, BAR, THREADS) do { \
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rec<MODE, BAR>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        for (int r = 0; r < 2; r++) { \
            hipLaunchKernelGGL((k_rec<MODE, BAR>), dim3(grid), dim3(THREADS), ((THREADS) * 68 + (THREADS) * 65) * 4, 0, out, cyc, iters, -0.5f, 0.25f); \
            CHECK(hipDeviceSynchronize()); } \
        uint64_t c = 0; CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost)); \
        CHECK(hipMemcpy(chk[MODE], out, 65 * 4, hipMemcpyDeviceToHost)); \
        printf("store mode %d (0 b128 rows, 1 b32 time-major, 2 addtid) bar=%d threads=%4d: %8.1f clk/step\n", MODE, (int)BAR, THREADS, (double)c / iters); \
    } while (0)
    RUN(0, true, 64);  RUN(1, true, 64);  RUN(2, true, 64);
    RUN(0, true, 128); RUN(1, true, 128); RUN(2, true, 128);
    RUN(0, true, 256); RUN(1, true, 256); RUN(2, true, 256);
    int bad = 0;
    for (int i = 1; i < 65; i++) bad += chk[1][i] != chk[2][i];
    printf("addtid layout vs b32 time-major: %d of 64 lanes differ\n", bad);
    return 0;
}
