// ubench_lds2.hip -- issue cost vs. latency of LDS instructions for ONE wave on gfx950.
// Not part of the product.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

// N reads issued back to back; t_issue = after the last issue, t_done = after lgkmcnt(0)
template <int N, bool WRITE, int FILL>
__global__ void k(float *out, uint64_t *cyc, int iters, uint32_t lane_stride, uint32_t active)
{
    extern __shared__ float lds[];
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (unsigned i = threadIdx.x; i < 8192 * (blockDim.x >> 6); i += blockDim.x) lds[i] = 1e-3f;
    __syncthreads();
    const uint32_t base = wave * 32768 + lane * lane_stride;
    f4 v[N];
    for (int i = 0; i < N; i++) v[i] = (f4){1.f, 2.f, 3.f, 4.f};
    float x = lane;
    uint64_t issue = 0, done = 0;
    if (lane < active)
    for (int it = 0; it < iters; it++) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll
        for (int i = 0; i < N; i++) {
            if (WRITE) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(base), "v"(v[i]), "n"(i * 16) : "memory");
            else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[i]) : "v"(base), "n"(i * 16));
#pragma unroll
            for (int f = 0; f < FILL; f++) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x));
        }
        const uint64_t t1 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const uint64_t t2 = __builtin_readcyclecounter();
        issue += t1 - t0;
        done += t2 - t0;
    }
    float s = x;
    for (int i = 0; i < N; i++) s += v[i].x;
    if (s == 0.12345f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { cyc[0] = issue; cyc[1] = done; }
}

static float *out;
static uint64_t *cyc;
template <int N, bool WRITE, int FILL>
static void run(int waves, uint32_t active = 64)
{
    const int iters = 1000;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k<N, WRITE, FILL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL((k<N, WRITE, FILL>), dim3(256), dim3(64 * waves), 32768 * waves, 0, out, cyc, iters, 272u, active);
        CHECK(hipDeviceSynchronize());
    }
    uint64_t c[2];
    CHECK(hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost));
    printf("%s b128 x%2d  fill %d fma  active %u waves %d: issued after %7.1f clk (%5.1f/instr)  complete after %7.1f clk\n", WRITE ? "write" : "read ",
           N, FILL, active, waves, (double)c[0] / iters, (double)c[0] / iters / N, (double)c[1] / iters);
}

int main()
{
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMalloc(&cyc, 64));
    run<1, false, 0>(1); run<2, false, 0>(1); run<4, false, 0>(1); run<8, false, 0>(1); run<16, false, 0>(1);
    run<1, true, 0>(1); run<2, true, 0>(1); run<4, true, 0>(1); run<8, true, 0>(1); run<16, true, 0>(1);
    run<16, false, 4>(1); run<16, false, 8>(1); run<16, true, 4>(1); run<16, true, 8>(1);
    run<16, false, 0>(4); run<16, true, 0>(4); run<16, false, 8>(4); run<16, true, 8>(4);
    for (uint32_t act : {32u, 16u}) for (int w : {1, 4, 8}) { run<16, true, 0>(w, act); run<16, false, 0>(w, act); }
    run<16, true, 0>(8); run<16, false, 0>(8); run<16, true, 0>(2); run<16, true, 0>(3);
    return 0;
}
