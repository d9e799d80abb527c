// ubench_lds.hip -- LDS instruction cost on gfx950 by access shape: every lane of a wave
// addresses  base + lane * lane_stride + k * step  (bytes) for k = 0..15, four waves per CU
// (own regions).  Reports clocks per instruction and wave, and bytes per clock per CU.
// Not part of the product.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

enum { R128, R2x64, R64, R32, W128, W2x64, W64, W32 };

template <int OP>
__global__ void k(float *out, uint64_t *cyc, int iters, uint32_t lane_stride, uint32_t step,
                  uint32_t region, uint32_t inner, uint32_t inner_stride)
{
    extern __shared__ float lds[];
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (unsigned i = threadIdx.x; i < region * (blockDim.x >> 6) / 4; i += blockDim.x) lds[i] = 1e-3f;
    __syncthreads();
    // lanes in groups of `inner`: group g at g * lane_stride, lane i of it at i * inner_stride
    const uint32_t base = wave * region + (lane / inner) * lane_stride + (lane % inner) * inner_stride;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (uint32_t k = 0; k < 16; k++) {
            const uint32_t addr = base + k * step;
            if (OP == R128) { f4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr)); asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); acc += v; }
            if (OP == R2x64) { f4 v; asm volatile("ds_read2_b64 %0, %1 offset1:1" : "=v"(v) : "v"(addr)); asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); acc += v; }
            if (OP == R64) { f2 v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr)); asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); acc.x += v.x; acc.y += v.y; }
            if (OP == R32) { float v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr)); asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); acc.x += v; }
            if (OP == W128) asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(acc) : "memory");
            if (OP == W2x64) asm volatile("ds_write2_b64 %0, %1, %2 offset1:1" :: "v"(addr), "v"(acc.xy), "v"(acc.zw) : "memory");
            if (OP == W64) asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(acc.xy) : "memory");
            if (OP == W32) asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(acc.x) : "memory");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const uint64_t t1 = __builtin_readcyclecounter();
    if (acc.x + acc.y + acc.z + acc.w == 0.12345f) out[0] = acc.x;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

static float *out;
static uint64_t *cyc;

template <int OP>
static void run(const char *name, int bytes, uint32_t lane_stride, uint32_t step, int waves, uint32_t inner = 1, uint32_t inner_stride = 0)
{
    const int iters = 1000;
    uint32_t region = 64 * lane_stride + 16 * step + 64;
    region = (region + 255) & ~255u;
    if ((size_t)region * waves > 160 * 1024) { printf("%-10s skip (LDS)\n", name); return; }
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(64 * waves), region * waves, 0, out, cyc, iters, lane_stride, step, region, inner, inner_stride);
        CHECK(hipDeviceSynchronize());
    }
    uint64_t c = 0;
    CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    const double per = (double)c / iters / 16.0;          // clocks per instruction of one wave (all waves in parallel)
    printf("%-10s inner %u x %u B, groups at %5u B, step %5u, waves %d: %7.2f clk/instr/wave  %7.1f B/clk/CU\n", name, inner, inner_stride, lane_stride, step,
           waves, per, 64.0 * bytes * waves / per);
}

int main(int argc, char **)
{
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMalloc(&cyc, 64));
    // the T-wave shape of k_eq_pipe: 8 lanes x 32 B inside a 272-byte row, two b128 per lane
    for (uint32_t is : {32u, 16u, 48u, 36u}) for (uint32_t rs : {272u, 288u, 264u, 320u}) {
        run<R128>("r128", 16, rs, 16, 4, 8, is);
        run<W128>("w128", 16, rs, 16, 4, 8, is);
    }
    run<R128>("r128", 16, 1088, 16, 4, 16, 16);   // store-wave shape: 16 lanes x 16 B per row, 4 rows
    if (argc > 1) return 0;
    const uint32_t strides[] = {16, 272, 264, 280, 288, 520, 528};
    for (uint32_t ls : strides) {
        const uint32_t step = ls == 16 ? 1024 : 16;
        run<R128>("r128", 16, ls, step, 4);
        run<R2x64>("r2x64", 16, ls, step, 4);
        run<W128>("w128", 16, ls, step, 4);
        run<W2x64>("w2x64", 16, ls, step, 4);
    }
    const uint32_t s64[] = {8, 264, 272, 136, 72};
    for (uint32_t ls : s64) {
        const uint32_t step = ls == 8 ? 512 : 8;
        run<R64>("r64", 8, ls, step, 4);
        run<W64>("w64", 8, ls, step, 4);
    }
    const uint32_t s32[] = {4, 260, 132, 68};
    for (uint32_t ls : s32) {
        const uint32_t step = ls == 4 ? 256 : 4;
        run<R32>("r32", 4, ls, step, 4);
        run<W32>("w32", 4, ls, step, 4);
    }
    // wave count sweep for the best shapes
    for (int w : {1, 2, 8}) {
        run<R128>("r128", 16, 16, 1024, w);
        run<W128>("w128", 16, 16, 1024, w);
        run<R2x64>("r2x64", 16, 264, 16, w);
        run<W2x64>("w2x64", 16, 264, 16, w);
    }
    return 0;
}
