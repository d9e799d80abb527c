// ubench_place.hip -- does the rate of a read + write stream depend on WHERE its two arrays lie?
// One input array of 1 GiB and several output candidates allocated one after another; the tile copy
// of the hot kernel's access shape (one short wave per 4 KiB, non-temporal 16-byte accesses) is
// timed for every (input, output) pair, in rounds.  Not part of the product.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64) void k_copy(const u32x4 *src, u32x4 *dst)
{
    const size_t v0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    u32x4 w[4];
#pragma unroll
    for (int u = 0; u < 4; u++) w[u] = __builtin_nontemporal_load(src + v0 + 64 * u);
#pragma unroll
    for (int u = 0; u < 4; u++) __builtin_nontemporal_store(w[u], dst + v0 + 64 * u);
}

int main(int argc, char **argv)
{
    const size_t bytes = (size_t)1 << 30;
    const int ncand = argc > 1 ? atoi(argv[1]) : 6;
    const int nin = 2;
    std::vector<void *> in(nin), out(ncand);
    for (int i = 0; i < nin; i++) { CHECK(hipMalloc(&in[i], bytes)); CHECK(hipMemset(in[i], 1, bytes)); }
    for (int i = 0; i < ncand; i++) { CHECK(hipMalloc(&out[i], bytes)); CHECK(hipMemset(out[i], 0, bytes)); }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)(bytes / 4096);
    for (int i = 0; i < 50; i++) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(64), 0, 0, (const u32x4 *)in[0], (u32x4 *)out[0]);
    CHECK(hipDeviceSynchronize());
    for (int rnd = 0; rnd < 3; rnd++) {
        for (int a = 0; a < nin; a++) {
            printf("round %d in %d (%p):", rnd, a, in[a]);
            for (int c = 0; c < ncand; c++) {
                CHECK(hipEventRecord(e0));
                for (int i = 0; i < 20; i++)
                    hipLaunchKernelGGL(k_copy, dim3(grid), dim3(64), 0, 0, (const u32x4 *)in[a], (u32x4 *)out[c]);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                printf(" %5.0f", 2.0 * bytes / (ms / 20 * 1e-3) / 1e9);
            }
            printf("  GB/s\n");
        }
    }
    for (int c = 0; c < ncand; c++) printf("out %d %p\n", c, out[c]);
    return 0;
}
