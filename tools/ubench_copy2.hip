// ubench_copy2.hip -- second sweep: one pass per wave (no loop), tile = U vectors per lane,
// nontemporal loads+stores, block sizes 64..1024.  Bounds are clamped/guarded everywhere.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <int U, int NT, int BS>
__global__ __launch_bounds__(BS) void k_tile(const u4 *s, u4 *d, size_t n)
{
    const size_t gw = (size_t)blockIdx.x * (BS / 64) + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    const size_t v0 = gw * (64 * U);
    u4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        size_t i = v0 + 64 * u + lane;
        if (i >= n) i = n - 1;
        v[u] = NT ? __builtin_nontemporal_load(s + i) : s[i];
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
        const size_t i = v0 + 64 * u + lane;
        v[u].x += 1;
        if (i < n) { if (NT) __builtin_nontemporal_store(v[u], d + i); else d[i] = v[u]; }
    }
}

static u4 *S, *D; static size_t N;
template <int U, int NT, int BS> static void go(const char *name)
{
    const size_t waves = (N + 64 * U - 1) / (64 * U);
    const unsigned grid = (unsigned)((waves + BS / 64 - 1) / (BS / 64));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; i++) k_tile<U, NT, BS><<<grid, BS>>>(S, D, N);
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) k_tile<U, NT, BS><<<grid, BS>>>(S, D, N);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("%-28s U=%d nt=%d bs=%4d grid=%8u  %7.3f ms %7.1f GB/s\n", name, U, NT, BS, grid, ms,
           2.0 * N * 16 / (ms * 1e-3) / 1e9);
    fflush(stdout);
}

int main()
{
    const size_t bytes = (size_t)1 << 30; N = bytes / 16;
    hipMalloc(&S, bytes); hipMalloc(&D, bytes); hipMemset(S, 1, bytes); hipMemset(D, 2, bytes);
    go<1, 1, 256>("tile"); go<2, 1, 256>("tile"); go<4, 1, 256>("tile"); go<8, 1, 256>("tile");
    go<4, 0, 256>("tile plain"); go<2, 0, 256>("tile plain");
    go<4, 1, 64>("tile"); go<4, 1, 128>("tile"); go<4, 1, 512>("tile"); go<4, 1, 1024>("tile");
    go<2, 1, 512>("tile"); go<2, 1, 1024>("tile"); go<8, 1, 128>("tile"); go<8, 1, 64>("tile");
    return 0;
}
