// ubench_copy.hip -- what read+write streaming patterns reach on MI355X HBM.
// Not part of the product; informs the launch geometry of k_run_fast and gives the
// "measured ceiling" lines of DESIGN.md.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#define uint4 u4

template <int NT_LD, int NT_ST>
__device__ __forceinline__ void cp(const uint4 *s, uint4 *d, size_t i)
{
    uint4 v;
    if (NT_LD) v = __builtin_nontemporal_load(s + i); else v = s[i];
    v.x += 1;
    if (NT_ST) __builtin_nontemporal_store(v, d + i); else d[i] = v;
}

// A: grid-stride, one vector per thread per step
template <int NT_LD, int NT_ST>
__global__ __launch_bounds__(256) void k_stride(const uint4 *s, uint4 *d, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        cp<NT_LD, NT_ST>(s, d, i);
}

// B: wave-contiguous chunks: wave w owns [w*CH, (w+1)*CH) vectors, 4 loads in flight
template <int NT_LD, int NT_ST, int UNROLL>
__global__ __launch_bounds__(256) void k_chunk(const uint4 *s, uint4 *d, size_t n, uint32_t vec_per_wave)
{
    const size_t gw = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    const size_t v0 = gw * vec_per_wave;
    if (v0 >= n) return;
    for (uint32_t b = 0; b < vec_per_wave; b += 64 * UNROLL) {
        uint4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            size_t i = v0 + b + 64 * u + lane;
            if (i >= n) i = n - 1;          // never out of bounds, whatever the geometry
            if (NT_LD) v[u] = __builtin_nontemporal_load(s + i); else v[u] = s[i];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const size_t i = v0 + b + 64 * u + lane;
            v[u].x += 1;
            if (i < n && b + 64 * u < vec_per_wave) {
                if (NT_ST) __builtin_nontemporal_store(v[u], d + i); else d[i] = v[u];
            }
        }
    }
}

// C: block-contiguous: block owns a contiguous range, threads stride by 256 inside it
template <int NT_LD, int NT_ST, int UNROLL>
__global__ __launch_bounds__(256) void k_block(const uint4 *s, uint4 *d, size_t n, uint32_t vec_per_block)
{
    const size_t v0 = (size_t)blockIdx.x * vec_per_block;
    if (v0 >= n) return;
    for (uint32_t b = 0; b < vec_per_block; b += 256 * UNROLL) {
        uint4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            size_t i = v0 + b + 256 * u + threadIdx.x;
            if (i >= n) i = n - 1;
            if (NT_LD) v[u] = __builtin_nontemporal_load(s + i); else v[u] = s[i];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const size_t i = v0 + b + 256 * u + threadIdx.x;
            v[u].x += 1;
            if (i < n && b + 256 * u < vec_per_block) {
                if (NT_ST) __builtin_nontemporal_store(v[u], d + i); else d[i] = v[u];
            }
        }
    }
}

static float timeit(void (*launch)(void), int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); launch();
    hipEventRecord(e0);
    for (int i = 0; i < iters; i++) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / iters;
}

static uint4 *S, *D;
static size_t N;
static uint32_t P;
static int G;


static void a00(void) { k_stride<0,0><<<G,256>>>(S, D, N); }
static void a01(void) { k_stride<0,1><<<G,256>>>(S, D, N); }
static void a11(void) { k_stride<1,1><<<G,256>>>(S, D, N); }
static void a00i(void) { k_stride<0,0><<<G,256>>>(S, S, N); }
static void b00(void) { k_chunk<0,0,4><<<(N / P + 3) / 4,256>>>(S, D, N, P); }
static void b01(void) { k_chunk<0,1,4><<<(N / P + 3) / 4,256>>>(S, D, N, P); }
static void b11(void) { k_chunk<1,1,4><<<(N / P + 3) / 4,256>>>(S, D, N, P); }
static void b00i(void) { k_chunk<0,0,4><<<(N / P + 3) / 4,256>>>(S, S, N, P); }
static void b00u8(void) { k_chunk<0,0,8><<<(N / P + 3) / 4,256>>>(S, D, N, P); }
static void c00(void) { k_block<0,0,4><<<N / (4 * (size_t)P),256>>>(S, D, N, 4 * P); }
static void c01(void) { k_block<0,1,4><<<N / (4 * (size_t)P),256>>>(S, D, N, 4 * P); }
static void c11(void) { k_block<1,1,4><<<N / (4 * (size_t)P),256>>>(S, D, N, 4 * P); }
static void c00i(void) { k_block<0,0,4><<<N / (4 * (size_t)P),256>>>(S, S, N, 4 * P); }

int main(int argc, char **argv)
{
    const size_t bytes = (size_t)1 << 30;
    N = bytes / 16;
    hipMalloc(&S, bytes); hipMalloc(&D, bytes);
    hipMemset(S, 1, bytes); hipMemset(D, 2, bytes);
    struct { const char *n; void (*f)(void); } v[] = {
        {"stride        ", a00}, {"stride st.nt  ", a01}, {"stride ld+st.nt", a11}, {"stride inplace", a00i},
        {"wavechunk     ", b00}, {"wavechunk st.nt", b01}, {"wavechunk ld+st.nt", b11}, {"wavechunk inplace", b00i},
        {"wavechunk u8  ", b00u8},
        {"blockchunk    ", c00}, {"blockchunk st.nt", c01}, {"blockchunk ld+st.nt", c11}, {"blockchunk inplace", c00i}};
    int grids[] = {256 * 4, 256 * 8, 256 * 16, 256 * 32};
    uint32_t per[] = {256, 512, 2048, 8192};
    for (int gi = 0; gi < 4; gi++) {
        G = grids[gi]; P = per[gi];
        printf("--- grid(stride)=%d  vec_per_wave=%u (%u KiB per wave, %u KiB per block)\n", G, P, P * 16 / 1024, P * 64 / 1024);
        for (unsigned i = 0; i < sizeof(v) / sizeof(v[0]); i++) {
            float ms = timeit(v[i].f, 10);
            printf("%-20s %7.3f ms  %7.1f GB/s (r+w)\n", v[i].n, ms, 2.0 * bytes / (ms * 1e-3) / 1e9);
            fflush(stdout);
        }
    }
    return 0;
}
