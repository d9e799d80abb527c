// ubench_roundtrip.hip -- what one launch-and-wait costs the host on MI355X, for the per-stream operator path
// (one 1 KiB block per pull: csrc/transform.c).  Variants:
//   sync        empty kernel, hipStreamSynchronize
//   sync+ext    empty kernel through hipExtLaunchKernelGGL with a stop event (as the engine launches), sync
//   flag        kernel stores a sequence number to pinned host memory (system-scope release); the host spins on it
//   io sync     kernel reads 1 KiB from pinned host memory, writes 1 KiB back (the HOSTPCM slots), sync
//   io flag     the same, completion by flag
//   resident    ONE launch of a kernel that stays: it polls a request word in pinned host memory, moves the 1 KiB
//               in and out, stores the request's number as completion, and polls again -- bounded: it leaves after
//               `max_idle_polls` polls without a request or after `max_blocks` blocks, whatever the host does
// hipcc --offload-arch=gfx950 -O3 tools/ubench_roundtrip.hip -o tools/ubench_roundtrip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <time.h>

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

static double now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

__global__ void k_empty() {}

__global__ void k_flag(volatile uint32_t *flag, uint32_t seq)
{
    if (threadIdx.x == 0) {
        __atomic_store_n((uint32_t *)flag, seq, __ATOMIC_RELEASE);      // system scope by default for host memory
    }
}

__global__ void k_io(const uint4 *in, uint4 *out, volatile uint32_t *flag, uint32_t seq)
{
    uint4 v = in[threadIdx.x];
    v.x += 1; v.y ^= v.x; v.z += v.y; v.w ^= v.z;
    out[threadIdx.x] = v;
    if (flag) {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_store((uint32_t *)flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// host -> GPU: mb[0] request number, mb[1] quit;  GPU -> host: st[0] last completed request
__global__ void k_resident(uint32_t *mb, uint32_t *st, const uint4 *in, uint4 *out, uint32_t max_idle_polls,
                           uint32_t max_blocks)
{
    __shared__ uint32_t s_req;
    uint32_t served = 0;
    for (uint32_t blk = 0; blk < max_blocks; blk++) {
        if (threadIdx.x == 0) {
            uint32_t r = served;
            for (uint32_t polls = 0; polls < max_idle_polls; polls++) {
                r = __hip_atomic_load(&mb[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
                if (r != served)
                    break;
                if (__hip_atomic_load(&mb[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM))
                    break;
            }
            s_req = r != served ? r : 0xffffffffu;          // nothing came (or quit): leave
        }
        __syncthreads();
        const uint32_t r = s_req;
        __syncthreads();
        if (r == 0xffffffffu)
            break;
        uint4 v = in[threadIdx.x];
        v.x += 1; v.y ^= v.x; v.z += v.y; v.w ^= v.z;
        out[threadIdx.x] = v;
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_store(&st[0], r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        served = r;
    }
}

int main()
{
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    uint32_t *h_flag, *d_flag;
    CHECK(hipHostMalloc((void **)&h_flag, 64, hipHostMallocMapped));
    CHECK(hipHostGetDevicePointer((void **)&d_flag, h_flag, 0));
    uint4 *h_in, *h_out, *d_in, *d_out;
    CHECK(hipHostMalloc((void **)&h_in, 1024, hipHostMallocMapped));
    CHECK(hipHostMalloc((void **)&h_out, 1024, hipHostMallocMapped));
    CHECK(hipHostGetDevicePointer((void **)&d_in, h_in, 0));
    CHECK(hipHostGetDevicePointer((void **)&d_out, h_out, 0));
    hipEvent_t ev;
    CHECK(hipEventCreate(&ev));
    const int N = 5000, W = 500;
    *h_flag = 0;
    uint32_t seq = 0;
    for (int mode = 0; mode < 5; mode++) {
        double t0 = 0;
        for (int i = 0; i < N + W; i++) {
            if (i == W)
                t0 = now();
            ++seq;
            switch (mode) {
            case 0:
                hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st);
                CHECK(hipStreamSynchronize(st));
                break;
            case 1:
                hipExtLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, nullptr, ev, 0);
                CHECK(hipStreamSynchronize(st));
                break;
            case 2:
                hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, st, d_flag, seq);
                while (__atomic_load_n(h_flag, __ATOMIC_ACQUIRE) != seq)
                    ;
                break;
            case 3:
                hipLaunchKernelGGL(k_io, dim3(1), dim3(64), 0, st, d_in, d_out, (volatile uint32_t *)nullptr, seq);
                CHECK(hipStreamSynchronize(st));
                break;
            case 4:
                hipLaunchKernelGGL(k_io, dim3(1), dim3(64), 0, st, d_in, d_out, d_flag, seq);
                while (__atomic_load_n(h_flag, __ATOMIC_ACQUIRE) != seq)
                    ;
                break;
            }
        }
        const double dt = now() - t0;
        static const char *names[] = {"empty kernel, hipStreamSynchronize", "empty kernel, ext launch + stop event, sync",
                                      "flag in pinned host memory, host spins", "1 KiB in + 1 KiB out over PCIe, sync",
                                      "1 KiB in + 1 KiB out over PCIe, flag"};
        printf("%-48s %6.2f us per launch-and-wait\n", names[mode], dt / N * 1e6);
        CHECK(hipStreamSynchronize(st));
    }
    // the resident kernel: one launch serves all requests
    {
        uint32_t *h_mb, *d_mb, *h_st, *d_st;
        CHECK(hipHostMalloc((void **)&h_mb, 64, hipHostMallocMapped));
        CHECK(hipHostMalloc((void **)&h_st, 64, hipHostMallocMapped));
        CHECK(hipHostGetDevicePointer((void **)&d_mb, h_mb, 0));
        CHECK(hipHostGetDevicePointer((void **)&d_st, h_st, 0));
        h_mb[0] = h_mb[1] = 0;
        h_st[0] = 0;
        const uint32_t total = (uint32_t)(N + W);
        hipLaunchKernelGGL(k_resident, dim3(1), dim3(64), 0, st, d_mb, d_st, d_in, d_out, 400000u, total);
        double t0 = 0;
        bool lost = false;
        for (uint32_t i = 1; i <= total && !lost; i++) {
            if (i == (uint32_t)W + 1)
                t0 = now();
            ((uint32_t *)h_in)[0] = i;
            __atomic_store_n(&h_mb[0], i, __ATOMIC_RELEASE);
            const double t_end = now() + 0.5;                 // the kernel is bounded; so is this wait
            while (__atomic_load_n(&h_st[0], __ATOMIC_ACQUIRE) != i) {
                if (now() > t_end) {
                    lost = true;
                    break;
                }
            }
        }
        const double dt = now() - t0;
        __atomic_store_n(&h_mb[1], 1u, __ATOMIC_RELEASE);     // quit (it would leave by itself as well)
        CHECK(hipStreamSynchronize(st));
        if (lost)
            printf("%-48s lost a request (kernel left early?)\n", "resident kernel, mailbox in host memory");
        else
            printf("%-48s %6.2f us per request-and-wait\n", "resident kernel, 1 KiB in + out, mailbox", dt / N * 1e6);
    }
    return 0;
}
