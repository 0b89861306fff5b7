// Diagnostic (not part of the library): what a pure streaming read of 16-byte records reaches on this device, for a few
// shapes of the loop -- the yardstick for part_reduce / rs_hist (4.2-4.3 TB/s on 320 MB).
//   hipcc --offload-arch=gfx950 -O3 tools/probe/bw_probe.hip -o /tmp/bw_probe && /tmp/bw_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

template <int ITEMS, int THREADS>
__global__ __launch_bounds__(THREADS) void rd(const uint4 *in, uint32_t n, uint32_t *out)
{
    const uint32_t t0 = blockIdx.x * (ITEMS * THREADS);
    uint4 e[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) e[j] = in[min(t0 + j * THREADS + threadIdx.x, n - 1u)];
    uint32_t x = 0;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) x ^= e[j].x ^ e[j].y ^ e[j].z ^ e[j].w;
    if (x == 0x12345678u) out[blockIdx.x] = x;      // (never: keeps the loads)
}

template <int ITEMS, int THREADS>
__global__ __launch_bounds__(THREADS) void rd_persist(const uint4 *in, uint32_t n, uint32_t *out)
{
    uint32_t x = 0;
    const uint32_t tiles = (n + ITEMS * THREADS - 1) / (ITEMS * THREADS);
    for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const uint32_t t0 = tile * (ITEMS * THREADS);
        uint4 e[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) e[j] = in[min(t0 + j * THREADS + threadIdx.x, n - 1u)];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) x ^= e[j].x ^ e[j].y ^ e[j].z ^ e[j].w;
    }
    if (x == 0x12345678u) out[blockIdx.x] = x;
}

template <class F>
static float time_it(F f, int reps)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main()
{
    const uint32_t n = 20000000u;
    uint4 *in;
    uint32_t *out;
    hipMalloc(&in, (size_t)n * 16);
    hipMalloc(&out, 1 << 22);
    hipMemset(in, 1, (size_t)n * 16);
    uint4 *big;                                      // 1 GiB streamed between two measurements: nothing of `in` stays in the Infinity Cache
    const uint32_t nbig = 1u << 26;
    hipMalloc(&big, (size_t)nbig * 16);
    hipMemset(big, 2, (size_t)nbig * 16);
    const double bytes = (double)n * 16;
#define RUN(NAME, KERNEL, GRID, THREADS)                                                                      \
    {                                                                                                         \
        float ms = time_it([&] { hipLaunchKernelGGL(KERNEL, dim3(GRID), dim3(THREADS), 0, 0, in, n, out); }, 20); \
        float msc = time_it([&] {                                                                             \
            hipLaunchKernelGGL((rd<8, 256>), dim3((nbig + 2047) / 2048), dim3(256), 0, 0, big, nbig, out);     \
            hipLaunchKernelGGL(KERNEL, dim3(GRID), dim3(THREADS), 0, 0, in, n, out); }, 10);                   \
        float mf = time_it([&] { hipLaunchKernelGGL((rd<8, 256>), dim3((nbig + 2047) / 2048), dim3(256), 0, 0, big, nbig, out); }, 10); \
        printf("%-28s warm %7.1f us %6.2f TB/s   cold %7.1f us %6.2f TB/s\n", NAME, ms * 1e3, bytes / ms / 1e9, (msc - mf) * 1e3, bytes / (msc - mf) / 1e9); \
    }
    RUN("8 x 16 B, 256 thr", (rd<8, 256>), (n + 2047) / 2048, 256)
    RUN("4 x 16 B, 256 thr", (rd<4, 256>), (n + 1023) / 1024, 256)
    RUN("16 x 16 B, 256 thr", (rd<16, 256>), (n + 4095) / 4096, 256)
    RUN("8 x 16 B, 512 thr", (rd<8, 512>), (n + 4095) / 4096, 512)
    RUN("8 x 16 B, 1024 thr", (rd<8, 1024>), (n + 8191) / 8192, 1024)
    RUN("persistent 8x, 2048 wg", (rd_persist<8, 256>), 2048, 256)
    RUN("persistent 8x, 4096 wg", (rd_persist<8, 256>), 4096, 256)
    {
        float mf = time_it([&] { hipLaunchKernelGGL((rd<8, 256>), dim3((nbig + 2047) / 2048), dim3(256), 0, 0, big, nbig, out); }, 10);
        printf("1 GiB stream: %.1f us = %.2f TB/s\n", mf * 1e3, (double)nbig * 16 / mf / 1e9);
    }
    return 0;
}
