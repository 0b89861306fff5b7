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

// copy: read + write of the same bytes (what a sort pass that only moved records would cost)
template <int ITEMS, int THREADS>
__global__ __launch_bounds__(THREADS) void cp(const uint4 *in, uint32_t n, uint4 *out)
{
    const uint32_t t0 = blockIdx.x * (ITEMS * THREADS);
    uint4 e[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) e[j] = in[min(t0 + j * THREADS + threadIdx.x, n - 1u)];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j)
        if (t0 + j * THREADS + threadIdx.x < n) out[t0 + j * THREADS + threadIdx.x] = e[j];
}
// ... with nontemporal stores / loads (NT bit 0: stores, bit 1: loads)
template <int ITEMS, int THREADS, int NT>
__global__ __launch_bounds__(THREADS) void cp_nt(const uint4 *in, uint32_t n, uint4 *out)
{
    const uint32_t t0 = blockIdx.x * (ITEMS * THREADS);
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    v4 e[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const v4 *src = reinterpret_cast<const v4 *>(in) + min(t0 + j * THREADS + threadIdx.x, n - 1u);
        e[j] = (NT & 2) ? __builtin_nontemporal_load(src) : *src;
    }
#pragma unroll
    for (int j = 0; j < ITEMS; ++j)
        if (t0 + j * THREADS + threadIdx.x < n) {
            v4 *dst = reinterpret_cast<v4 *>(out) + (t0 + j * THREADS + threadIdx.x);
            if (NT & 1) __builtin_nontemporal_store(e[j], dst); else *dst = e[j];
        }
}
// write only
template <int ITEMS, int THREADS, int NT>
__global__ __launch_bounds__(THREADS) void wr(const uint4 *in, uint32_t n, uint4 *out)
{
    const uint32_t t0 = blockIdx.x * (ITEMS * THREADS);
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    const v4 e = {t0, n, threadIdx.x, 7u};
#pragma unroll
    for (int j = 0; j < ITEMS; ++j)
        if (t0 + j * THREADS + threadIdx.x < n) {
            v4 *dst = reinterpret_cast<v4 *>(out) + (t0 + j * THREADS + threadIdx.x);
            if (NT & 1) __builtin_nontemporal_store(e, dst); else *dst = e;
        }
}
// ... with every record stored somewhere else inside a window of W records around its place (a local permutation: what the
// low-bits stage of the sort does to its groups)
template <int ITEMS, int THREADS, int W>
__global__ __launch_bounds__(THREADS) void cp_local(const uint4 *in, uint32_t n, uint4 *out)
{
    const uint32_t t0 = blockIdx.x * (ITEMS * THREADS);
    uint4 e[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) e[j] = in[min(t0 + j * THREADS + threadIdx.x, n - 1u)];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint32_t i = t0 + j * THREADS + threadIdx.x;
        const uint32_t g = i / W * W, k = (i % W * 37u + 11u) % W;      // (37 and W coprime: a permutation of the window)
        if (g + k < n) out[g + k] = e[j];
    }
}

// local permutation + nontemporal hints; LDS bytes per workgroup to bound the workgroups per CU (occupancy)
template <int ITEMS, int THREADS, int W, int NT, int LDS>
__global__ __launch_bounds__(THREADS) void cp_local_nt(const uint4 *in, uint32_t n, uint4 *out)
{
    __shared__ uint32_t s_pad[LDS / 4 + 1];
    if (threadIdx.x == 0) s_pad[0] = n;
    __syncthreads();
    const uint32_t t0 = blockIdx.x * (ITEMS * THREADS) + (s_pad[0] - n);
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    v4 e[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const v4 *src = reinterpret_cast<const v4 *>(in) + min(t0 + j * THREADS + threadIdx.x, n - 1u);
        e[j] = (NT & 2) ? __builtin_nontemporal_load(src) : *src;
    }
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint32_t i = t0 + j * THREADS + threadIdx.x;
        const uint32_t g = i / W * W, k = W == 1 ? 0u : (i % W * 37u + 11u) % W;
        if (g + k < n) {
            v4 *dst = reinterpret_cast<v4 *>(out) + (g + k);
            if (NT & 1) __builtin_nontemporal_store(e[j], dst); else *dst = e[j];
        }
    }
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
        uint4 *dst;
        hipMalloc(&dst, (size_t)n * 16);
        float c1 = time_it([&] { hipLaunchKernelGGL((cp<8, 256>), dim3((n + 2047) / 2048), dim3(256), 0, 0, in, n, dst); }, 20);
        float c2 = time_it([&] { hipLaunchKernelGGL((cp_local<8, 256, 32>), dim3((n + 2047) / 2048), dim3(256), 0, 0, in, n, dst); }, 20);
        float c3 = time_it([&] { hipLaunchKernelGGL((cp_local<8, 256, 128>), dim3((n + 2047) / 2048), dim3(256), 0, 0, in, n, dst); }, 20);
        float c4 = time_it([&] { hipLaunchKernelGGL((cp_local<8, 256, 1024>), dim3((n + 2047) / 2048), dim3(256), 0, 0, in, n, dst); }, 20);
        float n1 = time_it([&] { hipLaunchKernelGGL((cp_nt<8, 256, 1>), dim3((n + 2047) / 2048), dim3(256), 0, 0, in, n, dst); }, 20);
        float n2 = time_it([&] { hipLaunchKernelGGL((cp_nt<8, 256, 2>), dim3((n + 2047) / 2048), dim3(256), 0, 0, in, n, dst); }, 20);
        float n3 = time_it([&] { hipLaunchKernelGGL((cp_nt<8, 256, 3>), dim3((n + 2047) / 2048), dim3(256), 0, 0, in, n, dst); }, 20);
        float w0 = time_it([&] { hipLaunchKernelGGL((wr<8, 256, 0>), dim3((n + 2047) / 2048), dim3(256), 0, 0, in, n, dst); }, 20);
        float w1 = time_it([&] { hipLaunchKernelGGL((wr<8, 256, 1>), dim3((n + 2047) / 2048), dim3(256), 0, 0, in, n, dst); }, 20);
#define CPL(W, NT, LDS) { float t = time_it([&] { hipLaunchKernelGGL((cp_local_nt<8, 256, W, NT, LDS>), dim3((n + 2047) / 2048), dim3(256), 0, 0, in, n, dst); }, 20); \
                         printf("copy, window %4d, nt %d, %5d B LDS per workgroup: %7.1f us\n", W, NT, LDS, t * 1e3); }
        CPL(1, 0, 0) CPL(1, 3, 0) CPL(1, 1, 0) CPL(32, 0, 0) CPL(32, 3, 0) CPL(32, 1, 0) CPL(128, 3, 0) CPL(1024, 3, 0)
        CPL(1, 0, 21504) CPL(1, 3, 21504) CPL(32, 3, 21504) CPL(1, 0, 40000) CPL(1, 3, 40000)
        printf("write only 320 MB:           %7.1f us %6.2f TB/s   nontemporal %7.1f us %6.2f TB/s\n", w0 * 1e3, bytes / w0 / 1e9, w1 * 1e3, bytes / w1 / 1e9);
        printf("copy, nontemporal stores:    %7.1f us   loads: %7.1f us   both: %7.1f us\n", n1 * 1e3, n2 * 1e3, n3 * 1e3);
        printf("copy through uint4 e[8] (the array goes to SCRATCH memory: a compiler artefact, not the device): %7.1f us %6.2f TB/s\n", c1 * 1e3, 2 * bytes / c1 / 1e9);
        printf("... permuted inside 32 recs: %7.1f us %6.2f TB/s\n", c2 * 1e3, 2 * bytes / c2 / 1e9);
        printf("... inside 128 records:      %7.1f us %6.2f TB/s\n", c3 * 1e3, 2 * bytes / c3 / 1e9);
        printf("... inside 1024 records:     %7.1f us %6.2f TB/s\n", c4 * 1e3, 2 * bytes / c4 / 1e9);
    }
    {
        float mf = time_it([&] { hipLaunchKernelGGL((rd<8, 256>), dim3((nbig + 2047) / 2048), dim3(256), 0, 0, big, nbig, out); }, 10);
        printf("1 GiB stream: %.1f us = %.2f TB/s\n", mf * 1e3, (double)nbig * 16 / mf / 1e9);
    }
    return 0;
}
