// Diagnostic (not part of the library): what step E/F's two seams cost INSIDE one launch on this device against the two
// launch boundaries they would replace ("one cooperative / persistent kernel: classify tiles ->
// arrival counter -> the last-arriving workgroup sorts the seeds -> release flag -> finalize tiles").
//
// The skeleton of that kernel with the arithmetic taken out and the data flow kept:
//   producers (T workgroups, one per tile of 256 candidates): spin for `work` microseconds (ef_classify's body), store the
//     tile's 32-byte seed record and its 1.25 KB of per-candidate codes write-through (sc1), drain, barrier, ONE agent-scope
//     add on the arrival counter;
//   the workgroup whose add came last: agent acquire, reads all T records, leaves 2 KB ("the sorted seeds") write-through,
//     drains, sets the flag;
//   consumers (T more workgroups, dispatched behind the producers: they never hold a slot a producer of their XCD still
//     needs, so nobody waits for a workgroup that cannot start): ONE lane polls the flag (relaxed, s_sleep), agent acquire,
//     barrier, read the 2 KB and the tile's codes, write 1.25 KB of results.
// Against it the same three bodies as three launches on one stream (what the library does), and as two (the middle body
// inside every consumer: ef_finalize_own).  Every variant checks what it read (a wrong byte aborts), so a stale hand-off cannot
// pass as a fast one.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/handoff_probe.hip -o /tmp/handoff_probe && /tmp/handoff_probe [tiles=391] [work_us=7]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct Args {
    unsigned long long *rec;      // [T][4]   the tiles' seed records
    unsigned *codes;              // [T][320] per-candidate codes (1.25 KB per tile)
    unsigned *seeds;              // [512]    "the sorted seeds"
    unsigned *out;                // [T][320]
    unsigned *sync;               // [0] arrivals, [64] flag, [128] consumers done, [192] errors -- a 256-byte stretch each: the pollers' loads do not share a line with the arrivals
    unsigned T, epoch, work_ticks;
};

__device__ __forceinline__ void spin_work(unsigned ticks)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
}

__device__ __forceinline__ void produce(const Args &a, unsigned tile, unsigned tid)
{
    spin_work(a.work_ticks);
    for (unsigned i = tid; i < 320u; i += 256u)
        __hip_atomic_store((gu32 *)(a.codes + tile * 320u + i), a.epoch * 1000003u + tile * 320u + i, RLX_AGENT);
    if (tid < 4) __hip_atomic_store((gu64 *)(a.rec + tile * 4u + tid), ((unsigned long long)a.epoch << 32) | (tile * 4u + tid), RLX_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// all T records -> 512 words; `first`/`stride`: how the tiles are dealt to the threads
__device__ __forceinline__ unsigned gather_records(const Args &a, unsigned tid)
{
    unsigned bad = 0, acc = 0;
    for (unsigned b = tid; b < a.T; b += 256u) {
        const ulonglong4 r = reinterpret_cast<const ulonglong4 *>(a.rec)[b];
        bad |= (unsigned)(r.x >> 32) != a.epoch || (unsigned)r.x != b * 4u || (unsigned)r.w != b * 4u + 3u;
        acc += (unsigned)r.y + (unsigned)r.z;
    }
    if (bad) atomicAdd(&a.sync[192], 1u);
    return acc;
}

__device__ __forceinline__ void consume(const Args &a, unsigned tile, unsigned tid)
{
    unsigned bad = 0;
    const unsigned s0 = a.seeds[tid], s1 = a.seeds[256u + tid];
    bad |= s0 != a.epoch * 7u + tid || s1 != a.epoch * 7u + 256u + tid;
    for (unsigned i = tid; i < 320u; i += 256u) {
        const unsigned v = a.codes[tile * 320u + i];
        bad |= v != a.epoch * 1000003u + tile * 320u + i;
        a.out[tile * 320u + i] = v + s0;
    }
    if (bad) atomicAdd(&a.sync[192], 1u);
}

// ---- one launch ------------------------------------------------------------------------------------------------------
template <bool ACQ>
__global__ __launch_bounds__(256) void one_launch(const Args a)
{
    __shared__ unsigned s_last;
    const unsigned tid = threadIdx.x;
    if (blockIdx.x < a.T) {
        produce(a, blockIdx.x, tid);
        __syncthreads();
        if (tid == 0) s_last = __hip_atomic_fetch_add((gu32 *)&a.sync[0], 1u, RLX_AGENT) + 1u == a.T ? 1u : 0u;
        __syncthreads();
        if (!s_last) return;
        if (tid == 0 && ACQ) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        (void)gather_records(a, tid);
        __hip_atomic_store((gu32 *)(a.seeds + tid), a.epoch * 7u + tid, RLX_AGENT);
        __hip_atomic_store((gu32 *)(a.seeds + 256u + tid), a.epoch * 7u + 256u + tid, RLX_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store((gu32 *)&a.sync[64], a.epoch, RLX_AGENT);
        return;
    }
    const unsigned tile = blockIdx.x - a.T;
    if (tid == 0) {
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load((gu32 *)&a.sync[64], RLX_AGENT) != a.epoch) {
            __builtin_amdgcn_s_sleep(32);
            if (wall_clock64() - t0 > 200000000ull) { atomicAdd(&a.sync[192], 1000000u); break; }      // 2 s: give up, loudly
        }
        if (ACQ) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    consume(a, tile, tid);
    __syncthreads();
    if (tid == 0 && __hip_atomic_fetch_add((gu32 *)&a.sync[128], 1u, RLX_AGENT) + 1u == a.T) {          // the last one resets the words
        __hip_atomic_store((gu32 *)&a.sync[0], 0u, RLX_AGENT);
        __hip_atomic_store((gu32 *)&a.sync[128], 0u, RLX_AGENT);
    }
}

// ---- three launches / two launches -------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_produce(const Args a) { produce(a, blockIdx.x, threadIdx.x); }
__global__ __launch_bounds__(256) void k_sort(const Args a)
{
    const unsigned tid = threadIdx.x;
    (void)gather_records(a, tid);
    a.seeds[tid] = a.epoch * 7u + tid;
    a.seeds[256u + tid] = a.epoch * 7u + 256u + tid;
}
__global__ __launch_bounds__(256) void k_consume(const Args a) { consume(a, blockIdx.x, threadIdx.x); }
__global__ __launch_bounds__(256) void k_consume_own(const Args a)       // the middle body inside every consumer
{
    __shared__ unsigned s_seed[512];
    const unsigned tid = threadIdx.x;
    const unsigned acc = gather_records(a, tid);
    s_seed[tid] = a.epoch * 7u + tid + (acc & 0u);
    s_seed[256u + tid] = a.epoch * 7u + 256u + tid;
    __syncthreads();
    unsigned bad = 0;
    for (unsigned i = tid; i < 320u; i += 256u) {
        const unsigned v = a.codes[blockIdx.x * 320u + i];
        bad |= v != a.epoch * 1000003u + blockIdx.x * 320u + i;
        a.out[blockIdx.x * 320u + i] = v + s_seed[tid];
    }
    if (bad) atomicAdd(&a.sync[192], 1u);
}

int main(int argc, char **argv)
{
    const unsigned T = argc > 1 ? (unsigned)atoi(argv[1]) : 391u;
    const double work_us = argc > 2 ? atof(argv[2]) : 7.0;
    Args a;
    a.T = T;
    a.work_ticks = (unsigned)(work_us * 100.0);                // wall_clock64: 100 MHz
    CHECK(hipMalloc(&a.rec, (size_t)T * 32));
    CHECK(hipMalloc(&a.codes, (size_t)T * 1280));
    CHECK(hipMalloc(&a.out, (size_t)T * 1280));
    CHECK(hipMalloc(&a.seeds, 2048));
    CHECK(hipMalloc(&a.sync, 1024));
    CHECK(hipMemset(a.sync, 0, 1024));
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int N = 300;
    unsigned epoch = 0;
    auto run = [&](const char *name, auto launch) {
        for (int i = 0; i < 20; ++i) { a.epoch = ++epoch; launch(); }
        CHECK(hipStreamSynchronize(s));
        CHECK(hipEventRecord(e0, s));
        for (int i = 0; i < N; ++i) { a.epoch = ++epoch; launch(); }
        CHECK(hipEventRecord(e1, s));
        CHECK(hipStreamSynchronize(s));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned sync[4];
        CHECK(hipMemcpy(&sync[3], a.sync + 192, 4, hipMemcpyDeviceToHost));
        printf("%-62s %7.2f us per step   (wrong or stale reads: %u)\n", name, ms * 1e3 / N, sync[3]);
        CHECK(hipMemset(a.sync, 0, 1024));
    };
    printf("tiles %u, producer body %.1f us\n", T, work_us);
    run("three launches (produce / one-workgroup sort / consume)", [&] {
        hipLaunchKernelGGL(k_produce, dim3(T), dim3(256), 0, s, a);
        hipLaunchKernelGGL(k_sort, dim3(1), dim3(256), 0, s, a);
        hipLaunchKernelGGL(k_consume, dim3(T), dim3(256), 0, s, a);
    });
    run("two launches (produce / consume with the sort's reads inside)", [&] {
        hipLaunchKernelGGL(k_produce, dim3(T), dim3(256), 0, s, a);
        hipLaunchKernelGGL(k_consume_own, dim3(T), dim3(256), 0, s, a);
    });
    run("one launch, agent acquire on the sorter and on every consumer", [&] { hipLaunchKernelGGL(one_launch<true>, dim3(2 * T), dim3(256), 0, s, a); });
    run("one launch, NO acquire (plain loads behind the poll: may be stale)", [&] { hipLaunchKernelGGL(one_launch<false>, dim3(2 * T), dim3(256), 0, s, a); });
    run("producers alone", [&] { hipLaunchKernelGGL(k_produce, dim3(T), dim3(256), 0, s, a); });
    return 0;
}
