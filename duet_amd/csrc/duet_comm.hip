// duet_comm.hip -- the ONE collective of the contig-sharded product path inside the library ("one
// ncclAllGather of dense fixed-size records"; include/duet_ef.h, duet_comm_*).
//
// Round 3's ranks did the all-gather through torch.distributed: every rank process imported torch (about a second) and went
// through its rendezvous before any work started -- the fixed cost that made `duet --gpus N` slower than one GPU at every
// BASELINE size.  Here a rank needs numpy, this library and RCCL: the communicator is created with ncclCommInitRank from a
// unique id the caller hands around (duet_amd/comm.py: a TCP star on MASTER_ADDR:MASTER_PORT), the all-gather runs on device
// buffers of the context's device on the context's stream.  RCCL is loaded at run time (dlopen): the library itself stays
// loadable -- and every single-GPU path usable -- on a machine without it; the comm calls then fail loudly.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <unistd.h>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "duet_ef.h"
#include "duet_internal.h"

namespace {

// (the few declarations of rccl.h that are used: the library is not linked against RCCL)
struct NcclUniqueId {
    char internal[128];
};
typedef void *NcclComm;
typedef int (*fn_get_unique_id)(NcclUniqueId *);
typedef int (*fn_comm_init_rank)(NcclComm *, int, NcclUniqueId, int);
typedef int (*fn_all_gather)(const void *, void *, size_t, int /* ncclDataType_t */, NcclComm, hipStream_t);
typedef int (*fn_comm_destroy)(NcclComm);
typedef const char *(*fn_error_string)(int);
typedef int (*fn_get_version)(int *);
typedef int (*fn_comm_int)(NcclComm, int *);
constexpr int kNcclUint8 = 1;

struct Rccl {
    void *handle = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_all_gather all_gather = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_error_string error_string = nullptr;
    fn_get_version get_version = nullptr;
    fn_comm_int comm_count = nullptr, comm_user_rank = nullptr, comm_cu_device = nullptr;   // (optional: duet_comm_info)
    std::string why;
};

void rccl_load(Rccl &r);

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;                            // (contexts of different host threads may get here together)
    std::call_once(once, rccl_load, std::ref(r));
    return r;
}

void rccl_load(Rccl &r)
{
    const char *names[3] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (r.handle) break;
    }
    if (!r.handle) {
        const char *why = dlerror();
        r.why = std::string("RCCL is not available (dlopen librccl.so.1: ") + (why ? why : "?") + ")";
        return;
    }
    r.get_unique_id = (fn_get_unique_id)dlsym(r.handle, "ncclGetUniqueId");
    r.comm_init_rank = (fn_comm_init_rank)dlsym(r.handle, "ncclCommInitRank");
    r.all_gather = (fn_all_gather)dlsym(r.handle, "ncclAllGather");
    r.comm_destroy = (fn_comm_destroy)dlsym(r.handle, "ncclCommDestroy");
    r.error_string = (fn_error_string)dlsym(r.handle, "ncclGetErrorString");
    r.get_version = (fn_get_version)dlsym(r.handle, "ncclGetVersion");
    r.comm_count = (fn_comm_int)dlsym(r.handle, "ncclCommCount");
    r.comm_user_rank = (fn_comm_int)dlsym(r.handle, "ncclCommUserRank");
    r.comm_cu_device = (fn_comm_int)dlsym(r.handle, "ncclCommCuDevice");
    if (!r.get_unique_id || !r.comm_init_rank || !r.all_gather || !r.comm_destroy) {
        r.why = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy";
        r.handle = nullptr;
    }
}

std::string nccl_text(Rccl &r, const char *what, int code)
{
    return std::string(what) + ": " + (r.error_string ? r.error_string(code) : "RCCL error") + " (" + std::to_string(code) + ")";
}

}  // namespace

struct duet_comm {
    duet_ctx *ctx = nullptr;
    NcclComm comm = nullptr;
    int rank = 0, world = 1;
    double timeout_s = 300.0;                              // bound of every blocking step (DUET_RDZV_TIMEOUT at creation)
    bool broken = false;                                   // a step timed out: nothing more goes through this communicator
    DevBuf send, recv, slots;
};

namespace {

double env_timeout()
{
    const char *t = getenv("DUET_RDZV_TIMEOUT");
    const double v = t ? atof(t) : 0.0;
    return v > 0.0 ? v : 300.0;
}

// Wait for everything queued on `s`, but not for ever: the stream is polled (hipStreamQuery) against a deadline.  A collective
// whose peer never arrives leaves the stream busy for good; the caller gets DUET_ERR_TIMEOUT and is expected to exit.
int bounded_sync(duet_comm *cm, hipStream_t s, const char *what)
{
    duet_ctx *ctx = cm->ctx;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; ++spin) {
        const hipError_t e = hipStreamQuery(s);
        if (e == hipSuccess) return DUET_OK;
        if (e != hipErrorNotReady) return duet_fail(ctx, DUET_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (dt > cm->timeout_s) {
            cm->broken = true;
            return duet_fail(ctx, DUET_ERR_TIMEOUT, std::string(what) + " did not finish within " + std::to_string(cm->timeout_s) +
                                                     " s (a rank is missing or stuck)");
        }
        if (spin > 2000) usleep(dt < 0.01 ? 20 : 200);     // the first microseconds by spinning: a 2 MB gather takes tens of them
    }
}

// The tail of a rank's record block: status word and the rows every CHROM-text slot keeps (pred != 0), counted where pred lives.
// A workgroup takes 4096 consecutive candidates (contig-sorted: a handful of distinct slots), counts in LDS when the slots fit
// (wave-aggregated: one LDS add per distinct slot and wave), and adds its non-zero counters to the block's.
constexpr uint32_t kSlotLds = 4096;
__global__ __launch_bounds__(256) void comm_trailer(const uint8_t *__restrict__ pred, const uint32_t *__restrict__ slot, uint32_t C,
                                                    uint32_t n_slots, uint32_t *status_out, unsigned long long *kept, uint32_t *d_status)
{
    __shared__ uint32_t s_cnt[kSlotLds];
    const uint32_t tid = threadIdx.x;
    if (blockIdx.x == 0 && tid == 0) {
        status_out[0] = (d_status && (d_status[0] & 1u)) ? 5u : 0u;        // duet_amd/multi.py: RC_DIV_ZERO
        if (d_status) d_status[0] = 0;
    }
    if (n_slots == 0 || C == 0) return;
    const bool lds = n_slots <= kSlotLds;
    if (lds) {
        for (uint32_t i = tid; i < n_slots; i += 256u) s_cnt[i] = 0;
        __syncthreads();
    }
    const uint32_t c0 = blockIdx.x * 4096u;
    for (uint32_t i = 0; i < 16u; ++i) {
        const uint32_t c = c0 + i * 256u + tid;
        bool todo = c < C && pred[c] != 0;
        const uint32_t sl = todo ? slot[c] : 0u;
        if (todo && sl >= n_slots) todo = false;                            // (the host validates; never index outside)
        unsigned long long left = __ballot(todo);
        while (left) {
            const uint32_t lead = (uint32_t)__ffsll((long long)left) - 1u;
            const uint32_t sv = (uint32_t)__builtin_amdgcn_readlane((int)sl, lead);
            const unsigned long long same = __ballot(todo && sl == sv);
            if ((tid & 63u) == lead) {
                if (lds) atomicAdd(&s_cnt[sv], (uint32_t)__popcll(same));
                else atomicAdd(&kept[sv], (unsigned long long)__popcll(same));
            }
            left &= ~same;
        }
    }
    if (lds) {
        __syncthreads();
        for (uint32_t i = tid; i < n_slots; i += 256u)
            if (s_cnt[i]) atomicAdd(&kept[i], (unsigned long long)s_cnt[i]);
    }
}

}  // namespace

extern "C" {

int duet_comm_rccl_version(duet_ctx *ctx)
{
    Rccl &r = rccl();
    if (!r.handle) return duet_fail(ctx, DUET_ERR_NO_DEVICE, r.why);
    int v = 0;
    if (!r.get_version || r.get_version(&v)) return duet_fail(ctx, DUET_ERR_HIP, "ncclGetVersion failed");
    return v;
}

int duet_comm_unique_id(duet_ctx *ctx, unsigned char *id)
{
    if (!ctx || !id) return duet_fail(ctx, DUET_ERR_INVALID, "null argument");
    Rccl &r = rccl();
    if (!r.handle) return duet_fail(ctx, DUET_ERR_NO_DEVICE, r.why);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    NcclUniqueId u;
    static_assert(sizeof(u) == DUET_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    const int rc = r.get_unique_id(&u);
    if (rc) return duet_fail(ctx, DUET_ERR_HIP, nccl_text(r, "ncclGetUniqueId", rc));
    memcpy(id, &u, sizeof(u));
    return DUET_OK;
}

duet_comm *duet_comm_create(duet_ctx *ctx, const unsigned char *id, int rank, int world)
{
    if (!ctx || !id || world < 1 || rank < 0 || rank >= world) {
        duet_fail(ctx, DUET_ERR_INVALID, "duet_comm_create: bad argument");
        return nullptr;
    }
    Rccl &r = rccl();
    if (!r.handle) {
        duet_fail(ctx, DUET_ERR_NO_DEVICE, r.why);
        return nullptr;
    }
    if (hipSetDevice(ctx->device) != hipSuccess) {
        duet_fail(ctx, DUET_ERR_HIP, "hipSetDevice failed");
        return nullptr;
    }
    NcclUniqueId u;
    memcpy(&u, id, sizeof(u));
    // ncclCommInitRank blocks until every rank of `world` has called it -- for ever when one never does.  It runs on a helper
    // thread; this one waits for it with a deadline.  On expiry the helper is left behind (it cannot be cancelled) and the
    // caller is told to give up: a rank process then exits non-zero and its launcher ends the others.
    struct InitJob {
        std::mutex m;
        std::condition_variable cv;
        bool done = false;
        int rc = 0;
        NcclComm c = nullptr;
    };
    auto job = std::make_shared<InitJob>();
    const int device = ctx->device;
    fn_comm_init_rank init = r.comm_init_rank;
    std::thread([job, init, u, world, rank, device]() {
        NcclComm c = nullptr;
        int rc = hipSetDevice(device) == hipSuccess ? init(&c, world, u, rank) : -1;
        std::lock_guard<std::mutex> g(job->m);
        job->rc = rc;
        job->c = c;
        job->done = true;
        job->cv.notify_all();
    }).detach();
    const double limit = env_timeout();
    {
        std::unique_lock<std::mutex> g(job->m);
        if (!job->cv.wait_for(g, std::chrono::duration<double>(limit), [&] { return job->done; })) {
            duet_fail(ctx, DUET_ERR_TIMEOUT, "ncclCommInitRank did not finish within " + std::to_string(limit) +
                                                 " s (DUET_RDZV_TIMEOUT): rank " + std::to_string(rank) + " of " + std::to_string(world) +
                                                 " is waiting for a rank that never arrived");
            return nullptr;
        }
    }
    if (job->rc) {
        duet_fail(ctx, DUET_ERR_HIP, job->rc == -1 ? std::string("hipSetDevice failed on the set-up thread") : nccl_text(r, "ncclCommInitRank", job->rc));
        return nullptr;
    }
    duet_comm *cm = new duet_comm();
    cm->ctx = ctx; cm->comm = job->c; cm->rank = rank; cm->world = world;
    cm->timeout_s = limit;
    return cm;
}

int duet_comm_set_timeout(duet_comm *cm, double seconds)
{
    if (!cm || !(seconds > 0.0)) return duet_fail(cm ? cm->ctx : nullptr, DUET_ERR_INVALID, "duet_comm_set_timeout: bad argument");
    cm->timeout_s = seconds;
    return DUET_OK;
}

int duet_comm_allgather_device(duet_comm *cm, const void *send_dev, uint64_t bytes, void *recv_dev, void *stream_)
{
    if (!cm || !cm->ctx) return duet_fail(nullptr, DUET_ERR_INVALID, "null communicator");
    duet_ctx *ctx = cm->ctx;
    if (!send_dev || !recv_dev || bytes == 0) return duet_fail(ctx, DUET_ERR_INVALID, "null buffer");
    if (cm->broken) return duet_fail(ctx, DUET_ERR_TIMEOUT, "the communicator timed out earlier");
    Rccl &r = rccl();
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int rc = r.all_gather(send_dev, recv_dev, (size_t)bytes, kNcclUint8, cm->comm, (hipStream_t)stream_);
    if (rc) return duet_fail(ctx, DUET_ERR_HIP, nccl_text(r, "ncclAllGather", rc));
    return DUET_OK;
}

int duet_comm_allgather_host(duet_comm *cm, const void *send_host, uint64_t bytes, void *recv_host)
{
    if (!cm || !cm->ctx) return duet_fail(nullptr, DUET_ERR_INVALID, "null communicator");
    duet_ctx *ctx = cm->ctx;
    if (!send_host || !recv_host || bytes == 0) return duet_fail(ctx, DUET_ERR_INVALID, "null buffer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = duet_reserve(ctx, cm->send, (size_t)bytes))) return rc;
    if ((rc = duet_reserve(ctx, cm->recv, (size_t)bytes * (size_t)cm->world))) return rc;
    hipStream_t s = ctx->own_stream;
    HIP_TRY(ctx, hipMemcpyAsync(cm->send.ptr, send_host, (size_t)bytes, hipMemcpyHostToDevice, s));
    if ((rc = duet_comm_allgather_device(cm, cm->send.ptr, bytes, cm->recv.ptr, s))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(recv_host, cm->recv.ptr, (size_t)bytes * (size_t)cm->world, hipMemcpyDeviceToHost, s));
    return bounded_sync(cm, s, "the all-gather");
}

uint64_t duet_comm_block_bytes(uint32_t n_max, uint32_t n_slots)
{
    return ((5ull * n_max + 15ull) / 16ull) * 16ull + 16ull + 8ull * n_slots;
}

int duet_comm_ef_allgather(duet_comm *cm, const duet_ef_problem *pr, const uint32_t *cand_slot, uint32_t n_slots, uint32_t n_max,
                           uint8_t *gathered)
{
    if (!cm || !cm->ctx) return duet_fail(nullptr, DUET_ERR_INVALID, "null communicator");
    duet_ctx *ctx = cm->ctx;
    if (!pr || !gathered || n_max == 0) return duet_fail(ctx, DUET_ERR_INVALID, "duet_comm_ef_allgather: null argument");
    if (cm->broken) return duet_fail(ctx, DUET_ERR_TIMEOUT, "the communicator timed out earlier");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t rb = (size_t)(((5ull * n_max + 15ull) / 16ull) * 16ull), bytes = (size_t)duet_comm_block_bytes(n_max, n_slots);
    int rc;
    // the block and the receive buffer first: without them this rank cannot take part in the collective at all
    if ((rc = duet_reserve(ctx, cm->send, bytes))) return rc;
    if ((rc = duet_reserve(ctx, cm->recv, bytes * (size_t)cm->world))) return rc;
    hipStream_t s = ctx->own_stream;
    uint8_t *block = (uint8_t *)cm->send.ptr;
    // The rank's own part.  Whatever fails in it -- the arguments (checked BEFORE any array is touched: ADVICE round 5), memory, the
    // upload, a launch -- this rank still contributes a block (zeros, and a status word with the top bit set) to the ONE collective:
    // its peers then fail at once with that status instead of waiting for DUET_RDZV_TIMEOUT.
    auto local = [&]() -> int {
        int rc2;
        if ((rc2 = duet_ef_validate(ctx, pr))) return rc2;
        const uint32_t C = pr->n_cands;
        if (C > n_max) return duet_fail(ctx, DUET_ERR_INVALID, "the shard has more candidates than n_max");
        if (n_slots && C && !cand_slot) return duet_fail(ctx, DUET_ERR_INVALID, "cand_slot is null");
        if ((rc2 = duet_reserve(ctx, cm->slots, C ? (size_t)C * 4 : 16))) return rc2;
        // what the kernels do not write of the block: the slots of the ranks with more candidates, the padding, the trailer's counters
        HIP_TRY(ctx, hipMemsetAsync(block, 0, bytes, s));
        if (C) {
            duet_ef_problem d;
            if ((rc2 = duet_ef_upload(ctx, pr, &d, s))) return rc2;
            if (n_slots) HIP_TRY(ctx, hipMemcpyAsync(cm->slots.ptr, cand_slot, (size_t)C * 4, hipMemcpyHostToDevice, s));
            if ((rc2 = duet_ef_run_device(ctx, &d, block + 4 * (size_t)n_max, (uint32_t *)block, s))) return rc2;
        }
        hipLaunchKernelGGL(comm_trailer, dim3(C ? (C + 4095u) / 4096u : 1u), dim3(256), 0, s, block + 4 * (size_t)n_max,
                           (const uint32_t *)cm->slots.ptr, C, n_slots, (uint32_t *)(block + rb), (unsigned long long *)(block + rb + 16),
                           C ? ctx->d_status : (uint32_t *)nullptr);
        HIP_TRY(ctx, hipGetLastError());
        ctx->pending_check = false;                         // (the status word went into the block)
        return DUET_OK;
    };
    const int local_rc = local();
    std::string local_why;
    if (local_rc) {
        local_why = ctx->err;
        const uint32_t st = DUET_COMM_STATUS_RANK_FAILED | ((uint32_t)(-local_rc) & 0xFFFFu);
        if (hipMemsetAsync(block, 0, bytes, s) != hipSuccess ||
            hipMemcpyAsync(block + rb, &st, 4, hipMemcpyHostToDevice, s) != hipSuccess)
            return duet_fail(ctx, local_rc, local_why);     // (the device itself is gone: the peers' timeout is all that is left)
    }
    if ((rc = duet_comm_allgather_device(cm, block, bytes, cm->recv.ptr, s))) return local_rc ? duet_fail(ctx, local_rc, local_why) : rc;      // the ONE collective of the path
    HIP_TRY(ctx, hipMemcpyAsync(gathered, cm->recv.ptr, bytes * (size_t)cm->world, hipMemcpyDeviceToHost, s));
    rc = bounded_sync(cm, s, "the rank's kernels and the all-gather");
    if (local_rc) return duet_fail(ctx, local_rc, local_why);
    return rc;
}

int duet_comm_info(duet_comm *cm, int *rank, int *world, int *rccl_ranks, int *rccl_rank, int *rccl_device)
{
    if (!cm || !cm->ctx) return duet_fail(nullptr, DUET_ERR_INVALID, "null communicator");
    Rccl &r = rccl();
    if (rank) *rank = cm->rank;
    if (world) *world = cm->world;
    int n = -1, u = -1, d = -1;
    if (cm->comm && !cm->broken) {
        if (r.comm_count && r.comm_count(cm->comm, &n)) n = -1;
        if (r.comm_user_rank && r.comm_user_rank(cm->comm, &u)) u = -1;
        if (r.comm_cu_device && r.comm_cu_device(cm->comm, &d)) d = -1;
    }
    if (rccl_ranks) *rccl_ranks = n;
    if (rccl_rank) *rccl_rank = u;
    if (rccl_device) *rccl_device = d;
    return DUET_OK;
}

// A pattern that names its rank and its place: what every slot of the gathered buffer must hold.
static inline uint32_t selftest_word(uint32_t rank, uint32_t i) { return (rank + 1u) * 0x9E3779B1u ^ (i * 0x85EBCA6Bu + 0xC2B2AE35u); }

int duet_comm_selftest(duet_comm *cm, uint32_t words)
{
    if (!cm || !cm->ctx) return duet_fail(nullptr, DUET_ERR_INVALID, "null communicator");
    duet_ctx *ctx = cm->ctx;
    if (words == 0 || words > (1u << 24)) return duet_fail(ctx, DUET_ERR_INVALID, "duet_comm_selftest: 1 .. 2^24 words");
    std::vector<uint32_t> mine(words), all((size_t)words * (size_t)cm->world);
    for (uint32_t i = 0; i < words; ++i) mine[i] = selftest_word((uint32_t)cm->rank, i);
    int rc = duet_comm_allgather_host(cm, mine.data(), (uint64_t)words * 4u, all.data());
    if (rc) return rc;
    for (int r = 0; r < cm->world; ++r)
        for (uint32_t i = 0; i < words; ++i)
            if (all[(size_t)r * words + i] != selftest_word((uint32_t)r, i))
                return duet_fail(ctx, DUET_ERR_HIP, "duet_comm_selftest: rank " + std::to_string(cm->rank) + " received a wrong word " +
                                                    std::to_string(i) + " in the slot of rank " + std::to_string(r));
    return DUET_OK;
}

void duet_comm_destroy(duet_comm *cm)
{
    if (!cm) return;
    Rccl &r = rccl();
    if (cm->ctx) (void)hipSetDevice(cm->ctx->device);
    // A communicator that timed out may still have a collective stuck on the stream: ncclCommDestroy would block on it, and so
    // would hipFree (it synchronises the device) -- its device buffers are leaked, the process is expected to exit
    // (include/duet_ef.h says so)
    if (!cm->broken) {
        if (cm->comm && r.comm_destroy) (void)r.comm_destroy(cm->comm);
        if (cm->send.ptr) (void)hipFree(cm->send.ptr);
        if (cm->recv.ptr) (void)hipFree(cm->recv.ptr);
        if (cm->slots.ptr) (void)hipFree(cm->slots.ptr);
    }
    delete cm;
}

}  // extern "C"
