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
#include <functional>
#include <mutex>
#include <string>

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
constexpr int kNcclUint8 = 1;

struct Rccl {
    void *handle = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_all_gather all_gather = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_error_string error_string = nullptr;
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
    DevBuf send, recv;
};

extern "C" {

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
    NcclComm c = nullptr;
    const int rc = r.comm_init_rank(&c, world, u, rank);
    if (rc) {
        duet_fail(ctx, DUET_ERR_HIP, nccl_text(r, "ncclCommInitRank", rc));
        return nullptr;
    }
    duet_comm *cm = new duet_comm();
    cm->ctx = ctx; cm->comm = c; cm->rank = rank; cm->world = world;
    return cm;
}

int duet_comm_allgather_device(duet_comm *cm, const void *send_dev, uint64_t bytes, void *recv_dev, void *stream_)
{
    if (!cm || !cm->ctx) return duet_fail(nullptr, DUET_ERR_INVALID, "null communicator");
    duet_ctx *ctx = cm->ctx;
    if (!send_dev || !recv_dev || bytes == 0) return duet_fail(ctx, DUET_ERR_INVALID, "null buffer");
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
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return DUET_OK;
}

void duet_comm_destroy(duet_comm *cm)
{
    if (!cm) return;
    Rccl &r = rccl();
    if (cm->ctx) (void)hipSetDevice(cm->ctx->device);
    if (cm->comm && r.comm_destroy) (void)r.comm_destroy(cm->comm);
    if (cm->send.ptr) (void)hipFree(cm->send.ptr);
    if (cm->recv.ptr) (void)hipFree(cm->recv.ptr);
    delete cm;
}

}  // extern "C"
