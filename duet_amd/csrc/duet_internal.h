// duet_internal.h -- host-side state shared by the translation units of libduet_ef.so (not part of the ABI).
#ifndef DUET_INTERNAL_H
#define DUET_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "duet_ef.h"

struct DevBuf {
    void *ptr = nullptr;
    size_t cap = 0;
};

struct duet_ctx {
    int device = 0;
    std::string err;
    int profiling = 0;                     // 0 off, 1 ef_classify events, 2 every kernel, 3 ef_classify on every 8th run
    int ev_mode = 0;                       // mode the pooled events were recorded with
    uint32_t prof_tick = 0;                // run counter of the sampled mode
    uint32_t dbg = 0;
    uint32_t ef_heavy_t = 32;              // ef_classify: candidates with more marks take the wave-cooperative walk (DUET_EF_HEAVY_T overrides)
    unsigned long long *d_stamps = nullptr;
    hipStream_t own_stream = nullptr;
    uint64_t *d_untagged = nullptr;                           // one all-ones word: what ef_classify gathers for a mark without a tag
    uint32_t *rx_dtot = nullptr;                              // [256] digit totals of a radix pass (zero between passes)
    // E/F plan (workspace keyed by the contig layout)
    std::vector<uint32_t> plan_off;        // cached cand_ctg_off
    uint32_t plan_C = 0;
    hipStream_t plan_stream = (hipStream_t)-1;   // the stream the plan was built on
    uint32_t *plan_stage = nullptr;        // pinned staging block for the plan uploads
    size_t plan_stage_cap = 0;
    hipEvent_t plan_ev = nullptr;          // recorded after the staging block has been read
    DevBuf ws_small;                        // ctg_off | n_one | status | blk_ctg | blk_cnt
    DevBuf ws_start, ws_ent, ws_one, ws_tmp, ws_c2;
    uint32_t *d_ctg_off = nullptr, *d_n_one = nullptr, *d_status = nullptr, *d_blk_ctg = nullptr,
             *d_blk_cnt = nullptr;
    // host-run staging
    DevBuf h_in[9], h_out[2];
    // clustering (A0) workspace and host-run staging
    DevBuf cl_ws[16], cl_in[4], cl_out[6];
    DevBuf rows_ws[8];                     // device-side row emission
    DevBuf rows_in[5];                     // host-array entry: uploaded text pool, offsets, ranks, sign flags; the rows
    DevBuf eval_ws;                        // evaluator (duet_eval.hip): one arena
    DevBuf sv_ws[5];
    DevBuf sv_in[3], sv_out[2];            // duet_svim_phase_host: mark read indices, read tags, depth bins; pred, ps
    std::vector<uint32_t> sv_depth_off;    // the depth offsets the device copy in sv_ws[0] holds (uploaded only when they change)
    hipStream_t sv_depth_off_stream = nullptr;             // ... and the stream that upload is ordered on (compared, never used: the
                                                           // caller may have destroyed it since)
    hipEvent_t sv_depth_off_ev = nullptr;                  // ... recorded behind that upload: what a run on another stream waits for
    void *sv_depth_off_at = nullptr;                       // fused SVIM-mode pipeline: contig offsets, adapted columns, gathered marks
    hipStream_t cl_side[3] = {nullptr, nullptr, nullptr};     // the size classes of A0 agglomerate side by side
    hipEvent_t cl_fork = nullptr, cl_join[3] = {nullptr, nullptr, nullptr};
    uint32_t *cl_flags = nullptr;                             // [16] device words: what the side streams' gate kernels wait for (stage A0's forks, round 6)
    int cl_gates = 0;                                         // 0: not tried yet; 1: a gate on one stream sees a signal from another (the device runs them side by side); -1: it does not
    uint32_t cl_epoch = 0;                                    // ... the value the current run's forks write there
    // profiling events: 6 per run
    std::vector<hipEvent_t> ev_pool;
    std::vector<uint8_t> ev_kmask;         // per profiled run: the kernels that ran (bit i = kernel i; the two-launch E/F has no ef_seed_sort)
    size_t ev_used = 0;
    // the two-launch E/F (ef_finalize_own) leaves no ascending seed arrays behind: what duet_ef_get_seed_ps / duet_ef_stats need is
    // made on demand by ef_seed_sort from the same seed entries, with the last run's kernel arguments on the last run's stream
    std::vector<unsigned char> ef_last_params;
    hipStream_t ef_last_stream = nullptr;
    bool ef_seeds_stale = false;
    bool pending_check = false;
};

extern thread_local std::string duet_g_last_error;

inline int duet_fail(duet_ctx *ctx, int code, const std::string &msg)
{
    if (ctx) ctx->err = msg;
    duet_g_last_error = msg;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return duet_fail(ctx, e_ == hipErrorOutOfMemory ? DUET_ERR_OOM : DUET_ERR_HIP,      \
                             std::string(#expr) + ": " + hipGetErrorString(e_));                \
    } while (0)

// device-planned E/F run (duet_ef.hip), for the fused pipeline in duet_cluster.hip.  _prepare sizes the workspace for the bound
// and names the plan's place in it: ctg_off[K + 1], and K + 8 words to zero (seeds per contig, status words) -- a producer that
// writes both itself passes planned = true and no plan kernel is launched
int duet_ef_plan_on_device_prepare(duet_ctx *ctx, uint32_t K, uint32_t c_max, hipStream_t stream, uint32_t **ctg_off_out,
                                   uint32_t **zero_out);
int duet_ef_run_planned_on_device(duet_ctx *ctx, const duet_ef_problem *pr, uint32_t c_max, const uint32_t *d_n_cands,
                                  const uint32_t *d_ctg_off /* or null ... */, const uint16_t *d_cand_contig /* ... then the candidates' contig column */,
                                  uint8_t *out_pred, uint32_t *out_ps, hipStream_t stream, bool planned);

int duet_ef_materialise_seeds(duet_ctx *ctx);

// argument checks of an E/F problem (host or device arrays alike: pointers and counts only)
int duet_ef_validate(duet_ctx *ctx, const duet_ef_problem *pr);

// host arrays of an E/F problem -> the context's staging buffers (duet_ef.hip)
int duet_ef_upload(duet_ctx *ctx, const duet_ef_problem *pr, duet_ef_problem *d, hipStream_t s);

inline int duet_reserve(duet_ctx *ctx, DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap) return DUET_OK;
    if (b.ptr) HIP_TRY(ctx, hipFree(b.ptr));
    b.ptr = nullptr;
    b.cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    HIP_TRY(ctx, hipMalloc(&b.ptr, want));
    b.cap = want;
    return DUET_OK;
}

#endif
