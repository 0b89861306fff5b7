/*
 * duet_ef.h -- C ABI of the MI355X (gfx950) implementation of Duet's step E/F:
 * integration of read-haplotype tags with SV support-read marks, per-candidate haplotype vote and
 * the T1-T5 threshold decision.
 *
 * The reference (yekaizhou/duet v0.6) is pure Python and has no FFI; the seam this library sits
 * behind is the body of
 *     generate_phased_callset(vcf_path, sam_home, svlen_thres, suppread_thres, thread, include_all_ctgs)
 *                                                                   src/duet/sv_phasing_fn.py:185-230
 * between "callset built" (generate_callinfo, :36-68) and "rows sorted" (:229):
 *     filter            sv_phasing_fn.py:189-190
 *     PS-class          sv_phasing_fn.py:191-194
 *     seed sets         sv_phasing_fn.py:195-203
 *     vote + features   sv_phasing_fn.py:70-140   (get_phase_info)
 *     decision          sv_phasing_fn.py:142-183  (predict_hp)
 *     contig drop       sv_phasing_fn.py:209-210
 * and the join of mark names against the per-contig tag dict (:46-48), which the host performs
 * while flattening names to indices.
 *
 * Conventions: plain pointers and sizes, no C++ or torch types; no exception crosses the ABI; every
 * call returns DUET_OK (0) or a negative duet_status and duet_last_error() describes the failure.
 * A context is driven by one host thread at a time.
 */
#ifndef DUET_EF_H
#define DUET_EF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DUET_ABI_VERSION 1

/* The library is built with -fvisibility=hidden: only the functions declared in this header are exported. */
#define DUET_API __attribute__((visibility("default")))

typedef enum duet_status {
    DUET_OK = 0,
    DUET_ERR_INVALID = -1,    /* bad argument (NULL pointer, inconsistent sizes) */
    DUET_ERR_NO_DEVICE = -2,  /* no usable gfx950 device / HIP runtime failure at context creation */
    DUET_ERR_HIP = -3,        /* a HIP call failed; duet_last_error has hipGetErrorString */
    DUET_ERR_OOM = -4,        /* device allocation failed */
    DUET_ERR_TIMEOUT = -6,    /* a collective step (communicator set-up, the all-gather) did not finish within the communicator's
                                 time limit: a peer is missing or stuck.  The communicator is unusable afterwards and the process
                                 should exit (a helper thread may still sit inside RCCL) */
    DUET_ERR_DIV_ZERO = -5    /* a candidate that reaches the decision has svread + refread == 0: the
                                 reference raises ZeroDivisionError there (sv_phasing_fn.py:123) */
} duet_status;

/*
 * Read tag word (one per read that is in its contig's tag dict, sv_phasing_fn.py:28-29):
 *     bits 63..62  hap   1 or 2 (HP:i); 3 = any other value (never counted as a haplotype vote)
 *     bits 61..32  pc    PC:i, saturated at 2^30-2 (only `pc <= 8100` and sums of such pc are used)
 *     bits 31..0   ps    PS:i
 */
#define DUET_TAG(hap, pc, ps) (((uint64_t)(hap) << 62) | ((uint64_t)(pc) << 32) | (uint64_t)(uint32_t)(ps))
#define DUET_MARK_ABSENT 0xFFFFFFFFu   /* mark whose read name is not in its contig's tag dict */
#define DUET_PC_MAX 8100u              /* sv_phasing_fn.py:76,88,201 */

/*
 * One E/F problem in structure-of-arrays form.  Candidates are in callset order
 * (generate_callinfo, sv_phasing_fn.py:50-67): contig-major in chrom-list order, file order inside a
 * contig; cand_ctg_off[k]..cand_ctg_off[k+1] are the candidates of contig k.
 * Marks of candidate c are mark_read[cand_off[c] .. cand_off[c+1]) in RNAMES/READS list order
 * (duplicates kept), each an index into read_tag (already resolved against the candidate's OWN
 * contig table, sv_phasing_fn.py:47-48) or DUET_MARK_ABSENT.  Every candidate has >= 1 mark.
 *
 * cand_ctg_off is ALWAYS a host pointer (K+1 small integers; the library stages it).  The other
 * arrays are device pointers for duet_ef_run_device and host pointers for duet_ef_run_host.
 */
typedef struct duet_ef_problem {
    uint32_t n_contigs;             /* K */
    uint32_t n_cands;               /* C */
    uint32_t n_marks;               /* M = cand_off[C] */
    uint32_t n_reads;               /* R = length of read_tag */
    const uint32_t *cand_ctg_off;   /* [K+1] HOST */
    const uint64_t *read_tag;       /* [R]   */
    const uint32_t *cand_pos;       /* [C]   VCF POS */
    const uint32_t *cand_svlen;     /* [C]   abs(SVLEN) (sv_phasing_fn.py:62) */
    const uint32_t *cand_svread;    /* [C]   INFO support count (read_file.py:40-47) */
    const uint32_t *cand_refread;   /* [C]   reference-read count as the reference derives it (read_file.py:56-76) */
    const uint8_t *cand_gt_ok;      /* [C]   1 unless the genotype string is exactly "./." (sv_phasing_fn.py:190) */
    const uint32_t *cand_off;       /* [C+1] CSR offsets into mark_read */
    const uint32_t *mark_read;      /* [M]   */
    uint32_t svlen_thres;           /* -s / --sv_min_size */
    uint32_t suppread_thres;        /* -r / --min_support_read */
} duet_ef_problem;

/* Kernels of one run, in launch order. */
enum { DUET_K_CLASSIFY = 0, DUET_K_SEEDS = 1, DUET_K_FINALIZE = 2, DUET_N_KERNELS = 3 };

typedef struct duet_ef_stats {
    uint64_t algorithmic_bytes;     /* 12*M + 27*C + 8*R */
    uint32_t n_seed_ps;             /* distinct seed phase sets over all contigs (valid after a host run) */
    uint32_t n_profiled_runs;       /* runs averaged into kernel_ms (profiling mode only) */
    float kernel_ms[DUET_N_KERNELS];/* mean duration per kernel from HIP events on the run's stream */
    float total_ms;                 /* mean first-kernel-start to last-kernel-end */
} duet_ef_stats;

typedef struct duet_ctx duet_ctx;

DUET_API int duet_abi_version(void);

/* Context on HIP device `device_id`. NULL on failure; duet_last_error(NULL) then holds the reason. */
DUET_API duet_ctx *duet_ctx_create(int device_id);
DUET_API void duet_ctx_destroy(duet_ctx *ctx);
DUET_API const char *duet_last_error(const duet_ctx *ctx);

/* HIP events attached to the kernels' own dispatches (hipExtLaunchKernelGGL start/stop events on the run's
 * stream), resolved by duet_ef_profile_collect: 0 = none, 1 = the dominant kernel (ef_classify) only -- two
 * events per run --, 2 = every kernel (kernel_ms[] complete, total_ms = first start to last end; serialises
 * the stream a little more), 3 = like 1 but only on every 8th run (what bench.py uses inside its timed
 * region: event pairs cost ~6 us of stream time each, a quarter of a config-2 step). */
DUET_API int duet_ctx_set_profiling(duet_ctx *ctx, int mode);

/*
 * Run E/F on device-resident inputs, asynchronously on `stream` (a hipStream_t passed as void*, NULL =
 * the null stream).  out_pred[C] (0 = filtered, 1 = "1|0", 2 = "0|1", 3 = "1|1") and out_ps[C] are
 * device pointers.  out_ps[c] is the PS predict_hp returns when the reference calls it for c, else 0.
 * The call does not synchronise; DUET_ERR_DIV_ZERO is reported by duet_ef_check (or by the host run).
 */
DUET_API int duet_ef_run_device(duet_ctx *ctx, const duet_ef_problem *prob, uint8_t *out_pred, uint32_t *out_ps,
                       void *stream);

/* Synchronise `stream` and return the deferred status of the runs issued on this context since the
 * last check (DUET_OK or DUET_ERR_DIV_ZERO / DUET_ERR_HIP). */
DUET_API int duet_ef_check(duet_ctx *ctx, void *stream);

/* Host convenience: copies the arrays to the device, runs, copies the results back, synchronises.
 * stats may be NULL. */
DUET_API int duet_ef_run_host(duet_ctx *ctx, const duet_ef_problem *prob, uint8_t *out_pred, uint32_t *out_ps,
                     duet_ef_stats *stats);

/* Profiling mode: synchronise, average the per-kernel event timings of the runs since the last
 * collect into *stats, and reset. */
DUET_API int duet_ef_profile_collect(duet_ctx *ctx, duet_ef_stats *stats);

/* Diagnostics: low bits ablate E/F kernel phases (tools/ablate.py); DUET_DBG_CLUSTER_EXACT sends every A0
 * partition through the exact linkage instead of the bounding-box / threshold-graph fast paths (the outputs are
 * identical; tests use it to exercise both).  0 in production. */
#define DUET_DBG_EF_NO_SEED_HASH 0x40u   /* E/F: ef_seed_sort orders an unsorted seed list itself instead of taking its distinct values through a hash set first */
#define DUET_DBG_EF_FIN_TPB2 0x20u       /* E/F: ef_finalize takes two tiles of 256 candidates per workgroup whatever the size (default from 1 M candidates on) */
#define DUET_DBG_EF_FIN_TPB4 0x80u       /* ... four (default from 8 M candidates on) */
#define DUET_DBG_EF_HEAVY_ALL 0x80000u   /* E/F: ef_classify walks EVERY kept candidate wave-cooperatively (64 marks per step; default: those with more than 32 marks) */
#define DUET_DBG_EF_HEAVY_OFF 0x100000u  /* ... only those of more than 255 marks (the lane walk keeps its counts in bytes): every other candidate by its own lane, mark after mark */
#define DUET_DBG_EF_WALK_R4 0x200000u    /* E/F: ef_classify's lane walk with round 4's loop body (35 vector instructions per mark) instead of round 5's shorter one */
#define DUET_DBG_EF_FP_DECIDE 0x400000u  /* E/F: ef_classify takes every class-0 / class-1 decision through the binary64 expressions (rounds 1-4) instead of the
                                            integer form with the binary64 fallback */
#define DUET_DBG_EF_OWN_OFF 0x800000u     /* E/F: three launches (ef_classify, ef_seed_sort, ef_finalize) at every size; default up to 1024 tiles of 256 candidates
                                          * and 64 contigs: two -- every finalize tile builds its contig's seed set itself (ef_finalize_own) */
#define DUET_DBG_EF_OWN_ALL 0x1000000u    /* ... the two launches at every size (up to 64 contigs) */
#define DUET_DBG_EF_OWN_SMALLTAB 0x2000000u /* ... and their seed set in LDS holds 8 distinct seeds (default 2048): contigs with more take the array-free walk */
#define DUET_DBG_CLUSTER_EVENT_FORKS 0x4000000u /* A0: the side streams fork off behind hipEventRecord / hipStreamWaitEvent (rounds 1-5) instead of a signal kernel on the
                                          * main stream and a gate kernel on the side stream (round 6) */
#define DUET_DBG_CLUSTER_WIDE_OFF 0x8000000u /* A0: small inputs keep ONE wavefront per partition of more than 64 marks (rounds 1-5: cl_tight_big + cl_link_one) instead
                                              * of a workgroup of eight (cl_find_big + cl_wide_big: wide_unit) */
#define DUET_DBG_CLUSTER_WIDE_ALL 0x10000000u /* A0: small inputs send EVERY listed partition through the multi-wavefront units (cl_wide_list; tests: measured, they lose there) */
#define DUET_DBG_CLUSTER_EXACT 0x100u
#define DUET_DBG_CLUSTER_LARGE 0x200u   /* A0: take the launch structure of large inputs (> 4 M marks: one launch per size class,
                                           generic tile-offset scan in the sort, scans with a spine launch) whatever the size */
#define DUET_DBG_CLUSTER_PAIRS 0x400u   /* A0: sort (key, mark index) pairs even when the index fits the key's spare bits */
#define DUET_DBG_CLUSTER_NOBOX 0x800u   /* A0: no bounding-box test: every partition goes through the threshold-graph pair loops */
#define DUET_DBG_CLUSTER_KEYSORT 0x10000u /* A0: the key-only sort of rounds 1-3 (8-byte keys, the records gathered through the permutation afterwards)
                                            also where the 16-byte mark record could travel with the key (duet_recsort.hip.h) */
#define DUET_DBG_CLUSTER_RECSORT 0x40000u /* A0: the record sort also below 1.25 M marks (where the key-only sort is the default: launch-bound passes) */
#define DUET_DBG_CLUSTER_NOSYM 0x20000u  /* A0: the contracted linkage's pair tests column by column (every ordered pair) also where a unit holds one partition and
                                            could evaluate every unordered pair once */
#define DUET_DBG_CLUSTER_LSD 0x8000u     /* A0: plain LSD passes over all key bits also where small inputs would sort the low bits locally */
#define DUET_DBG_CLUSTER_SMALLCAP 0x4000u /* A0: the local sort of the low bits takes groups of at most 3 keys (default 128): the others go to the one-workgroup-per-group path */
#define DUET_DBG_CLUSTER_TIERS 0x2000u  /* A0: small inputs take the two-tier contracted linkage of large inputs in one launch per tier */
#define DUET_DBG_CLUSTER_KC2 0x1000u    /* A0: the contracted linkage keeps partitions with at most 2 groups (default 16); the others
                                           take the second lists (full-triangle linkage) */
DUET_API int duet_ctx_set_debug(duet_ctx *ctx, uint32_t flags);

/* Debug/inspection: copy contig k's sorted seed-PS array of the LAST run to `out` (capacity `cap`),
 * return its length (or a negative status). */
DUET_API int duet_ef_get_seed_ps(duet_ctx *ctx, uint32_t contig, uint32_t *out, uint32_t cap);

/* ------------------------------------------------------------------------------------------------------
 * Stage A0: span-position clustering of SV marks into candidates.
 *
 * Replaces what the reference delegates to the external `svim alignment ... --cluster_max_distance c`
 * (src/duet/sv_calling.py:13-15; help text src/duet/utils.py:27-28).  svim is not part of the reference
 * tree, so this stage follows this repository's own deterministic rule (oracle/cluster_oracle.c, DESIGN.md
 * section 9), modelled on SVIM 1.4.2: marks ordered by (contig, type, centre = pos + span/2); partitions
 * cut at a centre gap > part_gap or after part_max marks; span-position distance
 *     min(|dpos|, |dend|, |dcentre|) / normalizer + |dspan| / max(span)
 * in binary64; average linkage, merged while the closest pair is <= max_dist.
 * ---------------------------------------------------------------------------------------------------- */
typedef struct duet_cluster_problem {
    uint32_t n_marks;               /* M */
    uint32_t part_gap;              /* 1000 */
    uint32_t part_max;              /* 100 (1..128) */
    uint32_t n_contigs_hint;        /* the four hints only narrow the sort key; when any is 0 the library measures the
                                     * maxima itself (one small kernel + one host round trip) */
    uint32_t n_types_hint;          /* 0 = unknown */
    uint32_t max_pos_hint;          /* 0 = unknown */
    uint32_t max_span_hint;         /* 0 = unknown */
    uint32_t reserved;
    double max_dist;                /* -c / --cluster_max_distance (0.9) */
    double normalizer;              /* 900 */
    const uint16_t *mark_contig;    /* [M] */
    const uint8_t *mark_type;       /* [M] caller-defined SV type code */
    const uint32_t *mark_pos;       /* [M] */
    const uint32_t *mark_span;      /* [M] */
} duet_cluster_problem;

/* Candidates in (contig, type, centre) order; members of candidate j are
 * order[cand_off[j] .. cand_off[j+1]) (mark indices, sorted order); cand_pos / cand_span are floor means.
 * All arrays need room for M entries (cand_off: M+1).  n_cands is a device word for duet_cluster_run_device
 * and a host word for duet_cluster_run_host. */
typedef struct duet_cluster_result {
    uint32_t *order;
    uint32_t *cand_off;
    uint16_t *cand_contig;
    uint8_t *cand_type;
    uint32_t *cand_pos;
    uint32_t *cand_span;
    uint32_t *n_cands;
} duet_cluster_result;

/* (Up to 4 M marks the stage's side-stream chains fork off and join through one-lane signal / gate kernels and a per-call epoch word
 * of the context, not through events -- an event record cost the main stream 7-14 us each.  Such a call cannot be captured into a
 * hipGraph: set DUET_DBG_CLUSTER_EVENT_FORKS with duet_ctx_set_debug for that; it also applies to duet_svim_phase_device.
 * A gate waits for a kernel of another queue, which needs the device to run the two side by side: the first such call of a context
 * tries that out once (three one-lane kernels on two internal streams, a bounded wait of at most 20 ms, two stream synchronisations)
 * and a context on a device that takes one kernel at a time -- under a counter-collecting profiler, AMD_SERIALIZE_KERNEL -- forks
 * behind events instead.) */
DUET_API int duet_cluster_run_device(duet_ctx *ctx, const duet_cluster_problem *prob, const duet_cluster_result *res,
                            void *stream);
DUET_API int duet_cluster_run_host(duet_ctx *ctx, const duet_cluster_problem *prob, const duet_cluster_result *res);

/* ------------------------------------------------------------------------------------------------------
 * Fused SVIM-mode pipeline: raw SV marks -> A0 clustering -> E/F phasing, everything resident in HBM, no VCF
 * round trip.  What upstream obtains from the SVIM VCF per candidate is derived
 * on the device from the clusters:
 *     POS      = floor mean of the members' pos          SVLEN   = floor mean of the members' span
 *     support  = number of member marks (svread)          GT      = called ("not ./.")
 *     marks    = the members' read indices in cluster order
 *     refread  = max(depth(contig, POS) - support, 0), depth read from a binned coverage array -- this
 *                repository's stand-in for SVIM's AD[0] (reads at the locus that do not support the SV);
 *                like A0 itself it has no pinned reference (svim is external).
 * Candidates come out grouped by contig, then by (type, centre); out_pred / out_ps are indexed like the
 * cluster result's candidate arrays.  The call synchronises `stream` once (candidate counts per contig come
 * back to the host to lay out the E/F workspace).
 * ---------------------------------------------------------------------------------------------------- */
typedef struct duet_svim_problem {
    duet_cluster_problem marks;     /* device arrays */
    const uint32_t *mark_read;      /* [M] device: read index of each raw mark (into read_tag) or DUET_MARK_ABSENT */
    const uint64_t *read_tag;       /* [R] device */
    uint32_t n_reads;
    uint32_t n_contigs;             /* K: mark_contig values are < K */
    const uint32_t *depth;          /* device: coverage bins, contig k at depth_off[k] .. depth_off[k+1] */
    const uint32_t *depth_off;      /* [K+1] HOST */
    uint32_t depth_bin;             /* bin width in bp (>= 1) */
    uint32_t svlen_thres, suppread_thres;
    uint32_t reserved;
} duet_svim_problem;

/* res: device arrays as for duet_cluster_run_device (n_cands a device word); out_pred[M], out_ps[M] device.
 * n_cands_host != NULL: the call waits once for the clustering to learn the candidate count, stores it there and
 * plans E/F on the host.  n_cands_host == NULL: nothing waits -- E/F is planned on the device for the upper bound of M
 * candidates and reads the count from res->n_cands; the caller gets the count from there after synchronising. */
DUET_API int duet_svim_phase_device(duet_ctx *ctx, const duet_svim_problem *prob, const duet_cluster_result *res,
                           uint8_t *out_pred, uint32_t *out_ps, uint32_t *n_cands_host, void *stream);

/* Host convenience (what a rank of `duet -b svim-gpu --gpus N` calls: no device-memory framework in the process): every
 * array of *prob and *res is HOST memory (res->order may be NULL; res->n_cands a host word; the arrays need room for M
 * entries, cand_off M + 1), out_pred[M] / out_ps[M] host; uploads, runs the fused pipeline, downloads, synchronises.
 * Returns DUET_ERR_DIV_ZERO like duet_ef_run_host. */
DUET_API int duet_svim_phase_host(duet_ctx *ctx, const duet_svim_problem *prob, const duet_cluster_result *res,
                         uint8_t *out_pred, uint32_t *out_ps);

/* ---------------------------------------------------------------------------------------------
 * Rows of phased_sv.vcf on the device: from (pred, ps) to the text of the data rows.
 * Replaces, for the rows: the emission order of src/duet/sv_phasing_fn.py:204-228 (contig order, PS-class 0/1/2,
 * file order, pred 0 dropped), the stable sort of :229 (CHROM as text, POS as int), print_sv of
 * src/duet/write_file.py:6-17 and the SVLEN sign rule of sv_phasing_fn.py:225.  The header lines stay with the host.
 *
 * All pointers are device memory except cand_ctg_off.  The texts of candidate c are
 * pool[str_off[4c] .. str_off[4c+1]) = CHROM, then REF, ALT and SVTYPE up to str_off[4c+4].
 * cand_chrom_rank[c] = rank of c's CHROM text among the n_chrom_texts distinct CHROM texts in byte order
 * (shorter text first on a common prefix), which is how Python compares the strings at :229. */
typedef struct duet_rows_problem {
    uint32_t n_contigs, n_cands;
    const uint32_t *cand_ctg_off;        /* HOST [K+1] */
    const uint8_t *pred;                 /* [C] from duet_ef_run_device */
    const uint32_t *ps;                  /* [C] */
    const uint32_t *cand_pos, *cand_svlen;
    const uint8_t *cand_plus;            /* [C] 1: SVTYPE is exactly INS or DUP, SVLEN is written positive */
    const uint16_t *cand_chrom_rank;     /* [C] */
    uint32_t n_chrom_texts;
    uint32_t max_pos;                    /* 0 = unknown (32 key bits for POS) */
    const char *pool;
    uint64_t pool_bytes;
    const uint32_t *str_off;             /* [4C+1] */
    const uint32_t *cand_off, *mark_read;/* the E/F problem's CSR and tag table: the PS-class is recomputed from them */
    const uint64_t *read_tag;
} duet_rows_problem;

/* Writes the rows, in final order, to out_text (device, out_cap bytes; pool_bytes + 96 * n_cands always suffices);
 * *out_len = bytes written, *n_rows = rows.  Synchronises `stream` (twice). */
DUET_API int duet_rows_run_device(duet_ctx *ctx, const duet_rows_problem *prob, char *out_text, uint64_t out_cap, uint64_t *out_len,
                         uint32_t *n_rows, void *stream);

/* Host-array convenience of the two together: *prob as for duet_ef_run_host, *rows with HOST arrays cand_plus,
 * cand_chrom_rank, pool, str_off (+ n_cands, n_chrom_texts, max_pos, pool_bytes; the other fields are filled in here);
 * uploads, runs E/F and the row emission on the device and copies the rows' text to out_text (host, out_cap bytes).
 * Returns what duet_ef_run_host would (e.g. DUET_ERR_DIV_ZERO) before writing any row. */
DUET_API int duet_ef_rows_run_host(duet_ctx *ctx, const duet_ef_problem *prob, const duet_rows_problem *rows, char *out_text,
                          uint64_t out_cap, uint64_t *out_len, uint32_t *n_rows);

/* ---------------------------------------------------------------------------------------------
 * Accuracy evaluator: a phased callset scored against a truth set, the arithmetic of
 * src/scripts/evaluation.py:99-159.  The host parses the two VCFs the way :35-97 does and flattens them:
 *   truth ("base") records grouped by list key = 2 * contig + type (0 INS, 1 DEL), position-sorted (stably) inside a list;
 *   calls that sit on a listed contig and have one of the two types, with their list key, their phase-set group
 *   (= distinct (contig, phase set) pair, :109-111) and their haplotype as a code: 0 '1|0', 1 '0|1', 2 '1|1', >= 3 any other
 *   string (compared for equality only); record ids (:51) as dense integers -- equal id strings share one integer, because
 *   upstream counts SETS of ids.
 * Out: the sizes of the six sets of :100 -- call_tp, base_tp, call_tp_gt, base_tp_gt, call_tp_hp, base_tp_hp.  The ten
 * numbers upstream prints are quotients of these and of len(callinfo) / len(baseinfo), taken on the host. */
typedef struct duet_eval_problem {
    uint32_t n_base, n_calls, n_groups, n_keys;      /* n_keys = 2 * contigs */
    uint32_t n_base_uid, n_call_uid;                 /* distinct ids on either side */
    uint32_t refdist;                                /* -r / --refdist (:126) */
    uint32_t reserved;
    double ratio;                                    /* -p / --pctsim (:127) */
    const uint32_t *base_off;                        /* [n_keys + 1] */
    const uint32_t *base_pos, *base_len, *base_uid;  /* [n_base] */
    const uint8_t *base_hp;                          /* [n_base] */
    const uint32_t *call_key;                        /* [n_calls]; every call's list must be non-empty (upstream raises
                                                        IndexError otherwise, :120-125: the host checks) */
    const uint32_t *call_pos, *call_len, *call_uid, *call_group;
    const uint8_t *call_hp;
} duet_eval_problem;

typedef struct duet_eval_counts {
    uint32_t call_tp, base_tp, call_gt, base_gt, call_hp, base_hp;
} duet_eval_counts;

/* All arrays are HOST pointers; uploads, runs four small kernels, synchronises. */
DUET_API int duet_eval_run_host(duet_ctx *ctx, const duet_eval_problem *prob, duet_eval_counts *counts);

/* ---------------------------------------------------------------------------------------------
 * The collective of the contig-sharded path: candidates shard by contig over the GPUs of one node,
 * one process and one context per GPU, and ONE all-gather of fixed-size record blocks reassembles the call set
 * (src/duet/sv_phasing_fn.py:15-18, 195-210: nothing crosses contigs before the final sort at :229).  RCCL over xGMI,
 * loaded at run time (the library has no link-time dependency on it; these calls fail with DUET_ERR_NO_DEVICE where it is
 * missing).  The caller hands the 128-byte unique id of rank 0 to every rank by its own means (duet_amd/comm.py: a TCP
 * star on MASTER_ADDR:MASTER_PORT); duet_comm_create is collective (ncclCommInitRank) and returns NULL on failure
 * (duet_last_error(ctx)). */
#define DUET_COMM_ID_BYTES 128
typedef struct duet_comm duet_comm;
DUET_API int duet_comm_rccl_version(duet_ctx *ctx);      /* ncclGetVersion's code (e.g. 22707) of the RCCL the library loaded, or a negative status */
DUET_API int duet_comm_unique_id(duet_ctx *ctx, unsigned char *id /* [DUET_COMM_ID_BYTES] */);
DUET_API duet_comm *duet_comm_create(duet_ctx *ctx, const unsigned char *id, int rank, int world);
/* Every blocking step is bounded: duet_comm_create and the synchronising calls below give up after the time limit -- the
 * environment's DUET_RDZV_TIMEOUT in seconds (default 300) at creation, duet_comm_set_timeout afterwards -- with
 * DUET_ERR_TIMEOUT (duet_comm_create: NULL and that text in duet_last_error) instead of waiting for a rank that never comes. */
DUET_API int duet_comm_set_timeout(duet_comm *comm, double seconds);
/* every rank contributes `bytes` bytes, every rank receives world * bytes (rank-major).  _device: device pointers,
 * asynchronous on `stream`; _host: host pointers, staged through device buffers of the communicator, synchronises. */
DUET_API int duet_comm_allgather_device(duet_comm *comm, const void *send, uint64_t bytes, void *recv, void *stream);
DUET_API int duet_comm_allgather_host(duet_comm *comm, const void *send, uint64_t bytes, void *recv);
/* One rank's whole data path of the contig-sharded run, results never leaving the device before the collective
 * (src/duet/sv_phasing_fn.py:189-228 on the rank's contigs, then the reassembly in front of :229):
 *   *prob (HOST arrays, the rank's shard) is uploaded, ef_classify -> ef_seed_sort -> ef_finalize write straight into the
 *   rank's record block on the device,
 *       ps u32[n_max] | pred u8[n_max] | pad to 16 | status u32, 12 bytes pad | kept u64[n_slots]
 *   (n_max = the largest shard's candidate count, the same on every rank; status = 5 when a candidate that reaches the
 *   decision has svread + refread == 0, else 0; kept[s] = candidates c with pred != 0 and cand_slot[c] == s: the rows each
 *   CHROM text contributes, which is what tells every rank where its rows' numbering starts), ONE ncclAllGather of the
 *   blocks on the kernels' stream, one copy of the world * block bytes to `gathered` (host), bounded wait.
 * duet_comm_block_bytes gives the block size.  cand_slot may be NULL when n_slots == 0. */
DUET_API uint64_t duet_comm_block_bytes(uint32_t n_max, uint32_t n_slots);
DUET_API int duet_comm_ef_allgather(duet_comm *comm, const duet_ef_problem *prob, const uint32_t *cand_slot, uint32_t n_slots,
                                    uint32_t n_max, uint8_t *gathered);
/* A rank whose own part of duet_comm_ef_allgather fails (arguments, memory, upload, a launch) still contributes a block to
 * the collective -- zeros, with this bit set in the status word and the negated duet_status in the low 16 bits -- so that its
 * peers fail at once instead of waiting for the time limit; the failing rank returns its own error after the collective. */
#define DUET_COMM_STATUS_RANK_FAILED 0x80000000u
/* What RCCL itself says about the communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice; -1 where the loaded RCCL
 * lacks the call): `rccl_ranks` == world is the evidence that RCCL, not a stand-in, connected every rank. */
DUET_API int duet_comm_info(duet_comm *comm, int *rank, int *world, int *rccl_ranks, int *rccl_rank, int *rccl_device);
/* Every rank all-gathers `words` 32-bit words of a pattern that names its rank and the word's place, and checks every slot
 * of what comes back (collective: every rank must call it with the same `words`).  DUET_OK or the first wrong word. */
DUET_API int duet_comm_selftest(duet_comm *comm, uint32_t words);
/* Frees the communicator.  On one that timed out (DUET_ERR_TIMEOUT) nothing of the device is touched -- a collective may be
 * stuck on the stream, ncclCommDestroy and hipFree would wait for it: its buffers are leaked, the process is expected to exit. */
DUET_API void duet_comm_destroy(duet_comm *comm);

#ifdef __cplusplus
}
#endif
#endif /* DUET_EF_H */
