/*
 * otto_covis.h -- C-ABI of the MI355X covisitation-matrix builder.
 *
 * What this replaces in the reference: NOTHING the reference ships as code -- the
 * builder that writes DATA/covisitation/<mode>/top_15_<kind>_<i>.pqt is absent
 * (SURVEY.md F1).  The entry points below are what a maintainer's ctypes stub in
 * a new src/covisitation/<builder>.py binds; their output feeds,
 * unchanged, the consumers
 *     src/covisitation/inference.py:19-35,87-111,282-308   (covisitation_df_to_dict)
 *     src/ranker/regular_candidate_generation.py:75-101,270-296
 *     src/ranker/covisitation_candidate_generation.py:49-73,201-227
 * The only in-reference relative of the arithmetic is the session self-join of
 *     src/matrix_factorization/torch_trainer.py:198-223.
 * Semantics: SPEC-COVIS in DESIGN.md (restating SURVEY.md App. A).
 *
 * Conventions (SURVEY.md section 8 b): every function returns 0 or a negative
 * OTTO_E* code and sets otto_last_error(); the caller owns every buffer it passes
 * in; all pointers named d_* are DEVICE pointers (hipMalloc / torch CUDA tensors);
 * all work is enqueued on the caller's hipStream_t (passed as void*); one context
 * per device, not thread-safe; no exceptions or Python objects cross the ABI.
 */
#ifndef OTTO_COVIS_H
#define OTTO_COVIS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OTTO_OK 0
#define OTTO_EINVAL (-22)
#define OTTO_ENOMEM (-12)
#define OTTO_EHIP (-5)
#define OTTO_ESTATE (-1)

#define OTTO_COVIS_MAX_WINDOW 32
#define OTTO_COVIS_MAX_FILTERS 4
#define OTTO_COVIS_MAX_TYPE_WEIGHTS 4
#define OTTO_COVIS_MAX_AIDS (1u << 26)

/* kind groups: one pass over the expanded pairs reduces every kind of a group */
#define OTTO_COVIS_GROUP_TYPE 0   /* type-weighted kinds: W = 65536 * sum Wk[type_y]        */
#define OTTO_COVIS_GROUP_FILTER 1 /* pair-filter kinds:   W = 65536 * #sessions with a pair */
#define OTTO_COVIS_GROUP_TIME 2   /* time_weighted:       W = sum 65536 + 3*65536*(ts_x-t0)/(t1-t0) */

typedef struct otto_covis_params {
    int32_t window;        /* tail window W (SPEC-COVIS 2), 2..32                       */
    int32_t max_gap;       /* |ts_x - ts_y| <= max_gap seconds (SPEC-COVIS 3)           */
    uint32_t n_aids;       /* aids are 0..n_aids-1, n_aids <= 2^26                      */
    int32_t ts_min;        /* t0 of the time weight (global over ALL chunks / ranks)    */
    int32_t ts_max;        /* t1                                                        */
    int32_t want_time;     /* !=0: keep the time-weight channel (needed for GROUP_TIME) */
    int32_t n_filters;     /* 0..4 pair-filter kinds                                    */
    uint16_t filter_mask[OTTO_COVIS_MAX_FILTERS];              /* bit (type_x*3+type_y) */
    int32_t n_type_weights;                                    /* 0..4 type-weighted kinds */
    int32_t type_weight[OTTO_COVIS_MAX_TYPE_WEIGHTS][3];       /* Wk[type_y] in 1..255  */
} otto_covis_params;

typedef struct otto_covis_ctx otto_covis_ctx;

/* statistics filled by otto_covis_stats (all int64) */
enum {
    OTTO_COVIS_STAT_SESSIONS = 0,   /* sessions fed                                      */
    OTTO_COVIS_STAT_TAIL_EVENTS,    /* events inside tail windows (= run slots)          */
    OTTO_COVIS_STAT_PAIR_SLOTS,     /* reserved record slots  sum n(n-1)                 */
    OTTO_COVIS_STAT_PAIRS,          /* P: deduped ordered pairs expanded (valid after index) */
    OTTO_COVIS_STAT_RUNS,           /* non-empty (window, aid_x) runs                    */
    OTTO_COVIS_STAT_ITEMS_S,
    OTTO_COVIS_STAT_ITEMS_M,
    OTTO_COVIS_STAT_ITEMS_L,
    OTTO_COVIS_STAT_RETRIES,        /* overflow re-partition rounds since the index was built */
    OTTO_COVIS_STAT_PAIRS_S,        /* expanded pairs reduced by the S / M / L size bins  */
    OTTO_COVIS_STAT_PAIRS_M,
    OTTO_COVIS_STAT_PAIRS_L,
    OTTO_COVIS_STAT_RUNS_S,         /* runs gathered by the S / M / L size bins            */
    OTTO_COVIS_STAT_RUNS_M,
    OTTO_COVIS_STAT_RUNS_L,
    OTTO_COVIS_STAT_SHARED_RUNS,    /* runs that read a shared component list (= list words the pair-expand wrote for them) */
    OTTO_COVIS_STAT_ROW_RECORDS,    /* records the pair-expand wrote into private rows                                       */
    OTTO_COVIS_STAT_COUNT
};

const char* otto_last_error(void);

int otto_covis_create(otto_covis_ctx** ctx, const otto_covis_params* params);
void otto_covis_destroy(otto_covis_ctx* ctx);

/* Forget every fed chunk (keeps the workspace allocations). */
int otto_covis_reset(otto_covis_ctx* ctx);

/*
 * K1 pair-expand over one chunk of sessions.  Events of session s are
 * d_aid/d_ts/d_type[d_sess_off[s] .. d_sess_off[s+1]) with ts NON-DECREASING inside a session (SPEC-COVIS 1;
 * the gap-free window test relies on it).  May be called
 * several times (session chunks); records accumulate in the context.
 * Synchronises the stream once (to size the record buffers).
 */
int otto_covis_feed(otto_covis_ctx* ctx, const uint32_t* d_aid, const int32_t* d_ts, const uint8_t* d_type,
                    const int64_t* d_sess_off, int64_t n_sess, void* stream);

/*
 * Group the expanded pairs by aid_x, reduce and select: for every kind j of
 * `group` and every aid_x the top-k aid_y by (W desc, aid_y asc).
 *   d_out_y [n_kinds][n_aids][k]  uint32   aid_y, rank order
 *   d_out_w [n_kinds][n_aids][k]  uint64   Q16 weight W (wgt = W / 65536)
 *   d_out_n [n_kinds][n_aids]     int32    valid entries (0..k)
 * n_kinds = n_type_weights / n_filters / 1 for GROUP_TYPE / FILTER / TIME.
 * Synchronises the stream (overflow check; exactness is unconditional).
 */
int otto_covis_finalize(otto_covis_ctx* ctx, int group, int k, uint32_t* d_out_y, uint64_t* d_out_w,
                        int32_t* d_out_n, void* stream);

int otto_covis_stats(otto_covis_ctx* ctx, int64_t* out /* [OTTO_COVIS_STAT_COUNT] */);

/* Tuning knobs and A/B switches; results are exact for every value (each is exercised by tests/test_covis_gpu.py).
 *   "l_cap": expanded pairs per hash partition of a heavy aid_x in the wide table layout (default 6144; the packed
 *            2^14-slot layout takes twice as many);
 *   "partition": 1 (default) bucket heavy aids' pairs by partition once, 0 re-read and filter per partition;
 *   "fused": pair-expand kernel when no filter kind is configured: 2 (default) component lists (k_expand_lists), 1 one record per
 *            pair from registers (k_expand_fused), 0 class-sorted kernels;
 *   "fast_path": 1 (default) gap-free window shortcut in the pair-expand kernels;
 *   "bucket_index": 1 (default) group the runs by aid_x with LDS atomics per 1024-aid bucket, 0 one global atomic per run;
 *   "packed_heavy": packed 12-bit-counter tables for heavy aids with fewer than 4096 runs: 2 (default) 2^14 slots for aids that
 *                   fit one table, 2^13-slot partitions beyond; 1: 2^14-slot partitions of twice the size; 0: wide tables only;
 *   "guess": 1 (default) single-pass top-k of a heavy aid's partitions from a sibling partition's threshold;
 *   "hot": 2 (default) top-k walks over the heavy keys only + single-wave selection where they are few, 1 walks only, 0 off;
 *   "bkt_sh": log2 of the aids per index bucket (default clamp(aid_bits - 10, 10, 13)); "part_sized": 1 (default) partition buckets
 *            sized from the record counts without a count pass; "overlap_partition": 1 partition pass on a side stream beside the
 *            S / M bins (default 0); "s_wgs" / "p_wgs": workgroups per CU of the one-wave reduce bin (20) / the partition scatter (4; its chunks are dequeued dynamically);
 *   "debug_skip": timing diagnostics only (results invalid): 1 no gather, 2 no top-k, 4 no table clear, 8 no inserts,
 *                 16 / 32 pair-expand without record stores / row loops. */
int otto_covis_set_option(otto_covis_ctx* ctx, const char* name, int64_t value);

/*
 * Multi-GPU exchange (SURVEY.md section 8 e): the expanded runs whose aid_x lies in
 * [x_lo, x_hi) leave as  d_hdr[2*n_runs] = {aid_x, len}*,  d_rec[n_recs] (records of the
 * runs back to back) and, when want_time, d_tw[n_recs]; the aid_x owner appends them with
 * otto_covis_import_runs.  export_count sizes the buffers (synchronises the stream).
 */
int otto_covis_export_count(otto_covis_ctx* ctx, uint32_t x_lo, uint32_t x_hi, int64_t* n_runs, int64_t* n_recs,
                            void* stream);
int otto_covis_export_runs(otto_covis_ctx* ctx, uint32_t x_lo, uint32_t x_hi, uint32_t* d_hdr, uint32_t* d_rec,
                           uint32_t* d_tw, void* stream);
int otto_covis_import_runs(otto_covis_ctx* ctx, const uint32_t* d_hdr, int64_t n_runs, const uint32_t* d_rec,
                           const uint32_t* d_tw, int64_t n_recs, void* stream);
/* Zero-copy receive: room for n_recs records (and time extras) at the end of the context's own record arrays.
 * Let the all-to-all-v write there and pass exactly these pointers to otto_covis_import_runs: it then only
 * registers the runs (no device copy). *d_tw is NULL without the time channel. The pointers stay valid until the
 * next feed / import_reserve / reset. */
int otto_covis_import_reserve(otto_covis_ctx* ctx, int64_t n_recs, uint32_t** d_rec, uint32_t** d_tw, void* stream);

/* The same exchange in two passes over the runs for ALL owners at once: owner o holds aid_x in
 * [h_bounds[o], h_bounds[o+1]) (host array of n_owners+1 cut points, first/last treated as 0 / +inf).
 * plan: per-owner run and record counts (synchronises the stream); fill: every owner's piece, owner-major, in ONE
 * hdr / rec (/ tw) buffer -- directly the send buffer of an all-to-all-v with the planned counts as split sizes. */
int otto_covis_export_plan(otto_covis_ctx* ctx, int n_owners, const uint32_t* h_bounds, int64_t* h_n_runs,
                           int64_t* h_n_recs, void* stream);
int otto_covis_export_fill(otto_covis_ctx* ctx, int n_owners, const uint32_t* h_bounds, uint32_t* d_hdr, uint32_t* d_rec,
                           uint32_t* d_tw, void* stream);
/* The same two passes over the run slots [slot_lo, slot_hi) only (run slots: 0 .. OTTO_COVIS_STAT_TAIL_EVENTS): the exchange
 * is cut into a few slot ranges so that the fill of range c + 1 runs while range c is on the links. The planned counts
 * travel through the caller (plan_range's outputs are fill_range's inputs), not through the context. */
int otto_covis_export_plan_range(otto_covis_ctx* ctx, int n_owners, const uint32_t* h_bounds, int64_t slot_lo, int64_t slot_hi,
                                 int64_t* h_n_runs, int64_t* h_n_recs, void* stream);
int otto_covis_export_fill_range(otto_covis_ctx* ctx, int n_owners, const uint32_t* h_bounds, int64_t slot_lo, int64_t slot_hi,
                                 const int64_t* h_n_runs, const int64_t* h_n_recs, uint32_t* d_hdr, uint32_t* d_rec, uint32_t* d_tw,
                                 void* stream);

/* Test hook: copy the raw K1 output to HOST buffers (any pointer may be NULL).
 * h_rec/h_tw: [PAIR_SLOTS] uint32, h_run_x: [TAIL_EVENTS] uint32, h_run_desc: [TAIL_EVENTS] uint64
 * (desc = slot_offset << 8 | len). rec = aid_y | type_y << 26 | filter_bits << 28. */
int otto_covis_copy_records(otto_covis_ctx* ctx, uint32_t* h_rec, uint32_t* h_tw, uint32_t* h_run_x,
                            uint64_t* h_run_desc);

/* per-kernel device time of the last feed/finalize in milliseconds (hipEvents on the
 * caller's stream), order: see OTTO_COVIS_T_* ; returns count written. */
enum {
    OTTO_COVIS_T_WINSCAN = 0,
    OTTO_COVIS_T_EXPAND,     /* K1 pair-expand            */
    OTTO_COVIS_T_INDEX,      /* histogram + scan + scatter + item lists */
    OTTO_COVIS_T_PARTITION,  /* heavy aids: count + scan + scatter of records into hash-partition buckets */
    OTTO_COVIS_T_REDUCE_S,
    OTTO_COVIS_T_REDUCE_M,
    OTTO_COVIS_T_REDUCE_L,
    OTTO_COVIS_T_MERGE,
    OTTO_COVIS_T_COUNT
};
int otto_covis_timings(otto_covis_ctx* ctx, float* out_ms /* [OTTO_COVIS_T_COUNT] */);
/* Names of the kernels (template instantiations) launched under timing slot `slot` since the last reset, joined by " + "
 * (what a rocprofv3 kernel trace of the same run lists): bench.py labels its roofline objects with these. */
int otto_covis_kernel_names(otto_covis_ctx* ctx, int32_t slot, char* buf, int32_t n);

/* PMC calibration helper: streams n_u32 * 4 bytes with one 4-byte access per lane (write != 0: stores, else loads),
 * the access shape of the covisitation kernels, so FETCH_SIZE / WRITE_SIZE can be scaled to bytes (tools/pmc_traffic.py). */
int otto_debug_calibrate(uint32_t* d_buf, int64_t n_u32, int32_t write, void* stream);

#ifdef __cplusplus
}
#endif
#endif
