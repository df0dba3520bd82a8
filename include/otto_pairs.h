/*
 * otto_pairs.h -- C-ABI of the aid-pair dataset builders of the collaborative-filtering trainer.
 *
 * Reference code replaced (/root/reference/src/matrix_factorization/torch_trainer.py):
 *   otto_pairs_time   :190-227  'time' strategy: per session-chunk row sample -> session self-join (merge on session) -> drop
 *                               aid_x == aid_y -> target = 0 < ts_y - ts_x <= hour_difference hours -> groupby (aid_x, aid_y)
 *                               mean >= 0.5 or max. The only pair expansion the reference contains. Sampling is the
 *                               caller's: pass the sampled events.
 *                               DEPARTURE from the reference: the label rule here is 0 < dt <= max_dt_seconds with dt the
 *                               SIGNED difference in total seconds. The reference reads `(ts_y - ts_x).dt.seconds` (:206), the
 *                               seconds COMPONENT of the timedelta (0..86399): it drops whole days (25 h later reads as 1 h
 *                               later: label 1) and turns -10 min into 23 h 50 min (SURVEY.md App. E lists it as a defect).
 *                               The two rules differ only for |dt| >= 1 day; oracle/pairs_oracle.py and its hand-computed
 *                               fixture follow the rule stated here.
 *   otto_pairs_diff   :229-255  'diff' strategy: per session x1 = aid, x2 = next aid, x3 = the aid at the same position of a
 *                               random permutation of the session; positives (x1, x2) with x2 != x3, x1 != x2, x1 != x3,
 *                               negatives (x1, x3) with x2 != x3, x1 != x3; de-duplicated, positives win.
 * Both emit (aid pair, label) records, sort them by pair with the radix sort of otto_events.h and aggregate runs of
 * equal pairs. Output rows are sorted by (x1, x2).
 *
 * Conventions as in otto_covis.h. Events are the sorted SoA + CSR of otto_events_sort.
 */
#ifndef OTTO_PAIRS_H
#define OTTO_PAIRS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OTTO_PAIRS_AGG_MEAN 0 /* target = (mean of the labels >= 0.5) */
#define OTTO_PAIRS_AGG_MAX 1  /* target = max of the labels            */

/* upper bound of the records either builder emits (host value; synchronises the stream): 'time': sum n (n - 1), 'diff': 2 E */
int otto_pairs_raw_count(const int64_t* d_sess_off, int64_t n_sess, int32_t strategy_time, int64_t* h_raw, void* stream);
/* bytes of workspace for `raw` records */
int64_t otto_pairs_workspace(int64_t raw);

/* d_out_* int64 [capacity >= raw]; *h_n_rows = distinct pairs written. */
int otto_pairs_time(const uint32_t* d_aid, const int32_t* d_ts, const int64_t* d_sess_off, int64_t n_sess, int64_t raw,
                    int64_t max_dt_seconds, int32_t aggregation, int64_t* d_out_x1, int64_t* d_out_x2, int64_t* d_out_target,
                    int64_t* h_n_rows, void* d_workspace, int64_t workspace_bytes, void* stream);

/* d_shuffled_aid u32[E]: the aids of every session in a random order of the session's events (the caller permutes with
 * otto_events_sort on (session, random key): polars' shuffle() is unseeded in the reference). */
int otto_pairs_diff(const uint32_t* d_aid, const uint32_t* d_shuffled_aid, const int64_t* d_sess_off, int64_t n_sess, int64_t raw,
                    int64_t* d_out_x1, int64_t* d_out_x2, int64_t* d_out_target, int64_t* h_n_rows, void* d_workspace,
                    int64_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
