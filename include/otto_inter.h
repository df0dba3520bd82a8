/*
 * otto_inter.h -- C-ABI of the (session, candidate) interaction features.
 *
 * Replaces the polars group-by / join chain of the reference's src/ranker/interaction_feature_engineering.py:56-113 over the
 * candidate rows of src/ranker/*candidate_generation.py (columns session, candidates, candidate_scores):
 *   row features (:56-85)      per (session, candidate): occurrences of the candidate in the session (all / click / cart /
 *                              order events) and the 1-based position of its last occurrence (null when absent);
 *   session features (:87-100) per session over its candidate rows: score mean / std / min / max, occurrence count
 *                              mean / sum / max, last position mean / sum / max (nulls skipped);
 *   aid features (:102-113)    per candidate aid over all sessions: score mean / std / max, occurrence count mean / sum /
 *                              max, last position mean / sum / max.
 * Candidates come as the dense [n_sess][C] arrays otto_cand_lookup writes (-1 padded, one row per session, candidates
 * unique inside a row: the reference de-duplicates its rows at :33), events as the sorted SoA + CSR of the same sessions.
 * std is the sample standard deviation (ddof 1; NaN for a single row); NaN stands for polars' null.
 *
 * Conventions as in otto_covis.h.
 */
#ifndef OTTO_INTER_H
#define OTTO_INTER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OTTO_INTER_MAX_CAND 128    /* candidates per session (the generators keep 100)   */
#define OTTO_INTER_MAX_SESSION 512 /* events per session (OTTO: 500)                     */
#define OTTO_INTER_ROW_FEATURES 5  /* occurrence, last position (0 = null), click / cart / order occurrences: uint16 */
#define OTTO_INTER_SESSION_FEATURES 10
#define OTTO_INTER_AID_FEATURES 9

/* bytes of scratch (per-aid accumulators) */
int64_t otto_inter_workspace(uint32_t n_aids);

/* d_row u16 [n_sess][C][5]; d_sess_feat f32 [n_sess][10] (order of the reference's agg list, :87-98); d_aid_feat f32
 * [n_aids][9] (order of :102-112; rows of aids that are nobody's candidate are NaN). */
int otto_inter_features(const uint32_t* d_aid, const uint8_t* d_type, const int64_t* d_sess_off, int64_t n_sess,
                        const int32_t* d_cand, const float* d_score, int32_t C, uint32_t n_aids, uint16_t* d_row,
                        float* d_sess_feat, float* d_aid_feat, void* d_workspace, int64_t workspace_bytes, void* stream);

/* The same features over the CSR rows of the ranker's candidate table (otto_cand_ranker_table, include/otto_cand.h): session s
 * owns rows [d_cand_off[s], d_cand_off[s + 1]) of d_cand / d_score -- the session's own aids followed by the candidates, the
 * frame src/ranker/interaction_feature_engineering.py:25-28 reads. d_row u16 [n_rows][5]. */
int otto_inter_features_rows(const uint32_t* d_aid, const uint8_t* d_type, const int64_t* d_sess_off, int64_t n_sess,
                             const int64_t* d_cand_off, const int32_t* d_cand, const float* d_score, uint32_t n_aids,
                             uint16_t* d_row, float* d_sess_feat, float* d_aid_feat, void* d_workspace, int64_t workspace_bytes,
                             void* stream);

#ifdef __cplusplus
}
#endif
#endif
