/*
 * otto_mf.h -- C-ABI of the MI355X matrix-factorization trainer / scorer.
 *
 * Reference interfaces replaced (all under /root/reference/src):
 *   otto_mf_forward            matrix_factorization/torch_modules.py:13-19 (CollaborativeFiltering.forward)
 *                              and :32-38 (MatrixFactorization.forward): out[b] = <E1[i1[b]], E2[i2[b]]>
 *   otto_mf_eval               the per-batch body of validate(), matrix_factorization/torch_trainer.py:126-145
 *                              (forward + MSELoss / BCEWithLogitsLoss, predictions kept on the device)
 *   otto_mf_step_sparse_adam   the per-batch body of train(), torch_trainer.py:59-78, with
 *                              torch.optim.SparseAdam semantics (coalesced duplicate rows; moments touched
 *                              only on rows present in the batch; global step t; config.yaml:19-22)
 *   otto_mf_bpr_step           BPR pairwise training reached in the reference only through recbole
 *                              (recbole/trainer.py:28-40) -- arithmetic build-defined, SURVEY.md App. B.2
 *   otto_mf_score_topk         recbole/inference.py:76-80 full_sort_predict + PAD column -inf + topk(20)
 *
 * Conventions as in otto_covis.h: 0 / negative code + otto_last_error(); caller owns every buffer
 * (PyTorch holds the embedding tables and optimizer state and passes data_ptr()); d_* are device
 * pointers; launches go to the caller's hipStream_t; not thread-safe per context.
 * Embedding tables are row-major float32 [n, d]; d in {4, 8, 16, 32, 64, 128, 256}.
 */
#ifndef OTTO_MF_H
#define OTTO_MF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OTTO_MF_LOSS_MSE 0 /* torch.nn.MSELoss(reduction='mean')            */
#define OTTO_MF_LOSS_BCE 1 /* torch.nn.BCEWithLogitsLoss(reduction='mean')  */

#define OTTO_MF_BPR_HOGWILD 0 /* racing in-place SGD row updates (north star)                      */
#define OTTO_MF_BPR_BATCH 1   /* gradients from the pre-step tables, duplicates summed, then applied */

typedef struct otto_mf_ctx otto_mf_ctx;

/* n1/n2 rows of table 1/2 (shared_table != 0: one table used for both index columns, n2 ignored --
 * CollaborativeFiltering), factors d, largest batch any step/eval call will pass. */
int otto_mf_create(otto_mf_ctx** ctx, int64_t n1, int64_t n2, int32_t d, int64_t max_batch, int32_t shared_table);
void otto_mf_destroy(otto_mf_ctx* ctx);

int otto_mf_forward(otto_mf_ctx* ctx, const float* d_E1, const float* d_E2, const int64_t* d_i1, const int64_t* d_i2,
                    int64_t B, float* d_pred, void* stream);

/* d_pred may be NULL. *d_loss_out (device float) = mean loss of the batch. */
int otto_mf_eval(otto_mf_ctx* ctx, const float* d_E1, const float* d_E2, const int64_t* d_i1, const int64_t* d_i2,
                 const int64_t* d_target, int64_t B, int32_t loss_kind, float* d_pred, float* d_loss_out, void* stream);

/* validate() with scores (torch_trainer.py:126-161 + matrix_factorization/metrics.py:30-85) without copying the
 * predictions to the host: as otto_mf_eval, and the context's running sums grow by this batch's
 *   sum |p - t|, sum (p - t)^2, #{(p >= 0.5) == (t >= 0.5)}, B      (double precision, added in launch order)
 * with p = the raw output for OTTO_MF_LOSS_MSE models and sigmoid(output) for OTTO_MF_LOSS_BCE models: mean absolute error,
 * mean squared error and accuracy at the reference's threshold 0.5 are quotients of these sums. */
int otto_mf_eval_sums(otto_mf_ctx* ctx, const float* d_E1, const float* d_E2, const int64_t* d_i1, const int64_t* d_i2,
                      const int64_t* d_target, int64_t B, int32_t loss_kind, float* d_pred, float* d_loss_out, void* stream);
/* h_sums[4] (host) = the running sums above; reset != 0 clears them afterwards. Synchronises the stream. */
int otto_mf_read_sums(otto_mf_ctx* ctx, double* h_sums, int32_t reset, void* stream);

/* Row ids are range-checked in every kernel: a sample whose id falls outside its table is skipped (its prediction is NaN)
 * and counted. otto_mf_check synchronises the stream and returns 0 if no sample was skipped since the last call, else -22
 * (EINVAL) with the count in *n_bad (nullable) and the message in otto_last_error(); the counter is cleared. The reference
 * would raise an IndexError from nn.Embedding (torch_modules.py:15-16, 34-35) at the offending batch. */
int otto_mf_check(otto_mf_ctx* ctx, int64_t* n_bad, void* stream);

/* One optimizer step. (d_m*, d_v*) = SparseAdam exp_avg / exp_avg_sq, same shape as the table.
 * t = 1-based global step count AFTER this step (torch's state['step']). shared_table contexts
 * pass the same pointers for table 2.  *d_loss_out = mean loss of the batch before the update. */
int otto_mf_step_sparse_adam(otto_mf_ctx* ctx, float* d_E1, float* d_m1, float* d_v1, float* d_E2, float* d_m2,
                             float* d_v2, const int64_t* d_i1, const int64_t* d_i2, const int64_t* d_target, int64_t B,
                             int32_t loss_kind, double lr, double beta1, double beta2, double eps, int64_t t,
                             float* d_loss_out, void* stream);

/* BPR-SGD over B (user, positive item) rows: negative j ~ U{0..n2-1} from a counter-based RNG keyed by
 * (seed, epoch, row0 + b), redrawn while j == i; x = <U_u, V_i - V_j>; loss = softplus(-x);
 * U_u += lr*(s*(V_i-V_j) - l2*U_u), V_i += lr*(s*U_u - l2*V_i), V_j += lr*(-s*U_u - l2*V_j), s = sigmoid(-x).
 * *d_loss_sum (device float) receives the SUM of the B losses; d_neg_out (nullable) the sampled j. */
int otto_mf_bpr_step(otto_mf_ctx* ctx, float* d_U, float* d_V, const int64_t* d_u, const int64_t* d_i, int64_t B,
                     uint64_t seed, uint64_t epoch, int64_t row0, float lr, float l2, int32_t mode, float* d_loss_sum,
                     int64_t* d_neg_out, void* stream);

/* scores[b, n] = <U[b], V[n]> over all N items, column `pad_col` (if >= 0) forced to -inf, per row the
 * k best (score desc, id asc): d_ids [B, k] int32, d_scores [B, k] float32.  Exact f32 MFMA, the B x N
 * matrix is never written.  k <= 32. */
int otto_mf_score_topk(const float* d_U, const float* d_V, int64_t B, int64_t N, int32_t d, int32_t k,
                       int64_t pad_col, int32_t* d_ids, float* d_scores, void* d_workspace, int64_t workspace_bytes,
                       void* stream);
/* Exact merge of n_lists partial top-k lists per row (item-sharded scoring across GPUs, SURVEY.md section 8 e): d_part_scores /
 * d_part_ids are [n_lists][B][k] (id -1 or 0x7FFFFFFF = empty slot); output as otto_mf_score_topk, (score desc, id asc). */
int otto_mf_topk_merge(const float* d_part_scores, const int32_t* d_part_ids, int32_t n_lists, int64_t B, int32_t k,
                       int32_t* d_ids, float* d_scores, void* stream);
/* bytes of workspace otto_mf_score_topk needs for (B, k) */
int64_t otto_mf_score_workspace(int64_t B, int64_t N, int32_t k);

#ifdef __cplusplus
}
#endif
#endif
