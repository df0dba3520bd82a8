"""BPR matrix factorization: triplet sampling, embedding gather, fp32 dot, SGD scatter
(``otto_mf_bpr_step``) and full-sort top-k scoring (``otto_mf_score_topk``).

The reference reaches BPR only through recbole (``src/recbole/trainer.py:28-40``) and scores with
``model.full_sort_predict`` + ``scores[:, 0] = -inf`` + ``torch.topk(scores, 20)``
(``src/recbole/inference.py:58-61, 76-80``); parameter names follow recbole's BPR
(``user_embedding.weight``, ``item_embedding.weight``) so checkpoints keep their keys.
Arithmetic: SURVEY.md App. B.2 (build-defined, parity unpinned by the reference).
"""
import torch
import torch.nn as nn

from .engine import MFEngine, score_topk, BPR_HOGWILD, BPR_BATCH  # noqa: F401


class BPR(nn.Module):

    def __init__(self, n_users, n_items, embedding_size=64):
        super().__init__()
        self.user_embedding = nn.Embedding(n_users, embedding_size)
        self.item_embedding = nn.Embedding(n_items, embedding_size)
        nn.init.xavier_normal_(self.user_embedding.weight)
        nn.init.xavier_normal_(self.item_embedding.weight)
        self._engine = None

    def engine(self, batch):
        U = self.user_embedding.weight
        if self._engine is None or self._engine.max_batch < batch or self._engine.device != U.device:
            if self._engine is not None:
                self._engine.close()
            self._engine = MFEngine(U.shape[0], self.item_embedding.weight.shape[0], U.shape[1], int(batch), device=U.device)
        return self._engine

    def full_sort_topk(self, users, k=20, pad_col=0):
        """Top-k item ids/scores of ``users`` over ALL items; item ``pad_col`` (recbole's PAD, id 0)
        is excluded (``src/recbole/inference.py:79``)."""
        U = self.user_embedding.weight.data[users].contiguous()
        return score_topk(U, self.item_embedding.weight.data, k=k, pad_col=pad_col)


def train_epoch(model, users, items, lr, l2=0.0, seed=42, epoch=0, mode=BPR_HOGWILD, rows_per_launch=1 << 24, row0=0):
    """One pass of BPR-SGD over the (user, positive item) rows (device int64 tensors).
    Returns the mean loss (read once at the end)."""
    n = users.numel()
    n_launch = (n + rows_per_launch - 1) // rows_per_launch
    eng = model.engine(min(n, rows_per_launch))
    losses = torch.zeros(n_launch, dtype=torch.float32, device=users.device)
    U, V = model.user_embedding.weight.data, model.item_embedding.weight.data
    for q in range(n_launch):
        lo, hi = q * rows_per_launch, min(n, (q + 1) * rows_per_launch)
        eng.bpr_step(U, V, users[lo:hi], items[lo:hi], seed, epoch, row0 + lo, lr, l2, mode, loss_sum=losses[q:q + 1])
    return float(losses.sum().item()) / max(n, 1)


def sync_item_table(V, V_snapshot, group=None):
    """Data-parallel BPR (sessions sharded by chunk: user rows are rank-private, the item table is
    replicated): every rank trains locally from the same snapshot, then the item-table DELTAS are
    summed with one bucketed all-reduce (RCCL over xGMI) and applied to the snapshot, so every replica
    ends the round with identical tables.  Device-agnostic (gloo in the CPU tests)."""
    import torch.distributed as dist
    delta = V - V_snapshot
    flat = delta.view(-1)
    bucket = 64 << 20     # 256 MB of fp32 per collective: few, large messages for the per-link-bound ring
    for lo in range(0, flat.numel(), bucket):
        dist.all_reduce(flat[lo:lo + bucket], op=dist.ReduceOp.SUM, group=group)
    V_snapshot.add_(delta)
    V.copy_(V_snapshot)
