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


def train_epoch(model, users, items, lr, l2=0.0, seed=42, epoch=0, mode=BPR_HOGWILD, rows_per_launch=1 << 24, row0=0,
                sync=None, sync_every=4):
    """One pass of BPR-SGD over the (user, positive item) rows (device int64 tensors).
    Returns the mean loss (read once at the end).

    Data-parallel (one process per GPU, rows sharded by session chunk): pass ``sync`` = an :class:`ItemTableSync` over
    ``model.item_embedding.weight.data``; the item rows the launches touch are exchanged every ``sync_every`` launches
    and the pipeline is drained at the end of the epoch (replicas bit-identical). ``row0`` must be this rank's offset
    into the global row numbering so that the counter RNG draws different negatives on every rank.

    The exchange schedule is GLOBAL: shards of a session-chunk split hold different numbers of rows, so the number of
    launch slots of the epoch is the maximum over the ranks (one all-reduce per epoch) and a rank that has run out of
    rows keeps taking part in the exchanges with empty slots -- every rank issues the same collectives in the same order."""
    n = users.numel()
    n_launch = (n + rows_per_launch - 1) // rows_per_launch
    n_slots = sync.global_max(n_launch) if sync is not None else n_launch
    if sync is not None:
        # small launches report the rows they wrote (sparse exchange possible); decided from arguments every rank shares
        sync.tracking = rows_per_launch <= sync.mark.numel() // 4
    eng = model.engine(max(1, min(n, rows_per_launch)))
    losses = torch.zeros(max(n_launch, 1), dtype=torch.float32, device=users.device)
    U, V = model.user_embedding.weight.data, model.item_embedding.weight.data
    neg = torch.empty(max(1, min(n, rows_per_launch)), dtype=torch.int64, device=users.device) if sync is not None else None
    for q in range(n_slots):
        lo, hi = min(n, q * rows_per_launch), min(n, (q + 1) * rows_per_launch)
        if hi > lo:
            eng.bpr_step(U, V, users[lo:hi], items[lo:hi], seed, epoch, row0 + lo, lr, l2, mode, loss_sum=losses[q:q + 1],
                         neg_out=neg)
        if sync is not None:
            if hi > lo and sync.tracking:
                sync.touched(items[lo:hi], neg[:hi - lo])
            if (q + 1) % sync_every == 0:
                sync.exchange()
    if sync is not None:
        sync.finish()
    return float(losses.sum().item()) / max(n, 1)


class ItemTableSync:
    """Data-parallel BPR over session-chunk shards (SURVEY.md section 8 e): user rows are rank-private (no communication
    for the big table), the item table is replicated. Every rank keeps

    * ``V``      its working replica (the kernels update it in place),
    * ``base``   the COMMON table: the initial table plus every delta already exchanged -- bit-identical on all ranks,
    * two persistent exchange buffers (no allocation per call).

    ``exchange()`` (every few launches): this rank's delta since the last exchange, ``V - base``, is combined over the
    ranks and added to ``base``; ``V`` receives the OTHER ranks' share (its own is already in it). Two transports:

    * dense: one all-reduce of the delta table, started asynchronously on the communication stream and folded in at the
      NEXT exchange, so the collective runs under the following kernel launches (stale-synchronous: foreign updates
      arrive one period late). This is what the full-OTTO shape needs -- 16.7 M triplets per launch touch nearly every
      one of the 1.86 M item rows, so a sparse list would be the whole table plus ids;
    * sparse: in ``tracking`` mode (the caller reports the rows of every launch with ``touched(ids)``; set the same on
      every rank, e.g. from the launch size) and when the touched rows are few (small batches, tests, fine-tuning): an
      all-gather of (row id, delta row) pairs -- the exchange section 8 (e) describes -- applied in rank order on every
      rank. One launch declared ``untracked()`` on any rank makes the period dense: a sparse exchange that knew only
      some of the written rows would drop the others' updates.

    ``reduce``: ``'sum'`` (default) adds the ranks' deltas -- every triplet's SGD step is applied once, exactly as the
    single-process loop applies it, only later (one launch is already thousands of concurrent steps from one table; W
    ranks make that W launches). ``'mean'`` divides the combined delta by W (model averaging): a conservative choice
    when a period is so long that every rank alone drives the hot rows to their optimum and the sum would overshoot; it
    converges about W times slower per epoch (CPU rehearsal at W = 2 / 4 / 8 in tests/test_distributed_gloo.py: 'sum'
    ends within a few per cent of the 1-rank loss, 'mean' 9 - 35 % above it).

    ``finish()`` drains the pipeline with a blocking DENSE exchange and copies ``base`` into ``V``: replicas are
    bit-identical. Works on any torch.distributed backend (RCCL on the GPUs, gloo in the CPU tests)."""

    def __init__(self, V, group=None, sparse_fraction=0.125, reduce='sum'):
        import torch.distributed as dist
        if reduce not in ('sum', 'mean'):
            raise ValueError("reduce must be 'sum' or 'mean'")
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group)
        self.scale = 1.0 if reduce == 'sum' else 1.0 / self.world
        self.V = V
        self.base = V.clone()
        self.own = torch.empty_like(V)
        self.red = torch.empty_like(V)
        self.pending = None
        self.sparse_fraction = sparse_fraction
        self.mark = torch.zeros(V.shape[0], dtype=torch.bool, device=V.device)
        self.tracking = False            # the caller reports the rows of EVERY launch (same value on every rank)
        self.all_tracked = True          # tracking: no launch since the last exchange was declared untracked()
        self.stats = {'dense': 0, 'sparse': 0}

    def global_max(self, value):
        """max of an int over the ranks (one small all-reduce; the epoch's launch-slot count)."""
        t = torch.tensor([int(value)], dtype=torch.int64, device=self.V.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return int(t.item())

    def touched(self, *id_tensors):
        """Rows the last launch wrote (positives and, from ``neg_out``, negatives). Ignored unless ``tracking``."""
        if not self.tracking:
            return
        for ids in id_tensors:
            self.mark[ids] = True

    def untracked(self):
        """A launch whose rows were not reported: the current period is exchanged densely."""
        self.all_tracked = False

    def _fold(self):
        """Fold a finished asynchronous all-reduce in: base += s * sum, V += s * sum - own."""
        if self.pending is None:
            return
        self.pending.wait()
        self.pending = None
        if self.scale != 1.0:
            self.red.mul_(self.scale)
        self.base.add_(self.red)
        torch.sub(self.red, self.own, out=self.red)
        self.V.add_(self.red)

    def _sparse(self):
        dist, W = self.dist, self.world
        ids = torch.nonzero(self.mark).reshape(-1)
        n = torch.tensor([ids.numel()], dtype=torch.int64, device=ids.device)
        counts = [torch.zeros_like(n) for _ in range(W)]
        dist.all_gather(counts, n, group=self.group)
        counts = [int(c.item()) for c in counts]
        cap = max(max(counts), 1)
        pid = torch.zeros(cap, dtype=torch.int64, device=ids.device)
        pid[:ids.numel()] = ids
        rows = torch.zeros((cap, self.V.shape[1]), dtype=self.V.dtype, device=self.V.device)
        rows[:ids.numel()] = self.V[ids] - self.base[ids]
        all_ids = [torch.empty_like(pid) for _ in range(W)]
        all_rows = [torch.empty_like(rows) for _ in range(W)]
        dist.all_gather(all_ids, pid, group=self.group)
        dist.all_gather(all_rows, rows, group=self.group)
        me = dist.get_rank(self.group)
        for r in range(W):                                   # rank order on every rank: base stays bit-identical
            i_r, d_r = all_ids[r][:counts[r]], all_rows[r][:counts[r]]
            if self.scale != 1.0:
                d_r = d_r * self.scale
            self.base.index_add_(0, i_r, d_r)
            if r != me:
                self.V.index_add_(0, i_r, d_r)
            elif self.scale != 1.0:                          # own delta is in V in full: leave the scaled share
                self.V.index_add_(0, i_r, d_r - all_rows[r][:counts[r]])
        self.stats['sparse'] += 1

    def exchange(self, blocking=False, allow_sparse=True):
        self._fold()
        use_sparse = False
        if self.tracking and allow_sparse:
            # ONE decision for all ranks (`tracking` and `allow_sparse` are the same everywhere, so every rank is here):
            # sparse only if no rank declared a launch of the period untracked and the largest touched fraction is small;
            # both travel in one MAX all-reduce. Untracked mode (full-size launches) takes no vote and no host round trip.
            frac = float(self.mark.sum().item()) / self.mark.numel() if self.all_tracked else 0.0
            v = torch.tensor([frac, 0.0 if self.all_tracked else 1.0], dtype=torch.float64, device=self.V.device)
            self.dist.all_reduce(v, op=self.dist.ReduceOp.MAX, group=self.group)
            use_sparse = float(v[1].item()) == 0.0 and float(v[0].item()) * self.world <= self.sparse_fraction
        if use_sparse:
            self._sparse()
        else:
            torch.sub(self.V, self.base, out=self.own)
            self.red.copy_(self.own)
            self.pending = self.dist.all_reduce(self.red, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.stats['dense'] += 1
            if blocking:
                self._fold()
        if self.tracking:
            self.mark.zero_()
        self.all_tracked = True

    def finish(self):
        """Drain: after this call every replica equals the common table bit for bit."""
        self.exchange(blocking=True, allow_sparse=False)
        self.V.copy_(self.base)


def sync_item_table(V, V_snapshot, group=None):
    """Blocking one-shot form of :class:`ItemTableSync` (kept for callers that hold their own snapshot): the deltas
    ``V - V_snapshot`` of all ranks are summed and applied; both tensors end as the new common table. In place: ``V``
    carries the delta through the collective, no table-sized temporary."""
    import torch.distributed as dist
    V.sub_(V_snapshot)
    dist.all_reduce(V, op=dist.ReduceOp.SUM, group=group)
    V_snapshot.add_(V)
    V.copy_(V_snapshot)


def full_sort_topk_sharded(U_rows, V, k=20, pad_col=0, shard='items', group=None):
    """Full-sort top-k of the user rows ``U_rows`` [B, d] over ALL items on ``world`` ranks (SURVEY.md section 8 e,
    scoring: ``src/recbole/inference.py:209-225, 334-350`` loops over session batches on one GPU).

    * ``shard='users'``: rank r scores rows ``[B r / W, B (r + 1) / W)`` against the whole item table; the [B, k] result is
      all-gathered. No merge needed.
    * ``shard='items'``: rank r scores every row against items ``[N r / W, N (r + 1) / W)`` (its slice of a table that
      may be too large or too hot for one GPU), the W partial top-k lists per row are all-gathered ([W, B, k]) and merged
      exactly by (score desc, id asc) on the device (``otto_mf_topk_merge``): top-k of a maximum is mergeable, and the
      per-(row, item) dot products do not depend on the split, so the result equals the unsharded call bit for bit.

    Every rank returns the full (ids int32 [B, k], scores float32 [B, k])."""
    import torch.distributed as dist
    from .engine import topk_merge
    W, r = dist.get_world_size(group), dist.get_rank(group)
    B, N = U_rows.shape[0], V.shape[0]
    if shard == 'users':
        lo, hi = B * r // W, B * (r + 1) // W
        ids = torch.full((B, k), -1, dtype=torch.int32, device=U_rows.device)
        sc = torch.full((B, k), float('-inf'), dtype=torch.float32, device=U_rows.device)
        if hi > lo:
            ids[lo:hi], sc[lo:hi] = score_topk(U_rows[lo:hi].contiguous(), V, k=k, pad_col=pad_col)
        # every row is owned by exactly one rank: MAX merges the disjoint pieces
        dist.all_reduce(ids, op=dist.ReduceOp.MAX, group=group)
        dist.all_reduce(sc, op=dist.ReduceOp.MAX, group=group)
        return ids, sc
    if shard != 'items':
        raise ValueError("shard must be 'users' or 'items'")
    lo, hi = N * r // W, N * (r + 1) // W
    pid = torch.full((B, k), -1, dtype=torch.int32, device=U_rows.device)
    psc = torch.full((B, k), float('-inf'), dtype=torch.float32, device=U_rows.device)
    if hi > lo:
        li, ls = score_topk(U_rows, V[lo:hi].contiguous(), k=min(k, hi - lo), pad_col=(pad_col - lo) if lo <= pad_col < hi else -1)
        kk = li.shape[1]
        pid[:, :kk] = torch.where(li >= 0, li + lo, li)
        psc[:, :kk] = torch.where(li >= 0, ls, torch.full_like(ls, float('-inf')))
    all_id = [torch.empty_like(pid) for _ in range(W)]
    all_sc = [torch.empty_like(psc) for _ in range(W)]
    dist.all_gather(all_id, pid, group=group)
    dist.all_gather(all_sc, psc, group=group)
    return topk_merge(torch.stack(all_sc), torch.stack(all_id), k)
