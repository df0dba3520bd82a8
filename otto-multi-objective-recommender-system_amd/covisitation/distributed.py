"""Multi-GPU covisitation build: one process per GPU, sessions sharded by contiguous
session-chunk, ONE exchange step over RCCL/xGMI (SURVEY.md section 8 e).

Merging per-GPU top-k lists would be inexact (a pair's weight is a sum over
shards), so what is exchanged is the expanded pairs themselves: every rank runs
the pair-expand kernel on its session chunk, the (window, aid_x) runs are routed
to the owner of ``aid_x`` (disjoint aid ranges -- the reference's disk-part idea,
``src/covisitation/inference.py:87-89``) with a single all-to-all-v, and each owner
reduces and selects its aid range locally.  The result is bit-identical to the
single-GPU build.  xGMI is point-to-point, so an all-to-all (every link busy at
once) rather than a ring collective is the right shape; records are 4 bytes/pair.
"""
import numpy as np


def owner_bounds(n_aids, world):
    """world+1 cut points of the aid_x owner ranges."""
    return [int(round(i * n_aids / world)) for i in range(world + 1)]


def exchange_runs(export_fn, import_fn, bounds, group=None, want_time=False, stage_device=None, export_all_fn=None,
                  reserve_fn=None):
    """Route runs to their aid_x owners with one all-to-all-v per array.

    ``export_all_fn(bounds) -> (hdr [n,2], rec, tw|None, runs_per_owner, recs_per_owner)`` (engine.export_all: every
    owner's piece already owner-major in one buffer) or, per owner, ``export_fn(lo, hi) -> (hdr, rec, tw|None)``;
    ``import_fn(hdr, rec, tw)`` on the owner engine.  ``stage_device`` (e.g. ``'cpu'`` with a gloo group): move the
    buffers there for the collectives and back -- used to rehearse the multi-rank path on a single GPU and in the
    CPU tests; with nccl (= RCCL over xGMI) leave it None.  ``reserve_fn(n_recs) -> (rec, tw|None)`` (engine.import_reserve):
    receive the records straight into the owner engine's arrays (no copy on import; ignored when staging).
    Returns (runs_sent, recs_sent, runs_received, recs_received).
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if export_all_fn is not None:
        hdr_s, rec_s, tw_s, runs, recs = export_all_fn(bounds)
    else:
        pieces = [export_fn(bounds[r], bounds[r + 1]) for r in range(world)]
        runs = [p[0].shape[0] for p in pieces]
        recs = [p[1].numel() for p in pieces]
        hdr_s = torch.cat([p[0].reshape(-1, 2) for p in pieces])
        rec_s = torch.cat([p[1] for p in pieces])
        tw_s = torch.cat([p[2] for p in pieces]) if want_time else None
    home = hdr_s.device
    if stage_device is not None:
        hdr_s, rec_s = hdr_s.to(stage_device), rec_s.to(stage_device)
        tw_s = None if tw_s is None else tw_s.to(stage_device)
    dev = hdr_s.device
    send_counts = torch.tensor([runs, recs], dtype=torch.int64, device=dev).t().contiguous()     # [world, 2]
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts, group=group)
    sc = send_counts.cpu().numpy()
    rc = recv_counts.cpu().numpy()

    inplace = (None, None)
    if reserve_fn is not None and stage_device is None and int(rc[:, 1].sum()) > 0:
        inplace = reserve_fn(int(rc[:, 1].sum()))

    def a2a(send, per_item, s_cnt, r_cnt, out=None):
        recv = out if out is not None else torch.empty(int(r_cnt.sum()) * per_item, dtype=torch.int32, device=dev)
        dist.all_to_all_single(recv, send.reshape(-1), [int(v) * per_item for v in r_cnt], [int(v) * per_item for v in s_cnt],
                               group=group)
        return recv

    hdr = a2a(hdr_s, 2, sc[:, 0], rc[:, 0]).reshape(-1, 2)
    rec = a2a(rec_s, 1, sc[:, 1], rc[:, 1], inplace[0])
    tw = a2a(tw_s, 1, sc[:, 1], rc[:, 1], inplace[1]) if want_time else None
    if stage_device is not None:
        hdr, rec, tw = hdr.to(home), rec.to(home), None if tw is None else tw.to(home)
    import_fn(hdr.contiguous(), rec if rec.is_contiguous() else rec.contiguous(), tw)
    return int(sc[:, 0].sum()), int(sc[:, 1].sum()), int(rc[:, 0].sum()), int(rc[:, 1].sum())


def exchange_runs_chunked(engine, owner, bounds, group=None, want_time=False, n_chunks=4):
    """The same exchange cut into ``n_chunks`` ranges of run slots, so that the export of range c + 1 (a pass over its run
    descriptors + the record copies, ~6 ms for a full-size shard) runs on the compute stream while range c is on the links:
    the collectives are issued asynchronously right behind the fill they depend on (torch.distributed orders a collective
    after the work already queued on the current stream, nothing more). All counts are planned and exchanged first (one
    small all-to-all), the records of every range are received straight into the owner engine's arrays
    (``import_reserve``), range after range, and registered with ONE ``import_runs`` at the end. Device tensors only (RCCL
    or a world of one); the staged rehearsal path stays ``exchange_runs``.
    Returns (runs_sent, recs_sent, runs_received, recs_received)."""
    import torch
    import torch.distributed as dist
    W = dist.get_world_size(group)
    slots = engine.run_slots()
    K = max(1, min(int(n_chunks), slots if slots else 1))
    cuts = [slots * c // K for c in range(K + 1)]
    plans = [engine.export_plan_range(bounds, cuts[c], cuts[c + 1]) for c in range(K)]
    dev = engine.device
    # counts: [W dest][K][2] -> every source's [K][2] for me
    send_counts = torch.tensor([[[plans[c][0][o], plans[c][1][o]] for c in range(K)] for o in range(W)], dtype=torch.int64, device=dev)
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts.view(-1), send_counts.view(-1), group=group)
    rc = recv_counts.cpu().numpy()                          # [W src][K][2]
    tot_runs_in, tot_recs_in = int(rc[:, :, 0].sum()), int(rc[:, :, 1].sum())
    rec_in, tw_in = owner.import_reserve(tot_recs_in) if tot_recs_in else (None, None)
    if rec_in is None:
        rec_in = torch.empty(0, dtype=torch.int32, device=dev)
    hdr_in = torch.empty((tot_runs_in, 2), dtype=torch.int32, device=dev)
    handles, keep = [], []
    r0 = c0 = 0
    for c in range(K):
        runs, recs = plans[c]
        hdr_s = torch.empty((sum(runs), 2), dtype=torch.int32, device=dev)
        rec_s = torch.empty(sum(recs), dtype=torch.int32, device=dev)
        tw_s = torch.empty(sum(recs), dtype=torch.int32, device=dev) if want_time else None
        engine.export_fill_range(bounds, cuts[c], cuts[c + 1], runs, recs, hdr_s, rec_s, tw_s)
        nr, nc = int(rc[:, c, 0].sum()), int(rc[:, c, 1].sum())
        jobs = [(hdr_in[r0:r0 + nr].view(-1), hdr_s.view(-1), [int(v) * 2 for v in rc[:, c, 0]], [int(v) * 2 for v in runs]),
                (rec_in[c0:c0 + nc], rec_s, [int(v) for v in rc[:, c, 1]], [int(v) for v in recs])]
        if want_time:
            jobs.append((tw_in[c0:c0 + nc], tw_s, [int(v) for v in rc[:, c, 1]], [int(v) for v in recs]))
        for out, inp, osz, isz in jobs:
            handles.append(dist.all_to_all_single(out, inp, osz, isz, group=group, async_op=True))
        keep.append((hdr_s, rec_s, tw_s))                   # send buffers stay alive until the collectives are done
        r0 += nr
        c0 += nc
    for h in handles:
        h.wait()
    if tot_runs_in:
        owner.import_runs(hdr_in, rec_in[:tot_recs_in], tw_in[:tot_recs_in] if want_time else None)
    sent = (sum(sum(p_[0]) for p_ in plans), sum(sum(p_[1]) for p_ in plans))
    return sent[0], sent[1], tot_runs_in, tot_recs_in


def global_ts_range(ts, group=None):
    """t0 / t1 of the time weight must be global over all ranks (SPEC-COVIS 6)."""
    import torch
    import torch.distributed as dist
    lo = ts.min().to(torch.int64).reshape(1) if ts.numel() else torch.full((1,), 2 ** 62, dtype=torch.int64, device=ts.device)
    hi = ts.max().to(torch.int64).reshape(1) if ts.numel() else torch.full((1,), -2 ** 62, dtype=torch.int64, device=ts.device)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    return int(lo.item()), int(hi.item())


class ShardedCovisBuilder:
    """Session-chunk sharded build on ``world`` GPUs; rank r owns aid_x in
    ``[bounds[r], bounds[r+1])`` and returns top-k rows for that range only."""

    def __init__(self, n_aids, kinds, ts_min, ts_max, device, group=None, window=30, max_gap=86400, stage_device=None,
                 exchange_chunks=4):
        import torch.distributed as dist
        from .engine import CovisBuilder
        self.group = group
        self.stage_device = stage_device
        self.exchange_chunks = exchange_chunks    # device path: slot ranges in flight (export of c + 1 under the send of c); 0 = one piece
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.bounds = owner_bounds(n_aids, self.world)
        kw = dict(kinds=kinds, window=window, max_gap=max_gap, ts_min=ts_min, ts_max=ts_max, device=device)
        self.local = CovisBuilder(n_aids, **kw)    # expands this rank's session chunk
        self.owner = CovisBuilder(n_aids, **kw)    # holds the runs of this rank's aid range

    def reset(self):
        self.local.reset()
        self.owner.reset()

    def feed(self, aid, ts, typ, sess_off):
        self.local.feed(aid, ts, typ, sess_off)

    def finalize(self, k=20, out=None):
        if self.stage_device is None and self.exchange_chunks and self.exchange_chunks > 1:
            self.last_exchange = exchange_runs_chunked(self.local, self.owner, self.bounds, self.group, self.local.want_time,
                                                       self.exchange_chunks)
            return self.owner.finalize(k=k, out=out)
        self.last_exchange = exchange_runs(self.local.export_runs, self.owner.import_runs, self.bounds, self.group,
                                           self.local.want_time, self.stage_device, export_all_fn=self.local.export_all,
                                           reserve_fn=self.owner.import_reserve)
        return self.owner.finalize(k=k, out=out)
