"""world_size-2 gloo (CPU) test of the multi-GPU covisitation exchange
(otto_amd/covisitation/distributed.py): the real exchange code routes runs produced by a
CPU stand-in engine built on the oracle's per-window expansion; owners reduce with the
oracle; the union must equal the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT

import covis_oracle as co


class OracleEngine:
    """CPU stand-in with the export_runs/import_runs surface of CovisBuilder (tests only)."""
    FK = ('click_cart', 'click_order', 'cart_order', 'click_click')

    def __init__(self):
        self.runs = []   # (x, [rec...], [tw...])

    def feed(self, aid, ts, typ, off, t0, t1):
        sp = co.CovisSpec()
        for s in range(len(off) - 1):
            rows = co.expand_window_python(aid, ts, typ, int(off[s]), int(off[s + 1]), sp, self.FK, t0, t1)
            cur = None
            for x, y, ty, fb, ex in rows:
                if cur is None or cur[0] != x:
                    cur = (x, [], [])
                    self.runs.append(cur)
                cur[1].append(y | ty << 26 | fb << 28)
                cur[2].append(ex)

    def export_runs(self, lo, hi):
        sel = [r for r in self.runs if lo <= r[0] < hi]
        hdr = torch.tensor([[r[0], len(r[1])] for r in sel], dtype=torch.int32).reshape(-1, 2)
        rec = torch.tensor([v for r in sel for v in r[1]], dtype=torch.int64).to(torch.int32)
        tw = torch.tensor([v for r in sel for v in r[2]], dtype=torch.int32)
        return hdr, rec, tw

    def import_runs(self, hdr, rec, tw):
        p = 0
        for x, n in hdr.tolist():
            self.runs.append((x, (rec[p:p + n].to(torch.int64) & 0xFFFFFFFF).tolist(), tw[p:p + n].tolist()))
            p += n

    def reduce(self):
        """{(x, y): (c0, c1, c2, cnt_fbits[4], time W)}"""
        acc = {}
        for x, recs, tws in self.runs:
            for rc, ex in zip(recs, tws):
                y, ty, fb = rc & 0x3FFFFFF, (rc >> 26) & 3, rc >> 28
                a = acc.setdefault((x, y), [0, 0, 0, 0, 0, 0, 0, 0])
                a[ty] += 1
                for f in range(4):
                    a[3 + f] += (fb >> f) & 1
                a[7] += co.Q16 + ex
        return acc


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path[:0] = [ROOT, os.path.join(ROOT, 'oracle')]
    import torch.distributed as dist
    from otto_amd.synth import generate_sessions
    from otto_amd.covisitation.distributed import exchange_runs, owner_bounds, global_ts_range
    dist.init_process_group('gloo', rank=rank, world_size=world)
    ev = generate_sessions(160, n_aids=90, seed=77)
    half = ev.n_sessions // world
    lo, hi = rank * half, (rank + 1) * half if rank < world - 1 else ev.n_sessions
    e0, e1 = int(ev.sess_off[lo]), int(ev.sess_off[hi])
    t0, t1 = global_ts_range(torch.from_numpy(ev.ts[e0:e1]))
    assert (t0, t1) == (int(ev.ts.min()), int(ev.ts.max()))
    local, owner = OracleEngine(), OracleEngine()
    local.feed(ev.aid[e0:e1], ev.ts[e0:e1], ev.type[e0:e1], ev.sess_off[lo:hi + 1] - e0, t0, t1)
    bounds = owner_bounds(ev.n_aids, world)
    sent = exchange_runs(local.export_runs, owner.import_runs, bounds, want_time=True)
    assert sent[1] == sum(len(r[1]) for r in local.runs)
    acc = owner.reduce()
    assert all(bounds[rank] <= x < bounds[rank + 1] for x, _ in acc)
    q.put((rank, acc))
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_world2_matches_single_process():
    from otto_amd.synth import generate_sessions
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    parts = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    merged = {}
    for acc in parts.values():
        assert not (set(acc) & set(merged))
        merged.update(acc)
    ev = generate_sessions(160, n_aids=90, seed=77)
    want = co.covis_pairs_python(ev.aid, ev.ts, ev.type, ev.sess_off, co.CovisSpec())
    for kind, tw in co.TYPE_WEIGHTS.items():
        got = {p: co.Q16 * (a[0] * tw[0] + a[1] * tw[1] + a[2] * tw[2]) for p, a in merged.items()}
        assert got == want[kind], kind
    for f, kind in enumerate(OracleEngine.FK):
        got = {p: co.Q16 * a[3 + f] for p, a in merged.items() if a[3 + f]}
        assert got == want[kind], kind
    assert {p: a[7] for p, a in merged.items()} == want['time_weighted']


def test_owner_bounds_partition():
    from otto_amd.covisitation.distributed import owner_bounds
    for n, w in ((1855603, 8), (10, 3), (7, 8)):
        b = owner_bounds(n, w)
        assert b[0] == 0 and b[-1] == n and all(b[i] <= b[i + 1] for i in range(w))


# ---- data-parallel BPR: ItemTableSync over gloo with the oracle's batch step standing in for the HIP kernel -------------
def _planted_small(seed=0):
    rng = np.random.default_rng(seed)
    n_users, n_items, groups = 1200, 601, 12
    grp = rng.integers(0, groups, n_users)
    item_grp = np.r_[-1, rng.integers(0, groups, n_items - 1)]
    by = [np.flatnonzero(item_grp == g_) for g_ in range(groups)]
    u = np.repeat(np.arange(n_users), 10)
    i = np.array([rng.choice(by[grp[x]]) for x in u])
    return u, i, n_users, n_items


class _CpuBpr:
    """Stand-in for BPR + MFEngine on CPU tensors: `train_epoch` drives it exactly as it drives the HIP engine
    (tests only; the arithmetic is oracle/mf_oracle.bpr_step_batch)."""

    class _Emb:
        def __init__(self, w):
            self.weight = type('P', (), {'data': w})()

    def __init__(self, n_users, n_items, d, seed=1):
        g = torch.Generator().manual_seed(seed)
        self.user_embedding = self._Emb(torch.randn(n_users, d, generator=g) * 0.1)
        self.item_embedding = self._Emb(torch.randn(n_items, d, generator=g) * 0.1)
        self.rng = np.random.default_rng(5)

    def engine(self, batch):
        return self

    def bpr_step(self, U, V, users, items, seed, epoch, row0, lr, l2, mode, loss_sum=None, neg_out=None):
        import mf_oracle as mo
        j = self.rng.integers(0, V.shape[0], users.numel())
        Un, Vn = U.numpy(), V.numpy()                      # views: updated in place
        loss = mo.bpr_step_batch(Un, Vn, users.numpy(), items.numpy(), j, lr, l2)
        if loss_sum is not None:
            loss_sum += loss
        if neg_out is not None:
            neg_out[:users.numel()] = torch.from_numpy(j)


def _bpr_worker(rank, world, port, q, reduce, rows_per_launch, uneven, epochs, script):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path[:0] = [ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')]
    import torch.distributed as dist
    from otto_amd.matrix_factorization.bpr import ItemTableSync, train_epoch
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    u, i, n_users, n_items = _planted_small()
    cut = [n_users * r // world for r in range(world + 1)]
    if uneven:                                              # session-chunk shards of different sizes
        cut = [0] + [min(n_users, c + 37 * (r + 1)) for r, c in enumerate(cut[1:-1])] + [n_users]
    mine = (u >= cut[rank]) & (u < cut[rank + 1])
    model = _CpuBpr(n_users, n_items, 16)
    du, di = torch.from_numpy(u[mine]), torch.from_numpy(i[mine])
    V = model.item_embedding.weight.data
    sync = ItemTableSync(V, reduce=reduce)
    losses = []
    if script == 'epochs':
        for e in range(epochs):
            losses.append(train_epoch(model, du, di, lr=0.2, seed=1, epoch=e, rows_per_launch=rows_per_launch,
                                      row0=int(np.flatnonzero(mine)[0]), sync=sync, sync_every=2))
    else:
        # one period with a launch that is NOT reported and one that is: the exchange must not drop the first one's rows
        sync.tracking = True
        eng, U = model, model.user_embedding.weight.data
        neg = torch.empty(len(du), dtype=torch.int64)
        eng.bpr_step(U, V, du[:3000], di[:3000], 1, 0, 0, 0.2, 0.0, 0, neg_out=neg)
        sync.untracked()
        eng.bpr_step(U, V, du[3000:3040], di[3000:3040], 1, 0, 0, 0.2, 0.0, 0, neg_out=neg)
        sync.touched(di[3000:3040], neg[:40])
        sync.finish()
    q.put((rank, V.numpy().copy(), losses, int(mine.sum()), dict(sync.stats)))
    dist.barrier()
    dist.destroy_process_group()


def _run_bpr(world, reduce='sum', rows_per_launch=1024, uneven=False, epochs=20, script='epochs'):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_bpr_worker, args=(r, world, port, q, reduce, rows_per_launch, uneven, epochs, script))
             for r in range(world)]
    for p in procs:
        p.start()
    res = dict((r[0], r[1:]) for r in (q.get(timeout=300) for _ in procs))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


def _single_rank_losses(rows_per_launch, epochs):
    sys.path[:0] = [os.path.join(ROOT, 'oracle')]
    from otto_amd.matrix_factorization.bpr import train_epoch
    u, i, n_users, n_items = _planted_small()
    model = _CpuBpr(n_users, n_items, 16)
    du, di = torch.from_numpy(u), torch.from_numpy(i)
    return [train_epoch(model, du, di, lr=0.2, seed=1, epoch=e, rows_per_launch=rows_per_launch) for e in range(epochs)]


def test_data_parallel_bpr_world4_loss_band_two_sided():
    """W = 4 ranks, session-chunk shards, the item table exchanged by ItemTableSync over gloo (dense, asynchronous: launches
    of 1024 rows > n_items / 4): the row-weighted mean loss of the last epoch is within +-10 % of the single-process run
    with reduce='sum' (every triplet's step applied once, as the single process applies it) and replicas are identical.
    reduce='mean' (model averaging) is the conservative variant: it must converge too, but more slowly -- its final
    loss lies ABOVE the single-process loss."""
    ref = _single_rank_losses(1024, 20)
    for reduce in ('sum', 'mean'):
        res = _run_bpr(4, reduce=reduce)
        for r in range(1, 4):
            assert np.array_equal(res[0][0], res[r][0]), 'item-table replicas differ after the drain'
        n = np.array([res[r][2] for r in range(4)], dtype=np.float64)
        loss = sum(np.array(res[r][1]) * n[r] for r in range(4)) / n.sum()
        assert loss[-1] < 0.2 * loss[0]
        if reduce == 'sum':
            assert 0.9 * ref[-1] <= loss[-1] <= 1.1 * ref[-1], (loss[-1], ref[-1])
        else:
            assert ref[-1] < loss[-1] <= 2.0 * ref[-1], (loss[-1], ref[-1])
        assert res[0][3]['dense'] > 0 and res[0][3]['sparse'] == 0


def test_data_parallel_bpr_unequal_shards_issue_the_same_collectives():
    """Shards of 300 + 37 r users: the ranks' launch counts differ (tracked launches of 128 rows: sparse exchanges, with a
    vote all-reduce each) -- a rank that runs out of rows must keep taking part in the exchange schedule. A mismatch hangs
    (the workers are joined with a timeout) or leaves different replicas."""
    res = _run_bpr(3, rows_per_launch=128, uneven=True, epochs=2)
    rows = [res[r][2] for r in range(3)]
    assert len({(n + 127) // 128 for n in rows}) > 1, rows          # the launch counts do differ
    for r in range(1, 3):
        assert np.array_equal(res[0][0], res[r][0])
    assert res[0][3]['dense'] + res[0][3]['sparse'] >= 2 * ((max(rows) + 127) // 128 // 2)   # every slot pair was exchanged


def test_item_table_sync_untracked_launch_makes_the_period_dense():
    """A period with one launch that was not reported to touched() and one that was: a sparse exchange would ship only
    the second launch's rows and the drain would then overwrite the first launch's updates with the common table. Every
    rank's replica must contain BOTH launches' deltas of every rank = the sum the dense path forms."""
    res = _run_bpr(2, script='period')
    assert np.array_equal(res[0][0], res[1][0])
    assert res[0][3] == {'dense': 1, 'sparse': 0}
    # reference: both ranks' deltas on the common start table, computed without any exchange machinery
    import mf_oracle as mo
    u, i, n_users, n_items = _planted_small()
    total = None
    for rank in range(2):
        lo, hi = n_users * rank // 2, n_users * (rank + 1) // 2
        mine = (u >= lo) & (u < hi)
        m = _CpuBpr(n_users, n_items, 16)
        V0 = m.item_embedding.weight.data.clone()
        du, di = torch.from_numpy(u[mine]), torch.from_numpy(i[mine])
        U, V = m.user_embedding.weight.data, m.item_embedding.weight.data
        m.bpr_step(U, V, du[:3000], di[:3000], 1, 0, 0, 0.2, 0.0, 0)
        m.bpr_step(U, V, du[3000:3040], di[3000:3040], 1, 0, 0, 0.2, 0.0, 0)
        total = (V - V0) if total is None else total + (V - V0)
    want = (V0 + total).numpy()
    assert np.allclose(res[0][0], want, rtol=0, atol=1e-6)
