"""MF half of bench.py's metric (benchmark harness, not product code): BPR triplets/s (hogwild SGD, the north star's
mode) on the OTTO-shape (session, aid) stream, plus the reference-config R-MF (SparseAdam) samples/s and the full-sort
scoring rate.

N > 1: every rank owns the rows of its own session chunk (user rows are rank-private), the item table is replicated and
synchronised every ``SYNC_EVERY`` launches by ``bpr.ItemTableSync`` over RCCL (launches of 16.7 M rows touch nearly every
item row: asynchronous dense all-reduce of the delta table, folded in one period later; small launches: sparse (row id,
delta row) all-gather). ``--scaling weak``: ``--mf-rows`` rows per rank; ``strong``: ``--mf-rows`` rows split over the ranks."""
import time

import numpy as np

ROWS_PER_LAUNCH = 1 << 24
SYNC_EVERY = 4
HBM_PEAK_GBS = 8000.0


def run(a, dev, rank, world, cpu_baseline_fn=None, traffic_fn=None):
    import torch
    import torch.distributed as dist
    from otto_amd.synth import generate_sessions_torch, OTTO_N_AIDS, OTTO_N_SESSIONS
    from otto_amd.matrix_factorization.engine import MFEngine, BPR_HOGWILD
    from otto_amd.matrix_factorization.bpr import ItemTableSync
    d = a.mf_factors
    n_users, n_items = (OTTO_N_SESSIONS if a.sessions >= OTTO_N_SESSIONS else a.sessions), OTTO_N_AIDS
    data = generate_sessions_torch(n_users, n_aids=n_items, seed=142 + rank, device=dev, pop_seed=142)   # one item catalogue on every rank
    E = data['aid'].numel()
    rows = min(a.mf_rows if a.scaling == 'weak' else a.mf_rows // world, E)
    lens = data['sess_off'][1:] - data['sess_off'][:-1]
    users = torch.repeat_interleave(torch.arange(n_users, device=dev), lens, output_size=E)
    perm = torch.randperm(E, device=dev)[:rows]            # SGD visits rows in random order
    users = users[perm].contiguous()
    items = data['aid'].to(torch.int64)[perm].contiguous()
    del data, perm, lens
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    U = (torch.randn(n_users, d, device=dev, generator=g) * 0.1).contiguous()
    V = (torch.randn(n_items, d, device=dev, generator=g) * 0.1).contiguous()
    sync = ItemTableSync(V) if world > 1 else None
    eng = MFEngine(n_users, n_items, d, ROWS_PER_LAUNCH, device=dev)
    n_launch = (rows + ROWS_PER_LAUNCH - 1) // ROWS_PER_LAUNCH
    n_slots = sync.global_max(n_launch) if sync is not None else n_launch   # one exchange schedule for all ranks
    loss = torch.zeros(n_launch, device=dev)
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(n_launch)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(n_launch)]

    def epoch(e, timed):
        for q in range(n_slots):
            lo, hi = min(rows, q * ROWS_PER_LAUNCH), min(rows, (q + 1) * ROWS_PER_LAUNCH)
            if hi > lo:
                if timed:
                    ev0[q].record()
                eng.bpr_step(U, V, users[lo:hi], items[lo:hi], 42, e, lo, 0.05, 0.0, BPR_HOGWILD, loss_sum=loss[q:q + 1])
                if timed:
                    ev1[q].record()
            if world > 1 and (q + 1) % SYNC_EVERY == 0:
                sync.exchange()              # launches of 16.7 M rows are not tracked (ItemTableSync.tracking False): dense, asynchronous
        if world > 1:
            sync.finish()                    # drain: replicas bit-identical at the end of the epoch (as bpr.train_epoch does)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    eng.bpr_step(U, V, users[:ROWS_PER_LAUNCH], items[:ROWS_PER_LAUNCH], 42, 99, 0, 0.05, 0.0, BPR_HOGWILD, loss_sum=loss[:1])
    first_loss = None
    barrier()
    t0 = time.perf_counter()
    kms = 0.0
    for e in range(a.steps):
        epoch(e, True)
        torch.cuda.synchronize()
        kms += sum(s.elapsed_time(t) for s, t in zip(ev0, ev1))
        if first_loss is None:
            first_loss = float(loss.sum().item()) / rows
    barrier()
    dt = time.perf_counter() - t0
    last_loss = float(loss.sum().item()) / rows
    if world > 1:
        mx = torch.tensor([dt], dtype=torch.float64, device='cpu' if dist.get_backend() == 'gloo' else dev)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dt = float(mx.item())
    total = rows * a.steps * world
    per_triplet = 16 + 6 * 4 * d        # two int64 indices + three fp32 rows read and written back
    launch_ms = kms / (a.steps * n_launch)
    gbs = per_triplet * (rows / n_launch) / (launch_ms * 1e-3) / 1e9
    res = {
        'metric': 'BPR triplets/sec MF train', 'value': round(total / dt, 1), 'unit': 'triplets/s',
        'ms_per_epoch': round(1e3 * dt / a.steps, 3), 'epochs': a.steps, 'dtype': 'f32',
        'config': {'workload': f'BPR-MF hogwild SGD, {n_users} sessions x {n_items} aids x {d}-d fp32, {rows} (session, aid) rows per GPU '
                               f'in random order, uniform negatives from the counter RNG, launches of {ROWS_PER_LAUNCH}',
                   'rows_per_gpu': rows, 'factors': d,
                   'parallelism': 'single GPU' if world == 1 else f'data-parallel x{world} ({a.scaling} scaling): user rows private, item table synchronised every {SYNC_EVERY} launches by ItemTableSync (RCCL; exchanges this run: {dict(sync.stats) if sync is not None else {}}; dense = asynchronous all-reduce of the delta table folded in one period later, sparse = (id, delta row) all-gather)'},
        'roofline': {'kernel': 'k_bpr_hogwild', 'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': round(gbs / HBM_PEAK_GBS, 4),
                     'traffic': traffic_fn(('k_bpr_hogwild',)) if (traffic_fn and d == 64 and rows >= ROWS_PER_LAUNCH) else None,
                     'avg_ms': round(launch_ms, 4),
                     'algorithmic_bytes': int(per_triplet * (rows / n_launch))},
        'mean_loss_first_epoch': round(first_loss, 5), 'mean_loss_last_epoch': round(last_loss, 5),
    }
    if rank == 0 and world == 1:
        # reference-config R-MF: MSELoss + SparseAdam, 32 factors, batch 262,144 (models/matrix_factorization/config.yaml)
        B, dr = 262144, 32
        E1 = torch.randn(n_users, dr, device=dev, generator=g)
        E2 = torch.randn(n_items + 1, dr, device=dev, generator=g)
        st = [torch.zeros_like(E1), torch.zeros_like(E1), torch.zeros_like(E2), torch.zeros_like(E2)]
        er = MFEngine(n_users, n_items + 1, dr, B, device=dev)
        tg = torch.randint(0, 3, (rows,), device=dev)
        lo_ = torch.zeros(1, device=dev)
        nb = min(40, rows // B)
        for w in range(2):
            er.step_sparse_adam(E1, st[0], st[1], E2, st[2], st[3], users[:B], items[:B], tg[:B], 0, 0.05, (0.9, 0.999), 1e-8, w + 1, lo_)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for b in range(nb):
            sl = slice(b * B, (b + 1) * B)
            er.step_sparse_adam(E1, st[0], st[1], E2, st[2], st[3], users[sl], items[sl], tg[sl], 0, 0.05, (0.9, 0.999), 1e-8, b + 3, lo_)
        torch.cuda.synchronize()
        dtr = time.perf_counter() - t1
        # algorithmic bytes of one step (SURVEY.md section 8 d): 24 B of indices/target + 2 gathered rows per sample, and
        # per UNIQUE touched row p, m, v read and written once (6 x 4d)
        uniq = sum(int(torch.unique(c[b * B:(b + 1) * B]).numel()) for c in (users, items) for b in range(min(nb, 4))) / min(nb, 4)
        rmf_bytes = B * (24 + 2 * 4 * dr) + uniq * 6 * 4 * dr
        rmf_ms = 1e3 * dtr / nb
        res['rmf_sparse_adam'] = {'value': round(nb * B / dtr, 1), 'unit': 'samples/s', 'ms_per_batch': round(rmf_ms, 3),
                                  'config': f'MatrixFactorization MSELoss SparseAdam, {dr} factors, batch {B} (reference config.yaml)',
                                  'roofline': {'kernel': 'k_rmf_fwd + k_rmf_acc + k_rmf_apply (one optimizer step)', 'bound': 'hbm',
                                               'achieved': round(rmf_bytes / (rmf_ms * 1e-3) / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                               'frac': round(rmf_bytes / (rmf_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), 'traffic': None,
                                               'avg_ms': round(rmf_ms, 4), 'algorithmic_bytes': int(rmf_bytes),
                                               'unique_rows_per_batch': int(uniq)}}
        del E1, E2, st, er
        # full-sort scoring (recbole/inference.py:334: test batches of 4096 sessions x all items), d = 128 (config 5)
        from otto_amd.matrix_factorization.engine import score_topk
        Bs, ds = 4096, 128
        Us = torch.randn(Bs, ds, device=dev, generator=g)
        Vs = torch.randn(n_items + 1, ds, device=dev, generator=g)      # + PAD row 0
        score_topk(Us, Vs, k=20, pad_col=0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3
        e0.record()
        for _ in range(reps):
            ids, sc = score_topk(Us, Vs, k=20, pad_col=0)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        flop = 2.0 * Bs * (n_items + 1) * ds
        res['score_topk'] = {'value': round(Bs / (ms * 1e-3), 1), 'unit': 'sessions/s', 'ms_per_batch': round(ms, 3),
                             'config': f'B={Bs} sessions x N={n_items + 1} items x d={ds} fp32, fused top-20, PAD column 0',
                             'roofline': {'kernel': 'k_score<128> + k_score_merge', 'bound': 'mfma', 'achieved': round(flop / (ms * 1e-3) / 1e12, 2),
                                          'peak': 157.3, 'unit': 'TFLOP/s', 'frac': round(flop / (ms * 1e-3) / 1e12 / 157.3, 4), 'traffic': None}}
        del Us, Vs
        if cpu_baseline_fn is not None:
            res['cpu_baseline'] = cpu_baseline_fn(U, V, users, items, n_items, 2_000_000)
            res['rmf_sparse_adam']['cpu_baseline'] = rmf_cpu_baseline(users, items, tg, n_users, n_items + 1, dr, B)
        # BASELINE config 5's trainer shape: the same hogwild epoch at d = 128
        del U, V, eng
        torch.cuda.empty_cache()
        d2 = 128
        U2 = (torch.randn(n_users, d2, device=dev, generator=g) * 0.1).contiguous()
        V2 = (torch.randn(n_items, d2, device=dev, generator=g) * 0.1).contiguous()
        eng2 = MFEngine(n_users, n_items, d2, ROWS_PER_LAUNCH, device=dev)
        l2 = torch.zeros(1, device=dev)

        def epoch128(e):
            for q in range(n_launch):
                lo, hi = q * ROWS_PER_LAUNCH, min(rows, (q + 1) * ROWS_PER_LAUNCH)
                eng2.bpr_step(U2, V2, users[lo:hi], items[lo:hi], 42, e, lo, 0.05, 0.0, BPR_HOGWILD, loss_sum=l2)
        epoch128(99)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for e in range(2):
            epoch128(e)
        e1.record()
        torch.cuda.synchronize()
        ms128 = e0.elapsed_time(e1) / 2
        b128 = (16 + 6 * 4 * d2) * rows
        res['d128'] = {'value': round(rows / (ms128 * 1e-3), 1), 'unit': 'triplets/s', 'ms_per_epoch': round(ms128, 3),
                       'config': f'BPR-MF hogwild SGD, {n_users} x {n_items} x {d2}-d fp32, {rows} rows (BASELINE config 5 trainer shape, 1 GPU)',
                       'roofline': {'kernel': 'k_bpr_hogwild', 'bound': 'hbm', 'achieved': round(b128 / (ms128 * 1e-3) / 1e9, 1),
                                    'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(b128 / (ms128 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                    'traffic': None, 'algorithmic_bytes': int(b128)}}
        del U2, V2, eng2
    return res


def rmf_cpu_baseline(users, items, target, n_users, n_items, d, B, n_batches=20):
    """SURVEY.md section 8 d (2): the reference's own CPU path for a3 / a4 -- ``nn.Embedding(sparse=True)`` x 2, MSELoss,
    ``SparseAdam`` (``src/matrix_factorization/torch_modules.py:8-19``, ``torch_trainer.py:59-78``) -- at the reference
    config (32 factors, batch 262,144): 20 timed batches at the best of a few thread counts. (On the 256-thread host of
    the GPU box one batch takes ~12 s with every thread and ~0.2 s with 8: the sparse update is a few large memory
    passes, so the thread count is probed with 2 batches each, smallest first, and a count that is 3x slower than the best
    ends the probe.)"""
    import os
    import torch
    cores = os.cpu_count() or 1
    n = min(n_batches + 16, users.numel() // B)
    u, i, t = users[:n * B].cpu(), items[:n * B].cpu(), target[:n * B].cpu().float()
    E1 = torch.nn.Embedding(n_users, d, sparse=True)
    E2 = torch.nn.Embedding(n_items, d, sparse=True)
    opt = torch.optim.SparseAdam(list(E1.parameters()) + list(E2.parameters()), lr=0.05)
    crit = torch.nn.MSELoss()
    state = {'b': 0}

    def run(k):
        t0 = time.perf_counter()
        for _ in range(k):
            b = state['b'] % n
            state['b'] += 1
            sl = slice(b * B, (b + 1) * B)
            opt.zero_grad()
            loss = crit((E1(u[sl]) * E2(i[sl])).sum(1), t[sl])
            loss.backward()
            opt.step()
            loss.item()
        return (time.perf_counter() - t0) / k
    torch.set_num_threads(min(cores, 8))
    run(2)                                                    # warm-up: optimizer state allocation
    probe = {}
    for th in sorted({min(cores, 8), min(cores, 32), min(cores, 128), cores}):
        torch.set_num_threads(th)
        probe[th] = run(2)
        if probe[th] > 3 * min(probe.values()):
            break
    best = min(probe, key=probe.get)
    torch.set_num_threads(best)
    per = run(n_batches)
    return {'value': round(B / per, 1), 'unit': 'samples/s', 'cores': best, 'kind': 'port',
            'sample': f'{n_batches} batches of {B} of the same stream, torch {torch.__version__} CPU nn.Embedding(sparse=True) + MSELoss + '
                      f'SparseAdam at {best} threads ({per:.3f} s per batch); probe, s per batch by threads: '
                      + ', '.join(f'{k}: {v:.2f}' for k, v in sorted(probe.items())), 'host_cores_available': cores}
