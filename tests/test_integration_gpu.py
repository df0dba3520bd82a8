"""GPU integration tests at the level of the reference's own entry points: ``train()`` /
``validate()`` (torch_trainer.py:24-161), the trainer script body with the reference's YAML
schema, and the covisitation builder script writing the parquet parts its consumers read."""
import importlib
import os

import numpy as np
import pandas as pd
import pytest
import torch

import covis_oracle as co
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture()
def otto_root(tmp_path, monkeypatch):
    monkeypatch.setenv('OTTO_ROOT', str(tmp_path))
    import otto_amd.settings as settings
    importlib.reload(settings)
    import otto_amd.covisitation.builder as builder
    import otto_amd.matrix_factorization.torch_trainer as trainer
    builder.settings = settings
    trainer.settings = settings
    return tmp_path


@pytest.mark.parametrize('p,cls,loss', [('mf_', 'MatrixFactorization', 'MSELoss'), ('cf_', 'CollaborativeFiltering', 'BCEWithLogitsLoss')])
def test_train_validate_reproduce_reference_epochs(gpu_device, p, cls, loss):
    """Same call sequence as torch_trainer.py:352-404 on the golden batches -> the reference's own
    train()/validate() epoch values within 1e-4."""
    from otto_amd.matrix_factorization import torch_modules, torch_optim, torch_trainer
    g = np.load(os.path.join(GOLDEN, 'mf_golden.npz'))
    n1, n2, d, B, nb, ne, step_size = g[p + 'hyper'].tolist()
    if cls == 'MatrixFactorization':
        model = torch_modules.MatrixFactorization(n_sessions=n1, n_aids=n2, n_factors=d, sparse=True, dropout_probability=0.)
        with torch.no_grad():
            model.session_embeddings.weight.copy_(torch.from_numpy(g[p + 'w1_0']))
            model.aid_embeddings.weight.copy_(torch.from_numpy(g[p + 'w2_0']))
        keys = ('session', 'aid')
    else:
        model = torch_modules.CollaborativeFiltering(n_embeddings=n1, n_factors=d, sparse=True, dropout_probability=0.)
        with torch.no_grad():
            model.embeddings.weight.copy_(torch.from_numpy(g[p + 'w1_0']))
        keys = ('x1', 'x2')
    model.to(gpu_device)
    loader = [({keys[0]: torch.from_numpy(g[p + 'i1'][b]), keys[1]: torch.from_numpy(g[p + 'i2'][b]),
                'target': torch.from_numpy(g[p + 'target'][b])}, None) for b in range(nb)]
    criterion = getattr(torch.nn, loss)()
    optimizer = torch_optim.SparseAdam(model.parameters(), lr=float(g[p + 'lr']), betas=(0.9, 0.999))
    scheduler = torch.optim.lr_scheduler.StepLR(optimizer, step_size=step_size, gamma=0.5, last_epoch=-1)
    for e in range(ne):
        tl = torch_trainer.train(loader, model, criterion, optimizer, gpu_device, scheduler)
        vl, sc = torch_trainer.validate(loader, model, criterion, gpu_device, scores=True)
        assert tl == pytest.approx(g[p + 'epoch_train_loss'][e], rel=1e-4)
        assert vl == pytest.approx(g[p + 'epoch_val_loss'][e], rel=1e-4)
        assert list(sc) == g[p + 'score_names'].tolist()
        vals = list(sc.values())
        assert vals[0] == pytest.approx(g[p + 'epoch_val_s0'][e], rel=1e-4, abs=2.0 / (nb * B))
        # second score: mean squared error (MF) / ROC-AUC (CF), both from the device-side sums / sort; 1e-4 relative
        assert vals[1] == pytest.approx(g[p + 'epoch_val_s1'][e], rel=1e-4)
    out = model(loader[0][0][keys[0]].to(gpu_device), loader[0][0][keys[1]].to(gpu_device))
    assert out.shape == (B,) and out.dtype == torch.float32
    sd = model.state_dict()
    assert set(sd) == ({'session_embeddings.weight', 'aid_embeddings.weight'} if cls == 'MatrixFactorization' else {'embeddings.weight'})
    w = sd[list(sd)[0]].cpu().numpy()
    assert np.linalg.norm(w - g[p + 'w1_T']) <= 1e-3 * np.linalg.norm(g[p + 'w1_T'])


def test_trainer_script_body_with_reference_yaml_schema(gpu_device, otto_root):
    """YAML keys of models/matrix_factorization/config.yaml; sessions_aids.parquet in, model_best.pt out."""
    from otto_amd.matrix_factorization import torch_trainer
    from otto_amd.synth import generate_sessions
    ev = generate_sessions(300, n_aids=200, seed=1)
    data = otto_root / 'data' / 'matrix_factorization'
    data.mkdir(parents=True)
    pd.DataFrame({'session': ev.session_ids(), 'aid': ev.aid.astype(np.int64), 'target': ev.type.astype(np.int64)}).to_parquet(
        data / 'sessions_aids.parquet')
    config = {
        'dataset': {'load_dataset': True},
        'model': {'model_class': 'MatrixFactorization', 'model_checkpoint_path': None, 'n_sessions': 300, 'n_aids': 200,
                  'n_factors': 32, 'sparse': True, 'dropout_probability': 0.},
        'training': {'training_batch_size': 1024, 'validation_batch_size': 1024, 'scores': True, 'loss_function': 'MSELoss',
                     'loss_args': {}, 'optimizer': 'SparseAdam', 'optimizer_args': {'lr': 0.05, 'betas': [0.9, 0.999]},
                     'lr_scheduler': 'StepLR', 'lr_scheduler_args': {'step_size': 5000, 'gamma': 0.5, 'last_epoch': -1},
                     'epochs': 6, 'early_stopping_patience': 20, 'random_state': 42, 'deterministic_cudnn': False,
                     'device': str(gpu_device)},
        'persistence': {'model_directory': 'matrix_factorization', 'visualize_learning_curve': True, 'save_best_model': True,
                        'save_epoch_model': [2]},
    }
    model, summary, scores = torch_trainer.run(config)
    assert set(summary) == {'train_loss', 'val_loss', 'val_mean_absolute_error', 'val_mean_squared_error'}
    assert len(summary['train_loss']) == 6 and summary['train_loss'][-1] < summary['train_loss'][0]
    mdir = otto_root / 'models' / 'matrix_factorization'
    sd = torch.load(mdir / 'model_best.pt', weights_only=True)
    assert set(sd) == {'session_embeddings.weight', 'aid_embeddings.weight'} and sd['aid_embeddings.weight'].shape == (200, 32)
    assert (mdir / 'model_epoch_2.pt').exists() and (mdir / 'learning_curve.png').exists()


def test_builder_script_writes_consumer_parts(gpu_device, otto_root):
    """builder.py validation: splits parquet in -> top_15_<kind>_<i>.pqt / top_<kind>_<i>.pqt out, with the
    hard-coded part counts of the consumers, read back the way covisitation_df_to_dict does."""
    from otto_amd.covisitation import builder
    from otto_amd.covisitation.spec import REFERENCE_KINDS
    from otto_amd.synth import generate_sessions
    ev = generate_sessions(1200, n_aids=400, seed=8)
    fr = ev.to_frame()
    fr['session'] = fr['session'] + 11_000_000
    splits = otto_root / 'data' / 'splits'
    splits.mkdir(parents=True)
    cut = len(fr) // 2
    cut = int(np.searchsorted(fr['session'].to_numpy(), fr['session'].iloc[cut]))
    fr.iloc[:cut].to_parquet(splits / 'train.parquet')
    fr.iloc[cut:].to_parquet(splits / 'val.parquet')
    builder.main(['validation'])
    out = otto_root / 'data' / 'covisitation' / 'validation'
    names = sorted(os.listdir(out))
    for kind in REFERENCE_KINDS:
        n15 = 1 if kind == 'cart_order' else 4
        n20 = 2 if kind == 'cart_order' else 6
        assert [n for n in names if n.startswith(f'top_15_{kind}_')] == [f'top_15_{kind}_{i}.pqt' for i in range(n15)]
        assert [n for n in names if n.startswith(f'top_{kind}_')] == [f'top_{kind}_{i}.pqt' for i in range(n20)]
    n_aids = int(ev.aid.max()) + 1
    for k, prefix, parts in ((15, 'top_15', 4), (20, 'top', 6)):
        want = co.covis_topk_numpy(ev.aid, ev.ts, ev.type, ev.sess_off, co.CovisSpec(kinds=('cart_weighted', 'click_order')), k=k)
        for kind in ('cart_weighted', 'click_order'):
            merged = {}
            for i in range(parts):
                merged.update(pd.read_parquet(out / f'{prefix}_{kind}_{i}.pqt').groupby('aid_x')['aid_y'].apply(list).to_dict())
            wx, wy, ww = want[kind]
            ref = pd.DataFrame({'aid_x': wx.astype(np.int64), 'aid_y': wy.astype(np.int64)}).groupby('aid_x')['aid_y'].apply(list).to_dict()
            assert merged == ref, (kind, k)
        df = pd.read_parquet(out / f'{prefix}_cart_weighted_0.pqt')
        assert df['wgt'].dtype == np.float32 and (df['wgt'] >= 1).all()
    assert n_aids <= 400


def test_bpr_training_improves_recall_at_20(gpu_device):
    """BPR hogwild epochs on a planted-structure dataset: recall@20 (src/metrics.py semantics) of the
    full-sort top-20 goes up and beats popularity-free chance by a wide margin."""
    from otto_amd.matrix_factorization.bpr import BPR, train_epoch
    from otto_amd import metrics
    rng = np.random.default_rng(0)
    n_users, n_items, groups = 4000, 2001, 20
    grp = rng.integers(0, groups, n_users)
    item_grp = np.r_[-1, rng.integers(0, groups, n_items - 1)]     # item 0 = PAD
    by = [np.flatnonzero(item_grp == g) for g in range(groups)]
    u = np.repeat(np.arange(n_users), 12)
    i = np.array([rng.choice(by[grp[x]]) for x in u])
    held = np.array([rng.choice(by[grp[x]]) for x in range(n_users)])
    torch.manual_seed(0)
    model = BPR(n_users, n_items, 32)
    with torch.no_grad():
        model.user_embedding.weight.normal_(0, 0.1)
        model.item_embedding.weight.normal_(0, 0.1)
    model.to(gpu_device)
    du, di = torch.from_numpy(u).to(gpu_device), torch.from_numpy(i).to(gpu_device)

    def recall():
        ids, _ = model.full_sort_topk(torch.arange(n_users, device=gpu_device), k=20, pad_col=0)
        ids = ids.cpu().numpy()
        assert (ids != 0).all()
        return metrics.recall_at_20(ids.tolist(), [[h] for h in held])
    r0 = recall()
    losses = [train_epoch(model, du, di, lr=0.2, seed=1, epoch=e, rows_per_launch=8192) for e in range(40)]
    r1 = recall()
    assert losses[-1] < 0.5 * losses[0]
    assert r1 > 5 * max(r0, 20 / n_items) and r1 > 0.12


def test_streamed_host_ingest_equals_device_feed(gpu_device):
    """Section 8 f2: pinned, double-buffered host -> device ingest (several chunks, odd sizes) gives the same matrices as
    one feed of device-resident arrays."""
    import numpy as np
    import torch
    from otto_amd.synth import generate_sessions
    from otto_amd.ingest import feed_host_events
    from otto_amd.covisitation.engine import CovisBuilder, topk_to_rows
    kinds = ('click_weighted', 'time_weighted', 'cart_order')
    ev = generate_sessions(5000, n_aids=800, seed=77)
    mk = lambda: CovisBuilder(ev.n_aids, kinds=kinds, ts_min=int(ev.ts.min()), ts_max=int(ev.ts.max()), device=gpu_device)
    a = mk()
    a.feed(torch.from_numpy(ev.aid.astype(np.int32)).to(gpu_device), torch.from_numpy(ev.ts).to(gpu_device),
           torch.from_numpy(ev.type).to(gpu_device), torch.from_numpy(ev.sess_off).to(gpu_device))
    want = a.finalize(k=20)
    b = mk()
    with torch.cuda.device(gpu_device):
        seconds, nbytes = feed_host_events(b, ev, gpu_device, chunk_sessions=777)
    assert nbytes >= 9 * ev.n_events
    got = b.finalize(k=20)
    for kind in kinds:
        for g, w in zip(topk_to_rows(*got[kind]), topk_to_rows(*want[kind])):
            assert np.array_equal(g, w), kind


def _dp_worker(rank, world, port, q, root, backend='gloo', rows_per_launch=4096, epochs=40):
    """One rank of the data-parallel MF test: the ranks share cuda:0 (the box has one GPU); gloo carries the collectives of
    the 2-rank run, nccl (= RCCL) those of the world-of-one run."""
    import os
    import sys
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path[:0] = [root, os.path.join(root, 'oracle')]
    import numpy as np
    import torch
    import torch.distributed as dist
    from otto_amd.matrix_factorization.bpr import BPR, ItemTableSync, train_epoch, full_sort_topk_sharded
    dist.init_process_group(backend, rank=rank, world_size=world)
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    u, i, held, n_users, n_items, d = _planted(128)
    torch.manual_seed(0)
    model = BPR(n_users, n_items, d)
    with torch.no_grad():
        model.user_embedding.weight.normal_(0, 0.1)
        model.item_embedding.weight.normal_(0, 0.1)
    model.to(dev)
    # session-chunk shard: users [lo, hi) and their rows belong to this rank
    # (unequal chunks: 17,004 and 18,996 rows at two ranks -- the ranks' launch counts differ, the exchange schedule must not)
    cuts = [0] + [n_users * r // world - 83 for r in range(1, world)] + [n_users]
    lo, hi = cuts[rank], cuts[rank + 1]
    mine = (u >= lo) & (u < hi)
    du, di = torch.from_numpy(u[mine]).to(dev), torch.from_numpy(i[mine]).to(dev)
    row0 = int(np.flatnonzero(mine)[0])
    sync = ItemTableSync(model.item_embedding.weight.data)
    losses = [train_epoch(model, du, di, lr=0.2, seed=1, epoch=e, rows_per_launch=rows_per_launch, row0=row0, sync=sync, sync_every=2)
              for e in range(epochs)]
    V = model.item_embedding.weight.data
    # user rows are rank-private: gather them so every rank can score all users
    U = model.user_embedding.weight.data
    mask = torch.zeros(n_users, 1, device=dev)
    mask[lo:hi] = 1
    Uall = U * mask
    dist.all_reduce(Uall)
    ids_u, sc_u = full_sort_topk_sharded(Uall, V, k=20, pad_col=0, shard='users')
    ids_i, sc_i = full_sort_topk_sharded(Uall, V, k=20, pad_col=0, shard='items')
    from otto_amd.matrix_factorization.engine import score_topk
    ids_1, sc_1 = score_topk(Uall, V, k=20, pad_col=0)
    q.put((rank, V.cpu().numpy(), losses, ids_1.cpu().numpy(), sc_1.cpu().numpy(), ids_u.cpu().numpy(), sc_u.cpu().numpy(),
           ids_i.cpu().numpy(), sc_i.cpu().numpy(), dict(sync.stats)))
    dist.barrier()
    dist.destroy_process_group()


def _planted(d, seed=0):
    rng = np.random.default_rng(seed)
    n_users, n_items, groups = 3000, 1501, 15
    grp = rng.integers(0, groups, n_users)
    item_grp = np.r_[-1, rng.integers(0, groups, n_items - 1)]     # item 0 = PAD
    by = [np.flatnonzero(item_grp == g_) for g_ in range(groups)]
    u = np.repeat(np.arange(n_users), 12)
    i = np.array([rng.choice(by[grp[x]]) for x in u])
    held = np.array([rng.choice(by[grp[x]]) for x in range(n_users)])
    return u, i, held, n_users, n_items, d


def test_data_parallel_bpr_two_ranks_and_sharded_scoring(gpu_device):
    """BASELINE config 5 rehearsed with 2 processes on the one GPU (gloo carries the collectives): BPR d = 128,
    sessions sharded by chunk (17,004 and 18,996 rows: the ranks' launch counts differ), item table exchanged by
    ItemTableSync (reduce = 'sum') every 2 launches of 4,096 rows, then full-sort scoring sharded by users and by items.
      * replicas of the item table are bit-identical after every epoch's drain;
      * the loss and recall@20 lie in a TWO-SIDED band whose ends are single-process runs of the same rows. On this toy
        problem the loss after 40 epochs is governed by how stale the table is that a step reads (0.007 ... 0.23 from the
        launch size alone, tools/diag_dp_bpr.py; measured here: sequential 0.009, data-parallel 0.069, one launch per
        epoch 0.2), so the ends are the two stalenesses that bracket any exchange schedule: LOWER end = nothing staler than
        one launch -- one process, one table, the launches of rank 0 and rank 1 of every slot one after the other (same
        launch size and negatives); UPPER end = every step of an epoch reads the table the epoch started with -- one
        launch per epoch, what `train_epoch` does by default. Data parallelism with summed deltas exchanges inside the
        epoch, so it must land between them (10 % slack), and its recall@20 between theirs (0.05 slack);
      * both sharded scorings return exactly the unsharded result."""
    import queue
    import socket
    import torch.multiprocessing as mp
    from conftest import ROOT
    import mf_oracle as mo
    from otto_amd.matrix_factorization.bpr import BPR
    from otto_amd.matrix_factorization.engine import BPR_HOGWILD
    u, i, held, n_users, n_items, d = _planted(128)
    W, rpl, sync_every = 2, 4096, 2
    cuts = [0] + [n_users * r // W - 83 for r in range(1, W)] + [n_users]
    shards = []
    for r in range(W):
        mine = (u >= cuts[r]) & (u < cuts[r + 1])
        shards.append((torch.from_numpy(u[mine]).to(gpu_device), torch.from_numpy(i[mine]).to(gpu_device), int(np.flatnonzero(mine)[0])))
    n_slots = max((sh[0].numel() + rpl - 1) // rpl for sh in shards)

    du_all, di_all = torch.from_numpy(u).to(gpu_device), torch.from_numpy(i).to(gpu_device)

    def single_process(slots_per_table):
        """1: the ranks' launches in slot order, one after the other; n_slots: the whole epoch (all rows, user order) in ONE launch"""
        torch.manual_seed(0)
        ref = BPR(n_users, n_items, d)
        with torch.no_grad():
            ref.user_embedding.weight.normal_(0, 0.1)
            ref.item_embedding.weight.normal_(0, 0.1)
        ref.to(gpu_device)
        eng = ref.engine(W * rpl * slots_per_table)
        U, V = ref.user_embedding.weight.data, ref.item_embedding.weight.data
        losses = []
        for e in range(40):
            acc = torch.zeros(1, device=gpu_device)
            for q0 in range(0, n_slots, slots_per_table):
                group = []
                for q in range(q0, min(n_slots, q0 + slots_per_table)):
                    for su, si, row0 in shards:
                        lo, hi = min(su.numel(), q * rpl), min(su.numel(), (q + 1) * rpl)
                        if hi > lo:
                            group.append((su[lo:hi], si[lo:hi], row0 + lo))
                if slots_per_table == 1:
                    for gu, gi, r0 in group:                       # one launch per (slot, rank), one after the other
                        eng.bpr_step(U, V, gu, gi, 1, e, r0, 0.2, 0.0, BPR_HOGWILD, loss_sum=acc)
                elif group:                                        # ONE launch per epoch: every step reads the table of the epoch's start
                    eng.bpr_step(U, V, du_all, di_all, 1, e, 0, 0.2, 0.0, BPR_HOGWILD, loss_sum=acc)
            losses.append(float(acc.item()) / len(u))
        ids, _ = ref.full_sort_topk(torch.arange(n_users, device=gpu_device), k=20, pad_col=0)
        rec = np.mean([mo.click_recall([h], row.tolist()) for h, row in zip(held, ids.cpu().numpy())])
        return losses, rec
    lo_losses, r_lo = single_process(1)
    hi_losses, r_hi = single_process(n_slots)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_dp_worker, args=(r, 2, port, q, ROOT)) for r in range(2)]
    for p in ps:
        p.start()
    res = {}
    for _ in range(300):
        try:
            r, *rest = q.get(timeout=1)
            res[r] = rest
            if len(res) == 2:
                break
        except queue.Empty:
            if not all(p.is_alive() for p in ps) and q.empty():
                break
    for p in ps:
        p.join(60)
    assert len(res) == 2 and all(p.exitcode == 0 for p in ps), 'a data-parallel worker failed'
    V0, l0, id1, sc1, idu, scu, idi, sci, stats = res[0]
    V1 = res[1][0]
    assert np.array_equal(V0, V1), 'item-table replicas differ after the drain'
    assert stats['dense'] + stats['sparse'] > 0
    n_rows = [sh[0].numel() for sh in shards]
    mean_loss = (np.array(l0) * n_rows[0] + np.array(res[1][1]) * n_rows[1]) / sum(n_rows)
    print(f'data-parallel loss {mean_loss[-1]:.4f}; single process: sequential launches {lo_losses[-1]:.4f}, one launch per epoch {hi_losses[-1]:.4f}')
    assert mean_loss[-1] < 0.5 * mean_loss[0]
    assert lo_losses[-1] < hi_losses[-1]
    assert 0.9 * lo_losses[-1] <= mean_loss[-1] <= 1.1 * hi_losses[-1], (lo_losses[-1], mean_loss[-1], hi_losses[-1])
    r_dp = np.mean([mo.click_recall([h], row.tolist()) for h, row in zip(held, id1)])
    print(f'recall@20 {r_dp:.4f}; single process {r_lo:.4f} / {r_hi:.4f}')
    assert r_hi > 0.1 and min(r_lo, r_hi) - 0.05 <= r_dp <= max(r_lo, r_hi) + 0.05, (r_lo, r_dp, r_hi)
    for got_i, got_s in ((idu, scu), (idi, sci)):
        assert np.array_equal(got_i, id1) and np.array_equal(got_s, sc1), 'sharded scoring differs from the unsharded call'
    assert np.array_equal(res[1][6], idi) and np.array_equal(res[1][4], idu)      # every rank holds the full result


def test_data_parallel_bpr_over_rccl_world_of_one(gpu_device):
    """The same data-parallel code over the real RCCL backend (world of one process, as the covisitation exchange is tested):
    ItemTableSync's asynchronous dense all-reduce on device tensors and, with launches small enough to be tracked, its
    sparse (row id, delta row) all-gather; both sharded scorings (all-reduce MAX of int32 / float32, all-gather + device
    merge). With one rank the exchange adds nothing, so training must land where the plain loop lands (hogwild: loosely)
    and the sharded scorings must equal the unsharded call exactly."""
    import queue
    import socket
    import torch.multiprocessing as mp
    from conftest import ROOT
    from otto_amd.matrix_factorization.bpr import BPR, train_epoch
    u, i, held, n_users, n_items, d = _planted(128)
    torch.manual_seed(0)
    ref = BPR(n_users, n_items, d)
    with torch.no_grad():
        ref.user_embedding.weight.normal_(0, 0.1)
        ref.item_embedding.weight.normal_(0, 0.1)
    ref.to(gpu_device)
    du, di = torch.from_numpy(u).to(gpu_device), torch.from_numpy(i).to(gpu_device)
    for rows, epochs, want_stat in ((4096, 6, 'dense'), (32, 2, 'sparse')):    # 2 launches of 32 rows touch < 1/8 of the items
        ref_losses = [train_epoch(ref, du, di, lr=0.2, seed=1, epoch=e, rows_per_launch=rows) for e in range(epochs)]
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
        ctx = mp.get_context('spawn')
        q = ctx.Queue()
        p = ctx.Process(target=_dp_worker, args=(0, 1, port, q, ROOT, 'nccl', rows, epochs))
        p.start()
        res = None
        for _ in range(300):
            try:
                res = q.get(timeout=1)
                break
            except queue.Empty:
                if not p.is_alive():
                    break
        p.join(60)
        assert res is not None and p.exitcode == 0, f'RCCL worker failed (exit code {p.exitcode})'
        _, V, losses, id1, sc1, idu, scu, idi, sci, stats = res
        assert stats[want_stat] > 0, stats
        assert np.isfinite(V).all() and losses[-1] < losses[0]
        if rows == 4096:          # same launches as the first reference run from the same initial tables
            assert losses[-1] == pytest.approx(ref_losses[-1], rel=0.1)
        for got_i, got_s in ((idu, scu), (idi, sci)):
            assert np.array_equal(got_i, id1) and np.array_equal(got_s, sc1), 'sharded scoring differs from the unsharded call'


@pytest.mark.parametrize('variant', ['ms-uint64-strings', 'seconds-codes', 'arrow-table', 'datetime'])
def test_device_frame_to_events_equals_numpy_path(gpu_device, variant):
    """Section 8 f2: the device ingest (type-string map, ms -> s, stable (session, ts) radix sort, CSR offsets; include/
    otto_events.h) against the NumPy path (`events.frame_to_events`: lexsort) on SHUFFLED frames in the raw schemas the
    reference holds (`dataset_writer_pickle.py:29-33, 57-60`: uint64 ms + type strings; splits: seconds + codes).
    Identical SoA, CSR, session ids and permutation; ties on (session, ts) keep their input order."""
    import pyarrow as pa
    from otto_amd.events import frame_to_events, frame_to_events_device
    from otto_amd.synth import generate_sessions
    ev = generate_sessions(6000, n_aids=3000, seed=12)
    fr = ev.to_frame()
    fr['session'] = fr['session'] * 7 + 11_000_000                   # sparse session ids in the test split's range
    rng = np.random.default_rng(4)
    fr['ts'] = (fr['ts'] // 40) * 40                                  # many equal timestamps inside a session: ties
    fr = fr.iloc[rng.permutation(len(fr))].reset_index(drop=True)
    if variant in ('ms-uint64-strings', 'arrow-table'):
        fr['ts'] = fr['ts'].to_numpy().astype(np.uint64) * np.uint64(1000) + rng.integers(0, 1000, len(fr)).astype(np.uint64)
        fr['type'] = np.array(['clicks', 'carts', 'orders'])[fr['type'].to_numpy()]
        fr['session'], fr['aid'] = fr['session'].astype(np.uint32), fr['aid'].astype(np.uint32)
    elif variant == 'datetime':
        fr['ts'] = pd.to_datetime(fr['ts'], unit='s')
    want, want_ids = frame_to_events(fr)
    src = pa.Table.from_pandas(fr, preserve_index=False) if variant == 'arrow-table' else fr
    if variant == 'arrow-table':
        # the builder's loader hands over a LIST of tables (train + val), their columns in several chunks: no host concat
        cut = len(fr) // 3
        src = [pa.concat_tables([src.slice(0, cut // 2), src.slice(cut // 2, cut - cut // 2)]), src.slice(cut)]
        assert src[0].column('aid').num_chunks == 2
    elif variant == 'ms-uint64-strings':
        src = [fr.iloc[:len(fr) // 2], fr.iloc[len(fr) // 2:]]           # two pandas frames (train.pkl + test.pkl)
    got = frame_to_events_device(src, device=gpu_device)
    host = got.to_host()
    assert got.n_sessions == want.n_sessions == 6000 and got.n_aids == want.n_aids
    assert np.array_equal(host.sess_off, want.sess_off) and np.array_equal(got.session_ids.cpu().numpy(), want_ids)
    assert np.array_equal(host.aid, want.aid) and np.array_equal(host.ts, want.ts) and np.array_equal(host.type, want.type)
    ts_s = fr['ts'].to_numpy()
    ts_s = ts_s.astype('datetime64[s]').astype(np.int64) if variant == 'datetime' else ts_s.astype(np.int64) // (1000 if 'ms' in variant or variant == 'arrow-table' else 1)
    order = np.lexsort((ts_s, fr['session'].to_numpy().astype(np.int64)))
    assert np.array_equal(got.order.cpu().numpy().view(np.uint32), order.astype(np.uint32))


def test_device_event_sort_of_millions_of_rows_is_the_stable_sort(gpu_device):
    """The radix sort at a size where every workgroup walks several tiles and the digit offsets cross many workgroups
    (6 M rows, sparse session ids, coarse timestamps = many ties): permutation, sorted columns, CSR and session ids equal
    to a stable sort of (session, seconds) done by torch on the same device."""
    import ctypes as C
    from otto_amd import _lib
    n, n_sess = 6_000_011, 400_000
    g = torch.Generator(device=gpu_device)
    g.manual_seed(9)
    sess = (torch.randint(0, n_sess, (n,), device=gpu_device, generator=g) * 13 + 5_000_000).to(torch.int32)
    ts_ms = (1_659_304_800 + torch.randint(0, 3000, (n,), device=gpu_device, generator=g) * 600).to(torch.int64) * 1000 + 17
    aid = torch.randint(0, 1_855_603, (n,), device=gpu_device, generator=g).to(torch.int32)
    typ = torch.randint(0, 3, (n,), device=gpu_device, generator=g).to(torch.uint8)
    lib = _lib.lib()
    ws_b = lib.otto_events_sort_workspace(n)
    ws = torch.empty(int(ws_b), dtype=torch.uint8, device=gpu_device)
    o_aid = torch.empty(n, dtype=torch.int32, device=gpu_device)
    o_ts = torch.empty(n, dtype=torch.int32, device=gpu_device)
    o_type = torch.empty(n, dtype=torch.uint8, device=gpu_device)
    o_order = torch.empty(n, dtype=torch.int32, device=gpu_device)
    o_off = torch.empty(n + 1, dtype=torch.int64, device=gpu_device)
    o_id = torch.empty(n, dtype=torch.int32, device=gpu_device)
    ns = C.c_int64()
    p = lambda t: C.c_void_p(t.data_ptr())
    with torch.cuda.device(gpu_device):
        _lib.check(lib.otto_events_sort(p(sess), p(ts_ms), p(aid), p(typ), n, 1000, p(o_aid), p(o_ts), p(o_type), p(o_order), p(o_off), p(o_id),
                                        C.byref(ns), p(ws), int(ws_b), C.c_void_p(torch.cuda.current_stream(gpu_device).cuda_stream)),
                   'otto_events_sort')
    sec = ts_ms // 1000
    key = sess.to(torch.int64) * (1 << 32) + sec
    want_order = torch.sort(key, stable=True).indices
    assert torch.equal(o_order.to(torch.int64), want_order)
    assert torch.equal(o_aid, aid[want_order]) and torch.equal(o_ts.to(torch.int64), sec[want_order]) and torch.equal(o_type, typ[want_order])
    ids, counts = torch.unique_consecutive(sess[want_order], return_counts=True)
    S = int(ns.value)
    assert S == ids.numel() and torch.equal(o_id[:S], ids)
    assert torch.equal(o_off[:S + 1], torch.cat([torch.zeros(1, dtype=torch.int64, device=gpu_device), torch.cumsum(counts, 0)]))


def test_device_ingest_rejects_unknown_type_strings_and_handles_empty(gpu_device):
    from otto_amd import _lib
    from otto_amd.events import frame_to_events_device
    fr = pd.DataFrame({'session': np.array([1, 1], dtype=np.uint32), 'aid': np.array([5, 6], dtype=np.uint32),
                       'ts': np.array([1_659_304_800_000, 1_659_304_801_000], dtype=np.uint64), 'type': ['clicks', 'wishlist']})
    with pytest.raises(_lib.OttoError, match='none of clicks'):
        frame_to_events_device(fr, device=gpu_device)
    with pytest.raises(ValueError, match='null'):                         # the kernel never sees Arrow's validity bitmap
        frame_to_events_device(fr.assign(type=['clicks', None]), device=gpu_device)
    empty = frame_to_events_device(fr.iloc[:0], device=gpu_device)
    assert empty.n_events == 0 and empty.n_sessions == 0


def test_device_aid_pair_builders_equal_the_oracle(gpu_device):
    """Section 8 a6: `build_aid_pairs_device` ('time': session self-join + time predicate + per-pair mean / max; 'diff':
    next-aid positives / shuffled-aid negatives, de-duplicated, positives win) against the CPU restatement of
    `torch_trainer.py:190-255` in `oracle/pairs_oracle.py` (pinned by a hand-computed fixture, tests/test_host_logic.py)
    on the same events -- same set of labelled pairs (rows sorted by (x1, x2) before comparing;
    the consumer shuffles them anyway). 'time' with the full sample (the reference's row sample is unseeded) and, for 'diff',
    the same shuffle keys on both sides."""
    from otto_amd.events import frame_to_events_device
    import pairs_oracle as po
    from otto_amd.matrix_factorization.data import build_aid_pairs_device
    from otto_amd.synth import generate_sessions
    ev = generate_sessions(1500, n_aids=300, seed=21)
    fr = ev.to_frame()
    dev_ev = frame_to_events_device(fr, device=gpu_device, n_aids=300)

    def rows(t):
        a = np.stack([np.asarray(c, dtype=np.int64) for c in t], 1)
        return a[np.lexsort((a[:, 1], a[:, 0]))]
    for agg in ('mean', 'max'):
        want = po.pairs_time(fr, hour_difference=1, target_aggregation=agg)
        got = build_aid_pairs_device(dev_ev, 'time', hour_difference=1, target_aggregation=agg, sample_frac=1.0)
        g = rows([c.cpu().numpy() for c in got])
        w = rows([want['x1'], want['x2'], want['target']])
        assert g.shape == w.shape and np.array_equal(g, w), agg
        assert 0 < g[:, 2].mean() < 1
    keys = np.random.default_rng(3).integers(0, 2 ** 31, ev.n_events, dtype=np.uint64)
    keys[::5] = keys[1::5][:len(keys[::5])]                       # equal keys: the permutation must stay stable
    want = po.pairs_diff(fr, shuffle_keys=keys)
    got = build_aid_pairs_device(dev_ev, 'diff', shuffle_keys=keys)
    g = rows([c.cpu().numpy() for c in got])
    w = rows([want['x1'], want['x2'], want['target']])
    assert g.shape == w.shape and np.array_equal(g, w)
    # a sampled 'time' build is a subset of the pairs of the full one and deterministic in the seed
    a = build_aid_pairs_device(dev_ev, 'time', sample_frac=0.3, seed=5)
    b = build_aid_pairs_device(dev_ev, 'time', sample_frac=0.3, seed=5)
    assert all(torch.equal(x, y) for x, y in zip(a, b)) and 0 < a[0].numel() < g.shape[0] * 10
    with pytest.raises(ValueError):
        build_aid_pairs_device(dev_ev, 'nope')
    # 'diff' again at a size where the sorts inside span many workgroups (1 M events, 2 M raw records)
    big = generate_sessions(60000, n_aids=20000, seed=22)
    bfr = big.to_frame()
    bdev = frame_to_events_device(bfr, device=gpu_device, n_aids=20000)
    bkeys = np.random.default_rng(4).integers(0, 2 ** 31, big.n_events, dtype=np.uint64)
    want = po.pairs_diff(bfr, shuffle_keys=bkeys)
    got = build_aid_pairs_device(bdev, 'diff', shuffle_keys=bkeys)
    g = rows([c.cpu().numpy() for c in got])
    w = rows([want['x1'], want['x2'], want['target']])
    assert g.shape == w.shape and np.array_equal(g, w)
