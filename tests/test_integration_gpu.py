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
