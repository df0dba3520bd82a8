"""CPU tests pinning the MF oracle: R-MF against the golden vectors produced by the reference's own
modules/train()/validate() (tests/golden/make_mf_golden.py); BPR against PyTorch-CPU autograd."""
import os

import numpy as np
import pytest
import torch

import mf_oracle as mo
from conftest import GOLDEN

RTOL = 1e-4   # BASELINE.json north_star: "within 1e-4 relative of the CPU path"


@pytest.fixture(scope='module')
def gold():
    return np.load(os.path.join(GOLDEN, 'mf_golden.npz'))


def _run_oracle(g, p, kind, shared):
    n1, n2, d, B, nb, ne, step_size = g[p + 'hyper'].tolist()
    lr0 = float(g[p + 'lr'])
    E1 = g[p + 'w1_0'].copy()
    E2 = E1 if shared else g[p + 'w2_0'].copy()
    m1, v1 = np.zeros_like(E1), np.zeros_like(E1)
    m2, v2 = (m1, v1) if shared else (np.zeros_like(E2), np.zeros_like(E2))
    losses, t, first = [], 0, None
    epoch_val = []
    for e in range(ne):
        for b in range(nb):
            lr = lr0 * 0.5 ** (t // step_size)      # StepLR(step_size, gamma=0.5) stepped per batch
            t += 1
            l, pred = mo.sparse_adam_step(E1, m1, v1, E2, m2, v2, g[p + 'i1'][b], g[p + 'i2'][b], g[p + 'target'][b],
                                          kind, lr, step=t, shared=shared)
            losses.append(l)
            if first is None:
                first = (pred, E1.copy(), m1.copy(), v1.copy(), E2.copy(), m2.copy(), v2.copy())
        ev = [mo.eval_batch(E1, E2, g[p + 'i1'][b], g[p + 'i2'][b], g[p + 'target'][b], kind) for b in range(nb)]
        epoch_val.append((np.mean([x[0] for x in ev]), np.concatenate([x[1] for x in ev])))
    return losses, first, (E1, m1, v1, E2, m2, v2), epoch_val


@pytest.mark.parametrize('p,kind,shared', [('mf_', 'MSELoss', False), ('cf_', 'BCEWithLogitsLoss', True)])
def test_rmf_oracle_matches_reference_golden(gold, p, kind, shared):
    g = gold
    losses, first, final, epoch_val = _run_oracle(g, p, kind, shared)
    n1, n2, d, B, nb, ne, step_size = g[p + 'hyper'].tolist()
    np.testing.assert_allclose(first[0], g[p + 'step0_pred'], rtol=RTOL, atol=1e-6)
    np.testing.assert_allclose(first[1], g[p + 'step_w1'], rtol=RTOL, atol=1e-6)
    np.testing.assert_allclose(first[2], g[p + 'step_m1'], rtol=RTOL, atol=1e-7)
    np.testing.assert_allclose(first[3], g[p + 'step_v1'], rtol=RTOL, atol=1e-9)
    if not shared:
        np.testing.assert_allclose(first[4], g[p + 'step_w2'], rtol=RTOL, atol=1e-6)
        np.testing.assert_allclose(first[5], g[p + 'step_m2'], rtol=RTOL, atol=1e-7)
    np.testing.assert_allclose(losses, g[p + 'step_loss'], rtol=RTOL)
    np.testing.assert_allclose(np.mean(np.reshape(losses, (ne, nb)), axis=1), g[p + 'epoch_train_loss'], rtol=RTOL)
    np.testing.assert_allclose(final[0], g[p + 'w1_T'], rtol=1e-3, atol=2e-5)
    np.testing.assert_allclose([v[0] for v in epoch_val], g[p + 'epoch_val_loss'], rtol=RTOL)
    tgt = np.concatenate([g[p + 'target'][b] for b in range(nb)]).astype(np.float64)
    if not shared:   # regression_scores: MAE, MSE (src/matrix_factorization/metrics.py:60-85)
        mae = [np.abs(v[1] - tgt).mean() for v in epoch_val]
        mse = [((v[1] - tgt) ** 2).mean() for v in epoch_val]
        np.testing.assert_allclose(mae, g[p + 'epoch_val_s0'], rtol=RTOL)
        np.testing.assert_allclose(mse, g[p + 'epoch_val_s1'], rtol=RTOL)
    else:            # classification_scores: accuracy at 0.5 on sigmoid (metrics.py:30-57)
        acc = [(((1 / (1 + np.exp(-v[1].astype(np.float64)))) >= 0.5) == (tgt == 1)).mean() for v in epoch_val]
        np.testing.assert_allclose(acc, g[p + 'epoch_val_s0'], atol=2.0 / len(tgt))


def test_lr_trace_is_steplr(gold):
    for p in ('mf_', 'cf_'):
        n1, n2, d, B, nb, ne, step_size = gold[p + 'hyper'].tolist()
        want = [float(gold[p + 'lr']) * 0.5 ** ((t + 1) // step_size) for t in range(nb * ne)]
        np.testing.assert_allclose(gold[p + 'lr_trace'], want, rtol=1e-12)


def test_bpr_batch_step_matches_autograd():
    rng = np.random.default_rng(0)
    nu, ni, d, B = 40, 30, 16, 200
    U = (rng.standard_normal((nu, d)) * 0.3).astype(np.float32)
    V = (rng.standard_normal((ni, d)) * 0.3).astype(np.float32)
    u = rng.integers(0, nu, B)
    i = rng.integers(0, ni, B)
    j = mo.bpr_negatives(7, 2, 1000, i, ni)
    assert (j != i).all() and j.min() >= 0 and j.max() < ni
    lr, l2 = 0.05, 0.01
    tu, tv = torch.tensor(U, dtype=torch.float64, requires_grad=True), torch.tensor(V, dtype=torch.float64, requires_grad=True)
    x = (tu[u] * (tv[i] - tv[j])).sum(1)
    loss = torch.nn.functional.softplus(-x).sum() + 0.5 * l2 * ((tu[u] ** 2).sum() + (tv[i] ** 2).sum() + (tv[j] ** 2).sum())
    loss.backward()
    wantU, wantV = U - lr * tu.grad.numpy(), V - lr * tv.grad.numpy()
    l = mo.bpr_step_batch(U, V, u, i, j, lr, l2)
    np.testing.assert_allclose(U, wantU, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(V, wantV, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(l, torch.nn.functional.softplus(-x).sum().item(), rtol=1e-9)


def test_bpr_sequential_equals_batch_on_unique_rows():
    rng = np.random.default_rng(1)
    nu, ni, d, B = 64, 200, 8, 32
    U = rng.standard_normal((nu, d)).astype(np.float32)
    V = rng.standard_normal((ni, d)).astype(np.float32)
    u = rng.permutation(nu)[:B]
    perm = rng.permutation(ni)
    i, j = perm[:B], perm[B:2 * B]
    U2, V2 = U.copy(), V.copy()
    a = mo.bpr_step_batch(U, V, u, i, j, 0.1)
    b = mo.bpr_step_sequential(U2, V2, u, i, j, 0.1)
    np.testing.assert_allclose(U, U2, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(V, V2, rtol=1e-6, atol=1e-7)
    assert abs(a - b) < 1e-9 * max(1, abs(a))


def test_negative_sampler_is_uniform_and_keyed():
    pos = np.zeros(20000, dtype=np.int64)
    j = mo.bpr_negatives(42, 0, 0, pos, 50)
    assert (j != 0).all()
    h = np.bincount(j, minlength=50)[1:]
    assert h.min() > 300 and h.max() < 520          # ~408 expected per item
    assert not np.array_equal(j, mo.bpr_negatives(42, 1, 0, pos, 50))
    assert np.array_equal(j[100:200], mo.bpr_negatives(42, 0, 100, pos[:100], 50))


def test_score_topk_pad_and_ties():
    U = np.array([[1.0, 0.0], [0.0, 1.0]], dtype=np.float32)
    V = np.array([[9, 9], [1, 0], [1, 5], [0.5, 5], [1, 2]], dtype=np.float32)
    ids, sc = mo.score_topk(U, V, k=3, pad_col=0)
    assert ids[0].tolist() == [1, 2, 4] and ids[1].tolist() == [2, 3, 4]     # ties -> smaller id first; PAD never
    assert sc[0].tolist() == [1, 1, 1]
