"""CPU ORACLE for the matrix-factorization path -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module.

Pinning:
* R-MF (forward, MSE / BCE-with-logits loss, SparseAdam step): PINNED by
  ``tests/golden/mf_golden.npz``, which ``tests/golden/make_mf_golden.py`` produced by
  running the reference's own ``torch_modules.py`` + ``train()`` / ``validate()``
  (``/root/reference/src/matrix_factorization``) with torch's SparseAdam/StepLR on CPU in
  the build container.  Follows ``torch_modules.py:13-19,32-38`` (forward),
  ``torch_trainer.py:59-78`` (step order), ``torch/optim/_functional.py`` ``sparse_adam``.
* BPR step, negative sampler and full-sort top-k: PARITY UNPINNED.  The reference reaches
  BPR only through un-vendored recbole==1.1.1 (``requirements.txt:106``;
  ``src/recbole/trainer.py:28-40``, ``src/recbole/inference.py:58-61,76-80``); the
  arithmetic here is SURVEY.md App. B.2.  ``bpr_step_batch`` is cross-checked against
  PyTorch-CPU autograd in ``tests/test_mf_oracle.py``.
"""
import numpy as np

F = np.float32
MASK = (1 << 64) - 1


def forward(E1, E2, i1, i2):
    """out[b] = sum_f E1[i1[b], f] * E2[i2[b], f]  (torch_modules.py:15-17, 34-36)."""
    return (E1[i1].astype(F) * E2[i2].astype(F)).sum(axis=-1, dtype=F)


def loss_and_grad(kind, out, target):
    """Per-sample loss and dloss/dout (reduction='mean' applied by the caller)."""
    out = out.astype(np.float64)
    t = target.astype(np.float64)
    if kind == 'MSELoss':
        e = out - t
        return e * e, 2.0 * e
    if kind == 'BCEWithLogitsLoss':
        l = np.maximum(out, 0) - out * t + np.log1p(np.exp(-np.abs(out)))
        s = 1.0 / (1.0 + np.exp(-out))
        return l, s - t
    raise ValueError(kind)


def _coalesced(n, d, idx, rows):
    g = np.zeros((n, d), dtype=np.float64)
    np.add.at(g, idx, rows)
    touched = np.unique(idx)
    return touched, g[touched].astype(F)


def _adam_rows(E, m, v, touched, g, lr, betas, eps, step):
    """torch/optim/_functional.py sparse_adam on the coalesced rows (float32 like torch)."""
    b1, b2 = betas
    m_old, v_old = m[touched], v[touched]
    mu = (g - m_old) * F(1 - b1)
    m[touched] = m_old + mu
    vu = (g * g - v_old) * F(1 - b2)
    v[touched] = v_old + vu
    numer = mu + m_old
    denom = np.sqrt(vu + v_old) + F(eps)
    step_size = lr * np.sqrt(1 - b2 ** step) / (1 - b1 ** step)
    E[touched] = E[touched] + F(-step_size) * (numer / denom)


def sparse_adam_step(E1, m1, v1, E2, m2, v2, i1, i2, target, kind, lr, betas=(0.9, 0.999), eps=1e-8, step=1,
                     shared=False, batch_size=None):
    """One train() batch (torch_trainer.py:59-78). Tables updated in place; returns (mean loss, pred).
    ``batch_size``: divisor of reduction='mean' when it differs from len(i1) (a batch some of whose samples were
    skipped by the device kernels' range check)."""
    B = len(i1) if batch_size is None else int(batch_size)
    out = forward(E1, E2, i1, i2)
    l, g = loss_and_grad(kind, out, target)
    c = (g / B)[:, None]
    g1 = c * E2[i2].astype(np.float64)
    g2 = c * E1[i1].astype(np.float64)
    if shared:
        touched, gr = _coalesced(E1.shape[0], E1.shape[1], np.concatenate([i1, i2]), np.concatenate([g1, g2]))
        _adam_rows(E1, m1, v1, touched, gr, lr, betas, eps, step)
    else:
        t1, gr1 = _coalesced(E1.shape[0], E1.shape[1], i1, g1)
        t2, gr2 = _coalesced(E2.shape[0], E2.shape[1], i2, g2)
        _adam_rows(E1, m1, v1, t1, gr1, lr, betas, eps, step)
        _adam_rows(E2, m2, v2, t2, gr2, lr, betas, eps, step)
    return float(l.mean()), out


def eval_batch(E1, E2, i1, i2, target, kind):
    out = forward(E1, E2, i1, i2)
    l, _ = loss_and_grad(kind, out, target)
    return float(l.mean()), out


# ---------------------------------------------------------------------------
# BPR
# ---------------------------------------------------------------------------
def mix64(z):
    z = (z + 0x9E3779B97F4A7C15) & MASK
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
    return z ^ (z >> 31)


def bpr_negative(seed, epoch, row, pos, n_items):
    """Counter-based negative for one global row (python ints): uniform by multiply-high of 32
    random bits, redrawn (16 attempts) while equal to the positive."""
    base = mix64(seed ^ ((epoch * 0xD1342543DE82EF95) & MASK)) ^ ((row * 0xA0761D6478BD642F) & MASK)
    for att in range(16):
        r = mix64(base ^ ((att * 0xE7037ED1A0B428DB) & MASK))
        j = ((r >> 32) * n_items) >> 32
        if j != pos:
            return j
    return (pos + 1) % n_items


def bpr_negatives(seed, epoch, row0, pos, n_items):
    return np.array([bpr_negative(seed, epoch, row0 + b, int(p), n_items) for b, p in enumerate(pos)], dtype=np.int64)


def bpr_step_batch(U, V, u, i, j, lr, l2=0.0):
    """Deterministic batch step: every gradient from the pre-step tables, duplicates summed, then
    applied. Tables updated in place (float32); returns the SUM of softplus(-x)."""
    Uu, Vi, Vj = U[u].astype(np.float64), V[i].astype(np.float64), V[j].astype(np.float64)
    x = (Uu * (Vi - Vj)).sum(axis=1)
    s = 1.0 / (1.0 + np.exp(x))
    loss = np.maximum(-x, 0) + np.log1p(np.exp(-np.abs(x)))
    gU = np.zeros(U.shape, dtype=np.float64)
    gV = np.zeros(V.shape, dtype=np.float64)
    np.add.at(gU, u, s[:, None] * (Vi - Vj) - l2 * Uu)
    np.add.at(gV, i, s[:, None] * Uu - l2 * Vi)
    np.add.at(gV, j, -s[:, None] * Uu - l2 * Vj)
    U += (lr * gU).astype(F)
    V += (lr * gV).astype(F)
    return float(loss.sum())


def bpr_step_sequential(U, V, u, i, j, lr, l2=0.0):
    """Triplet-by-triplet SGD (what hogwild degenerates to without races)."""
    loss = 0.0
    for a, b, c in zip(u, i, j):
        eu, ei, ej = U[a].astype(np.float64), V[b].astype(np.float64), V[c].astype(np.float64)
        x = float((eu * (ei - ej)).sum())
        s = 1.0 / (1.0 + np.exp(x))
        loss += max(-x, 0) + np.log1p(np.exp(-abs(x)))
        U[a] = (eu + lr * (s * (ei - ej) - l2 * eu)).astype(F)
        V[b] = (ei + lr * (s * eu - l2 * ei)).astype(F)
        V[c] = (ej + lr * (-s * eu - l2 * ej)).astype(F)
    return loss


def score_topk(U, V, k=20, pad_col=-1):
    """recbole/inference.py:76-80: scores = U @ V.T, PAD column -inf, top-k (score desc, id asc)."""
    S = U.astype(np.float64) @ V.astype(np.float64).T
    if pad_col >= 0:
        S[:, pad_col] = -np.inf
    ids = np.empty((U.shape[0], k), dtype=np.int32)
    sc = np.empty((U.shape[0], k), dtype=np.float64)
    for b in range(U.shape[0]):
        order = np.lexsort((np.arange(S.shape[1]), -S[b]))[:k]
        ids[b], sc[b] = order, S[b, order]
    return ids, sc


# recall@20 acceptance formula (src/metrics.py:4-61; vectorised form src/covisitation/inference.py:251-257)
def click_recall(y_true, y_pred):
    return np.nan if len(y_true) == 0 else int(y_true[0] in y_pred)


def cart_order_recall(y_true, y_pred):
    y_true, y_pred = set(y_true), set(y_pred)
    tp = len(y_true & y_pred)
    fn = len(y_true - y_pred)
    return tp / min(20, tp + fn) if tp + fn else np.nan
