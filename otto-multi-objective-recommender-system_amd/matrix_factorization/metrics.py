"""Validation scores of the MF trainer, same names and definitions as the reference's
``src/matrix_factorization/metrics.py`` (``round_probabilities`` ``:8-27``, ``classification_scores`` ``:30-57``,
``regression_scores`` ``:60-85``) -- computed where the predictions live.

The reference concatenates every prediction on the host and calls scikit-learn (``torch_trainer.py:144-158``: 223.6 M
floats per epoch at the config's size). Here mean absolute / squared error and accuracy are quotients of running sums
the eval kernel accumulates (``otto_mf_eval_sums``), and ROC-AUC is the rank statistic of a device sort: nothing but a
few scalars crosses PCIe. The array forms below accept NumPy arrays or torch tensors on any device; pinned by the
reference's own epoch scores in ``tests/golden/mf_golden.npz``.
"""
import numpy as np
import torch


def _tensor(x):
    return x if torch.is_tensor(x) else torch.as_tensor(np.asarray(x))


def round_probabilities(probabilities, threshold):
    """1 where probability >= threshold, else 0 (uint8), same container kind as the input."""
    if torch.is_tensor(probabilities):
        return (probabilities >= threshold).to(torch.uint8)
    return (np.asarray(probabilities) >= threshold).astype(np.uint8)


def roc_auc(y_true, y_score):
    """Area under the ROC curve = P(score of a positive > score of a negative) + 0.5 P(tie): the Mann-Whitney statistic
    with average ranks for ties (what ``sklearn.metrics.roc_auc_score`` returns for binary labels), from ONE sort on the
    device the scores live on."""
    t = _tensor(y_true).reshape(-1)
    s = _tensor(y_score).reshape(-1).to(torch.float64)
    pos = t.to(s.device) > 0.5
    n_pos = int(pos.sum())
    n_neg = pos.numel() - n_pos
    if n_pos == 0 or n_neg == 0:
        raise ValueError('ROC AUC needs both classes in y_true')
    order = torch.argsort(s)
    s_sorted, pos_sorted = s[order], pos[order]
    # tie groups: average 1-based rank of the group = (first + last + 1) / 2
    head = torch.ones_like(pos_sorted)
    head[1:] = s_sorted[1:] != s_sorted[:-1]
    gid = torch.cumsum(head.to(torch.int64), 0) - 1
    first = torch.nonzero(head).reshape(-1)
    last = torch.cat((first[1:], torch.tensor([pos.numel()], device=first.device))) - 1
    avg_rank = ((first + last).to(torch.float64) * 0.5 + 1.0)[gid]
    u = float(avg_rank[pos_sorted].sum()) - n_pos * (n_pos + 1) / 2.0
    return u / (float(n_pos) * float(n_neg))


def classification_scores(y_true, y_pred, threshold=0.5):
    """{'accuracy', 'roc_auc'} of probabilities ``y_pred`` against binary ``y_true``."""
    t, p = _tensor(y_true).reshape(-1), _tensor(y_pred).reshape(-1)
    t = t.to(p.device)
    hits = ((p >= threshold) == (t > 0.5)).sum()
    return {'accuracy': float(hits) / t.numel(), 'roc_auc': roc_auc(t, p)}


def regression_scores(y_true, y_pred):
    """{'mean_absolute_error', 'mean_squared_error'} (float64 accumulation)."""
    t, p = _tensor(y_true).reshape(-1).to(torch.float64), _tensor(y_pred).reshape(-1).to(torch.float64)
    e = p - t.to(p.device)
    return {'mean_absolute_error': float(e.abs().mean()), 'mean_squared_error': float((e * e).mean())}


def scores_from_sums(sums, classification, roc_auc_value=None):
    """Score dict from the eval kernel's running sums (sum |p - t|, sum (p - t)^2, hits, count)."""
    s_abs, s_sq, hits, n = sums
    if n <= 0:
        raise ValueError('no samples were scored')
    if classification:
        return {'accuracy': hits / n, 'roc_auc': roc_auc_value}
    return {'mean_absolute_error': s_abs / n, 'mean_squared_error': s_sq / n}
