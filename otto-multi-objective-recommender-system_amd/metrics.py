"""recall@20, the acceptance metric of the candidate / scoring path.

Definitions follow the reference (``src/metrics.py:4-28`` clicks, ``:31-61`` carts / orders; aggregate
``src/covisitation/inference.py:251-257``: sum of hits / sum of min(|ground truth|, 20), weights 0.1 / 0.3 / 0.6) and
are pinned by ``tests/golden/metrics_golden.json`` (outputs of the reference functions). Written set-first: a
session's score is a hit count over a capped ground-truth size.
"""
import math

_CAP = 20
TYPE_WEIGHTS = {'clicks': 0.1, 'carts': 0.3, 'orders': 0.6}


def _hits_and_size(y_true, y_pred):
    truth = frozenset(y_true)
    return len(truth.intersection(y_pred)), len(truth)


def click_recall(y_true, y_pred):
    """1 / 0: is the (single) ground-truth click among the predictions; NaN without a ground truth."""
    for first in y_true:
        return 1 if first in frozenset(y_pred) else 0
    return math.nan


def cart_order_recall(y_true, y_pred):
    """|truth ∩ predictions| / min(20, |truth|) over DISTINCT ground-truth aids; NaN without a ground truth."""
    hits, size = _hits_and_size(y_true, y_pred)
    return hits / min(_CAP, size) if size else math.nan


def recall_at_20(predictions, labels):
    """Aggregate over sessions: sum |pred[:20] ∩ gt| / sum min(|gt|, 20) (duplicates in gt count once in the hits,
    as listed in the denominator -- the form ``src/covisitation/inference.py:251-257`` computes)."""
    hits = denom = 0
    for pred, gt in zip(predictions, labels):
        hits += len(frozenset(gt).intersection(pred[:_CAP]))
        denom += min(len(gt), _CAP)
    return hits / denom if denom else math.nan


def weighted_recall(click, cart, order):
    w = TYPE_WEIGHTS
    return w['clicks'] * click + w['carts'] * cart + w['orders'] * order
