"""recall@20 with the definitions of the reference's ``src/metrics.py`` (``click_recall`` ``:4-28``,
``cart_order_recall`` ``:31-61``) and the aggregate every validation script logs
(``src/covisitation/inference.py:251-257``: sum(hits) / sum(min(len(gt), 20)), weighted 0.1/0.3/0.6)."""
import numpy as np


def click_recall(y_true, y_pred):
    if len(y_true) == 0:
        recall = np.nan
    else:
        recall = int(y_true[0] in y_pred)
    return recall


def cart_order_recall(y_true, y_pred):
    y_true = set(y_true)
    y_pred = set(y_pred)
    tp = len(y_true.intersection(y_pred))
    fn = len(y_true - y_pred)
    try:
        recall = tp / min(20, (tp + fn))
    except ZeroDivisionError:
        recall = np.nan
    return recall


def recall_at_20(predictions, labels):
    """Aggregate recall over sessions: sum |pred[:20] & gt| / sum min(|gt|, 20)."""
    hits = sum(len(set(p[:20]).intersection(set(g))) for p, g in zip(predictions, labels))
    denom = sum(min(len(g), 20) for g in labels)
    return hits / denom if denom else np.nan


def weighted_recall(click, cart, order):
    return 0.1 * click + 0.3 * cart + 0.6 * order
