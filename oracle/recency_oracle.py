"""CPU ORACLE for the recency-weighted candidates -- TEST INFRASTRUCTURE, NOT PRODUCT.

Restates, expression by expression, the per-session body of the reference's
``src/ranker/recency_weighted_candidate_generator.py:61-93`` (the first half of ``src/covisitation/inference.py:143-165``
is the same accumulation):

    session_unique_aids   = list(dict.fromkeys(session_aids[::-1]))                                   (:65)
    click_recency_weights = np.logspace(0.1, 1, len(session_aids), base=2, endpoint=True) - 1         (:68)
    cart_recency_weights  = np.logspace(0.5, 1, len(session_aids), base=2, endpoint=True) - 1         (:69, order: :70)
    Counter[aid] += recency_weight * event_type_coefficient[event_type]   in event order              (:75-78)
    [aid / weight for aid, weight in Counter.most_common(len(session_unique_aids))]                   (:81-93)

with ``event_type_coefficient = {0: 1, 1: 6, 2: 1}`` (:24). PARITY: the same NumPy / stdlib calls as the reference;
the reference holds no fixture for this loop (its input is the validation split, absent from the tree).
"""
from collections import Counter

import numpy as np

EVENT_TYPE_COEFFICIENT = {0: 1, 1: 6, 2: 1}
CURVES = ((0.1, 1.0), (0.5, 1.0))            # clicks; carts and orders (same curve in the reference)


def session_recency(session_aids, session_event_types, curves=CURVES, coef=EVENT_TYPE_COEFFICIENT):
    """-> [(sorted aids, sorted weights)] per curve."""
    session_aids = list(map(int, session_aids))
    session_event_types = list(map(int, session_event_types))
    session_unique_aids = list(dict.fromkeys(session_aids[::-1]))
    out = []
    for start, stop in curves:
        recency_weights = np.logspace(start, stop, len(session_aids), base=2, endpoint=True) - 1
        weights = Counter()
        for aid, event_type, recency_weight in zip(session_aids, session_event_types, recency_weights):
            weights[aid] += (recency_weight * coef[event_type])
        common = weights.most_common(len(session_unique_aids))
        out.append(([aid for aid, _ in common], [float(w) for _, w in common]))
    return out


def all_recency(aid, typ, sess_off, curves=CURVES, coef=EVENT_TYPE_COEFFICIENT):
    return [session_recency(aid[int(sess_off[s]):int(sess_off[s + 1])], typ[int(sess_off[s]):int(sess_off[s + 1])], curves, coef)
            for s in range(len(sess_off) - 1)]
