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


# ---- the recency branch of the standalone model: src/covisitation/inference.py:143-199 --------------------------------
INFERENCE_EVENT_TYPE_COEFFICIENT = {0: 1, 1: 9, 2: 6}       # :72


def session_recency_predictions(session_aids, session_event_types, top_time_weighted, top_cart_weighted, top_cart_order,
                                neighbours_of, n_pred=20, coef=INFERENCE_EVENT_TYPE_COEFFICIENT):
    """One session of the `recency_weight` branch, expression by expression (:145-199). ``top_*``: {aid: [aid_y, ...]};
    ``neighbours_of``: {aid: [the 45 nearest neighbours]} standing in for the fastText / Annoy query of :166-167 (an aid
    without an entry has no neighbours). -> ((click aids, weights), (cart ...), (order ...))."""
    import itertools
    session_aids = list(map(int, session_aids))
    session_event_types = list(map(int, session_event_types))
    session_unique_click_aids = np.unique(np.array(session_aids)[np.array(session_event_types) == 0]).tolist()
    session_unique_click_and_cart_aids = np.unique(np.array(session_aids)[np.array(session_event_types) <= 1]).tolist()
    session_unique_cart_and_order_aids = np.unique(np.array(session_aids)[np.array(session_event_types) >= 1]).tolist()

    click_recency_weights = np.logspace(0.1, 1, len(session_aids), base=2, endpoint=True) - 1
    cart_recency_weights = np.logspace(0.5, 1, len(session_aids), base=2, endpoint=True) - 1
    order_recency_weights = np.logspace(0.5, 1, len(session_aids), base=2, endpoint=True) - 1
    session_aid_click_weights = Counter()
    session_aid_cart_weights = Counter()
    session_aid_order_weights = Counter()
    for aid, event_type, click_recency_weight, cart_recency_weight, order_recency_weight in zip(
            session_aids, session_event_types, click_recency_weights, cart_recency_weights, order_recency_weights):
        session_aid_click_weights[aid] += (click_recency_weight * coef[event_type])
        session_aid_cart_weights[aid] += (cart_recency_weight * coef[event_type])
        session_aid_order_weights[aid] += (order_recency_weight * coef[event_type])

    fasttext_similar_aids = list(neighbours_of.get(session_aids[-1], []))
    for aid in fasttext_similar_aids:
        session_aid_click_weights[aid] += 0.05
        session_aid_cart_weights[aid] += 0.05
        session_aid_order_weights[aid] += 0.15

    covisited_clicks_aids = list(itertools.chain(*[top_time_weighted[aid] for aid in session_unique_click_aids if aid in top_time_weighted]))
    for aid in covisited_clicks_aids:
        session_aid_click_weights[aid] += 0.05
    sorted_click = session_aid_click_weights.most_common(n_pred)

    cart_weighted_covisited_aids = list(itertools.chain(*[top_cart_weighted[aid] for aid in session_unique_click_and_cart_aids if aid in top_cart_weighted]))
    for aid in cart_weighted_covisited_aids:
        session_aid_cart_weights[aid] += 0.05
    sorted_cart = session_aid_cart_weights.most_common(n_pred)

    covisited_cart_and_order_aids = list(itertools.chain(*[top_cart_order[aid] for aid in session_unique_cart_and_order_aids if aid in top_cart_order]))
    for aid in covisited_cart_and_order_aids:
        session_aid_order_weights[aid] += 0.15
    sorted_order = session_aid_order_weights.most_common(n_pred)

    return tuple(([a for a, _ in c], [float(w) for _, w in c]) for c in (sorted_click, sorted_cart, sorted_order))
