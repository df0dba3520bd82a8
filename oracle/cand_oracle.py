"""CPU ORACLE for the covisitation candidate lookup -- TEST INFRASTRUCTURE, NOT PRODUCT.

Restates, expression by expression, the per-session body of the reference's
``src/ranker/covisitation_candidate_generation.py:108-157`` (the same loop, plus fastText neighbours, is
``src/ranker/regular_candidate_generation.py:138-197`` and ``src/covisitation/inference.py:204-247``):

    session_unique_aids               = list(dict.fromkeys(session_aids[::-1]))                       (:112)
    session_unique_click_and_cart_aids = np.unique(aids[types <= 1]).tolist()                         (:116)
    session_unique_cart_and_order_aids = np.unique(aids[types >= 1]).tolist()                         (:117)
    <kind>_covisited_aids = list(itertools.chain(*[top_<kind>[aid] for aid in <source> if aid in top_<kind>]))   (:119-124)
    covisited = concatenation of the recipe's lists                                                    (:127, :133, :138)
    [(aid, count) for aid, count in Counter(covisited).most_common(100) if aid not in session_unique_aids]  (:128)

``Counter.most_common`` is Python's own: ties keep first-insertion order -- the property the builder's rank order
feeds.  PARITY: pinned by construction to the reference's expressions (the same stdlib calls); the
reference holds no fixture for this loop (its inputs, the covisitation parquets, do not exist in the tree).
"""
from collections import Counter
import itertools

import numpy as np

# recipes of covisitation_candidate_generation.py:127,133,138: (matrix kind, source list)
CLICK_RECIPE = (('time_weighted', 'U'), ('click_weighted', 'CC'), ('cart_weighted', 'CC'), ('click_cart', 'CC'), ('cart_order', 'CC'))
CART_RECIPE = (('time_weighted', 'U'), ('cart_weighted', 'CC'), ('cart_order', 'CC'))
ORDER_RECIPE = CART_RECIPE
# recipes of the standalone model, src/covisitation/inference.py:227,231,235 -- 'neighbours' = the fastText / Annoy
# neighbours of the session's LAST aid (:223-224), here one more {aid: [45 neighbours]} dict
INFERENCE_CLICK_RECIPE = CLICK_RECIPE + (('neighbours', 'LAST'),)
INFERENCE_CART_RECIPE = CART_RECIPE + (('neighbours', 'LAST'),)


def matrix_to_dict(y, n):
    """Dense top-k arrays -> the dict the consumers build with covisitation_df_to_dict (covisitation/inference.py:19-35)."""
    return {int(x): y[x, :n[x]].tolist() for x in np.flatnonzero(n > 0)}


def session_candidates(session_aids, session_event_types, top, recipe, n_common=100):
    session_aids = list(map(int, session_aids))
    aids = np.array(session_aids)
    types = np.array(session_event_types)
    session_unique_aids = list(dict.fromkeys(session_aids[::-1]))
    sources = {
        'U': session_unique_aids,
        'CC': np.unique(aids[types <= 1]).tolist(),
        'CO': np.unique(aids[types >= 1]).tolist(),
        'LAST': session_aids[-1:],
    }
    covisited = []
    for kind, src in recipe:
        d = top[kind]
        covisited += list(itertools.chain(*[d[aid] for aid in sources[src] if aid in d]))
    out = [(aid, count) for aid, count in Counter(covisited).most_common(n_common) if aid not in session_unique_aids]
    return [a for a, _ in out], [c for _, c in out]


def all_candidates(aid, typ, sess_off, top, recipe, n_common=100):
    res = []
    for s in range(len(sess_off) - 1):
        lo, hi = int(sess_off[s]), int(sess_off[s + 1])
        res.append(session_candidates(aid[lo:hi], typ[lo:hi], top, recipe, n_common))
    return res


def session_predictions(session_aids, sorted_aids, most_frequent, n_pred=20):
    """src/covisitation/inference.py:236-237 (and :238-241 for carts / orders)."""
    session_unique_aids = list(dict.fromkeys(list(map(int, session_aids))[::-1]))
    predictions = session_unique_aids + sorted_aids[:n_pred - len(session_unique_aids)]
    predictions = predictions + most_frequent[:n_pred - len(predictions)]
    return predictions


def session_ranker_rows(session_aids, sorted_aids, sorted_counts, labels=None):
    """``src/ranker/regular_candidate_generation.py:160-193`` for one session and one event type: ``sorted_aids`` /
    ``sorted_counts`` = ``[(aid, weight) for aid, weight in Counter(...).most_common(100) if aid not in session_unique_aids]``
    (:160); returns (predictions, scores, labels)."""
    session_unique_aids = list(dict.fromkeys(list(map(int, session_aids))[::-1]))
    scores = np.arange(1, len(session_unique_aids) + 1).tolist()[::-1] + list(sorted_counts)          # :162
    predictions = session_unique_aids + list(sorted_aids)                                               # :178
    lab = None if labels is None else [int(aid in labels) for aid in predictions]                       # :191-193
    return predictions, scores, lab

