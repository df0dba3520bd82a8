"""CPU ORACLE for the covisitation builder -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product path (``otto_amd.covisitation``) never
does and fails loudly when the HIP library is missing.

PARITY UNPINNED: the reference ships no covisitation *builder* (SURVEY.md F1:
``src/covisitation/`` holds only the consumer ``inference.py``), so there is no
reference output, test or golden vector to pin this arithmetic against.  What is
pinned by the reference is the output *contract* (files ``top_15_<kind>_<i>.pqt``
with columns ``aid_x, aid_y`` in rank order: ``src/covisitation/inference.py:19-35,
87-111``; ``src/ranker/regular_candidate_generation.py:75-101``) and the literal
type-weight vectors (``src/baseline/aid_weight.py:34,82``,
``src/covisitation/inference.py:72``).  The arithmetic follows SURVEY.md App. A
(SPEC-COVIS), restated in DESIGN.md, whose only in-reference relative is the
session self-join idiom of ``src/matrix_factorization/torch_trainer.py:198-223``
(merge on session -> drop aid_x == aid_y -> time-gap predicate -> groupby
(aid_x, aid_y) aggregate).

Two independent restatements live here and are checked against each other:

* :func:`covis_pairs_python` -- pure-Python loops, a literal transcription of the
  spec, for small inputs;
* :func:`covis_pairs_numpy` -- vectorised NumPy, used at 100k-session scale and as
  the timed ``cpu_baseline``.

SPEC-COVIS (normative):
 1. events sorted by (session, ts), stable; ``ts`` int32 seconds; type in {0,1,2}.
 2. window: only the LAST ``W`` events of a session take part (tail window).
 3. ordered pairs (i, j) of window positions with aid_i != aid_j and
    |ts_i - ts_j| <= max_gap.
 4. per kind a 3x3 type mask M[type_i][type_j].
 5. dedupe: per (session, aid_x, aid_y, kind) only the FIRST valid pair in
    (i, j) lexicographic order contributes.
 6. weight (Q16 integer): time_weighted 65536 + (3*65536*(ts_i - t0)) // (t1 - t0)
    (t0, t1 = global min/max ts; 0 extra if t1 == t0); type-weighted kinds
    65536 * Wk[type_j]; filter kinds 65536.
 7. W[x, y] = sum over sessions (exact integer).
 8. top-k per aid_x by (W desc, aid_y asc).
 9. rows sorted (aid_x asc, rank asc); wgt = float32(W / 65536).
"""
from dataclasses import dataclass, field
import numpy as np

Q16 = 65536

# kind name -> (group, parameter).  group 'time': time-decay weight; 'type': Wk[type_y];
# 'filter': 3x3 mask M[type_x][type_y], unit weight.
TYPE_WEIGHTS = {
    'click_weighted': (1, 6, 3),   # src/baseline/aid_weight.py:34
    'cart_weighted': (1, 9, 6),    # src/covisitation/inference.py:72
    'order_weighted': (1, 3, 6),   # src/baseline/aid_weight.py:82
}
FILTER_MASKS = {
    # M[type_x][type_y]
    'click_cart': ((0, 1, 0), (0, 0, 0), (0, 0, 0)),
    'click_order': ((0, 0, 1), (0, 0, 0), (0, 0, 0)),
    'cart_order': ((0, 0, 0), (0, 1, 1), (0, 1, 1)),
    'click_click': ((1, 0, 0), (0, 0, 0), (0, 0, 0)),   # BASELINE.json config 1
}
ALL_KINDS = ('time_weighted', 'click_weighted', 'cart_weighted', 'order_weighted',
             'click_cart', 'click_order', 'cart_order', 'click_click')


@dataclass
class CovisSpec:
    window: int = 30
    max_gap: int = 86400
    kinds: tuple = ALL_KINDS
    ts_min: int = None   # t0 / t1 of step 6; default = min/max of the input
    ts_max: int = None


def _mask(kind):
    if kind in FILTER_MASKS:
        return np.array(FILTER_MASKS[kind], dtype=bool)
    return np.ones((3, 3), dtype=bool)


def _t01(ts, spec):
    t0 = int(ts.min()) if spec.ts_min is None else int(spec.ts_min)
    t1 = int(ts.max()) if spec.ts_max is None else int(spec.ts_max)
    return t0, t1


def _weight(kind, ts_x, type_y, t0, t1):
    """Q16 weight of one contribution (python ints)."""
    if kind == 'time_weighted':
        extra = (3 * Q16 * (int(ts_x) - t0)) // (t1 - t0) if t1 > t0 else 0
        return Q16 + extra
    if kind in TYPE_WEIGHTS:
        return Q16 * TYPE_WEIGHTS[kind][int(type_y)]
    return Q16


def covis_pairs_python(aid, ts, typ, sess_off, spec=CovisSpec()):
    """Literal transcription of SPEC-COVIS steps 2-7. Returns {kind: {(x, y): W}}."""
    t0, t1 = _t01(np.asarray(ts), spec) if len(ts) else (0, 0)
    out = {k: {} for k in spec.kinds}
    masks = {k: _mask(k) for k in spec.kinds}
    for s in range(len(sess_off) - 1):
        lo, hi = int(sess_off[s]), int(sess_off[s + 1])
        lo = max(lo, hi - spec.window)
        n = hi - lo
        for k in spec.kinds:
            seen = set()
            acc = out[k]
            M = masks[k]
            for i in range(n):
                ai, ti, yi = int(aid[lo + i]), int(ts[lo + i]), int(typ[lo + i])
                for j in range(n):
                    if i == j:
                        continue
                    aj, tj, yj = int(aid[lo + j]), int(ts[lo + j]), int(typ[lo + j])
                    if ai == aj or abs(ti - tj) > spec.max_gap or not M[yi][yj]:
                        continue
                    if (ai, aj) in seen:
                        continue
                    seen.add((ai, aj))
                    acc[(ai, aj)] = acc.get((ai, aj), 0) + _weight(k, ti, yj, t0, t1)
    return out


def covis_pairs_numpy(aid, ts, typ, sess_off, spec=CovisSpec(), chunk_elems=8_000_000, stats=None):
    """Vectorised SPEC-COVIS steps 2-7.

    Returns {kind: (x uint32[], y uint32[], W uint64[])} sorted by (x, y).
    ``stats`` (dict) receives ``P`` = number of deduped ordered pairs of the
    all-ones-mask expansion (the denominator of aid-pairs/sec) and ``tail_events``.
    """
    aid = np.asarray(aid).astype(np.int64)
    ts = np.asarray(ts).astype(np.int64)
    typ = np.asarray(typ).astype(np.int64)
    sess_off = np.asarray(sess_off).astype(np.int64)
    L = np.diff(sess_off)
    n_win = np.minimum(L, spec.window)
    w_start = sess_off[1:] - n_win
    t0, t1 = _t01(ts, spec) if len(ts) else (0, 0)
    masks = {k: _mask(k) for k in spec.kinds}
    parts = {k: ([], []) for k in spec.kinds}   # keys (x<<32|y), weights
    P = 0
    need_base = True

    for n in np.unique(n_win):
        n = int(n)
        if n < 2:
            continue
        starts = w_start[n_win == n]
        step = max(1, chunk_elems // (n * n))
        ii, jj = np.meshgrid(np.arange(n), np.arange(n), indexing='ij')
        for c in range(0, len(starts), step):
            st = starts[c:c + step]
            idx = st[:, None] + np.arange(n)[None, :]
            A, T, Y = aid[idx], ts[idx], typ[idx]                  # [m, n]
            m = A.shape[0]
            eq = A[:, :, None] == A[:, None, :]                     # [m, n, n]
            cls = np.argmax(eq, axis=2)                             # first position with the same aid
            base = (~eq) & (np.abs(T[:, :, None] - T[:, None, :]) <= spec.max_gap)
            ckey = (np.arange(m)[:, None, None] * 32 + cls[:, :, None]) * 32 + cls[:, None, :]
            if need_base:
                P += int(np.unique(ckey[base]).shape[0])
            for k in spec.kinds:
                v = base & masks[k][Y[:, :, None], Y[:, None, :]] if k in FILTER_MASKS else base
                flat = np.flatnonzero(v.ravel())                    # (m, i, j) lexicographic order
                if flat.size == 0:
                    continue
                _, first = np.unique(ckey.ravel()[flat], return_index=True)
                sel = flat[first]
                mm, rem = np.divmod(sel, n * n)
                i, j = np.divmod(rem, n)
                x, y = A[mm, i], A[mm, j]
                if k == 'time_weighted':
                    extra = (3 * Q16 * (T[mm, i] - t0)) // (t1 - t0) if t1 > t0 else 0
                    w = Q16 + extra
                elif k in TYPE_WEIGHTS:
                    w = Q16 * np.array(TYPE_WEIGHTS[k], dtype=np.int64)[Y[mm, j]]
                else:
                    w = np.full(sel.shape, Q16, dtype=np.int64)
                parts[k][0].append((x.astype(np.uint64) << np.uint64(32)) | y.astype(np.uint64))
                parts[k][1].append(np.asarray(w, dtype=np.uint64))
    out = {}
    for k in spec.kinds:
        if parts[k][0]:
            key = np.concatenate(parts[k][0])
            w = np.concatenate(parts[k][1])
            order = np.argsort(key, kind='stable')
            key, w = key[order], w[order]
            ukey, first = np.unique(key, return_index=True)
            W = np.add.reduceat(w, first).astype(np.uint64)
        else:
            ukey = np.zeros(0, dtype=np.uint64)
            W = np.zeros(0, dtype=np.uint64)
        out[k] = ((ukey >> np.uint64(32)).astype(np.uint32), (ukey & np.uint64(0xFFFFFFFF)).astype(np.uint32), W)
    if stats is not None:
        stats['P'] = P
        stats['tail_events'] = int(n_win.sum())
    return out


def pairs_dict_to_arrays(d):
    """{(x, y): W} -> (x, y, W) sorted by (x, y)."""
    if not d:
        z = np.zeros(0, dtype=np.uint32)
        return z, z.copy(), np.zeros(0, dtype=np.uint64)
    items = sorted(d.items())
    x = np.array([k[0] for k, _ in items], dtype=np.uint32)
    y = np.array([k[1] for k, _ in items], dtype=np.uint32)
    W = np.array([v for _, v in items], dtype=np.uint64)
    return x, y, W


def topk_rows(x, y, W, k=20):
    """SPEC-COVIS step 8-9: per aid_x keep k by (W desc, aid_y asc); rows (aid_x asc, rank asc).

    Returns (aid_x uint32[], aid_y uint32[], W uint64[]).
    """
    if len(x) == 0:
        return x, y, W
    order = np.lexsort((y, np.iinfo(np.uint64).max - W, x))   # x asc, W desc, y asc
    x, y, W = x[order], y[order], W[order]
    new = np.r_[True, x[1:] != x[:-1]]
    grp_start = np.flatnonzero(new)
    rank = np.arange(len(x)) - np.repeat(grp_start, np.diff(np.r_[grp_start, len(x)]))
    keep = rank < k
    return x[keep], y[keep], W[keep]


def covis_topk_numpy(aid, ts, typ, sess_off, spec=CovisSpec(), k=20, stats=None):
    pairs = covis_pairs_numpy(aid, ts, typ, sess_off, spec, stats=stats)
    return {kind: topk_rows(*pairs[kind], k=k) for kind in spec.kinds}


def wgt_float32(W):
    """SPEC-COVIS step 9 output weight."""
    return (np.asarray(W, dtype=np.float64) / Q16).astype(np.float32)


def expand_window_python(aid, ts, typ, lo, hi, spec, filter_kinds=(), t0=0, t1=0):
    """One session's deduped all-ones-mask pairs with the per-pair attributes the
    device records carry (used to check the pair-expand kernel on its own).

    Returns a list, ordered by (first window position of x, first window position
    of y), of tuples ``(x, y, type_y, fbits, extra)`` where ``type_y`` / ``extra``
    come from the FIRST valid (i, j) in lexicographic order (SPEC-COVIS 5), and bit f
    of ``fbits`` says a valid pair of (x, y) passes ``filter_kinds[f]``'s mask.
    """
    lo = max(lo, hi - spec.window)
    n = hi - lo
    first = {}
    fbits = {}
    pos = {}
    for i in range(n):
        pos.setdefault(int(aid[lo + i]), i)
    masks = [_mask(k) for k in filter_kinds]
    for i in range(n):
        ai, ti, yi = int(aid[lo + i]), int(ts[lo + i]), int(typ[lo + i])
        for j in range(n):
            aj, tj, yj = int(aid[lo + j]), int(ts[lo + j]), int(typ[lo + j])
            if i == j or ai == aj or abs(ti - tj) > spec.max_gap:
                continue
            key = (ai, aj)
            if key not in first:
                extra = (3 * Q16 * (ti - t0)) // (t1 - t0) if t1 > t0 else 0
                first[key] = (yj, extra)
            b = 0
            for f, M in enumerate(masks):
                if M[yi][yj]:
                    b |= 1 << f
            fbits[key] = fbits.get(key, 0) | b
    keys = sorted(first, key=lambda k: (pos[k[0]], pos[k[1]]))
    return [(k[0], k[1], first[k][0], fbits[k], first[k][1]) for k in keys]
