"""CPU ORACLE for the aid-pair dataset builders -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/`` may import this module. It restates the two branches of the reference's
``src/matrix_factorization/torch_trainer.py:190-260`` with pandas / NumPy; the product path is the device builder
(``csrc/otto_pairs.hip``, ``matrix_factorization/data.py:build_aid_pairs_device``), which is checked against this file.

PARITY: the 'time' branch is plain pandas in the reference and is restated expression by expression below; its two
unseeded draws (the 15 % row sample ``:200``) are inputs here. The 'diff' branch is polars 0.15 (not installable
offline): restated from the source text, PARITY UNPINNED. Pinned otherwise only by the hand-computed fixture in
``tests/test_host_logic.py`` (negative dt, dt over one day, a 0.5 tie of the mean, a tie on the shuffle key, positive over
negative).

ONE DELIBERATE DEPARTURE, also stated in ``include/otto_pairs.h``: the reference labels a pair by
``(ts_y - ts_x).dt.seconds / 3600`` (``:206``). ``.dt.seconds`` is the seconds COMPONENT of a timedelta (0 .. 86399):
it drops whole days (an event 25 h later counts as 1 h later: label 1) and turns a negative difference of -10 min
into 23 h 50 min. SURVEY.md App. E lists this as a reference defect; the builders here use the signed difference in
total seconds, ``0 < dt <= hour_difference * 3600``. The two rules agree except when |dt| >= 1 day.
"""
import numpy as np
import pandas as pd


def ts_seconds(ts):
    """int64 seconds from seconds, milliseconds (values above 1e11) or datetime64 columns."""
    v = np.asarray(ts)
    if np.issubdtype(v.dtype, np.datetime64):
        return v.astype('datetime64[s]').astype(np.int64)
    v = v.astype(np.int64)
    return v // 1000 if len(v) and int(v.max()) > 10 ** 11 else v


def pairs_time(df, hour_difference=1, target_aggregation='mean', chunk_size=30000, row_mask=None):
    """``torch_trainer.py:190-227``. ``row_mask`` (bool per row of the (session, ts)-sorted frame) stands for the
    reference's unseeded ``.sample(frac=0.15)``; None keeps every row. Chunking by ``chunk_size`` sessions only bounds
    the size of the self-join (a session never spans two chunks), exactly as in the reference."""
    df = df.assign(ts=ts_seconds(df['ts'])).sort_values(['session', 'ts'], kind='stable').reset_index(drop=True)
    if row_mask is not None:
        df = df[np.asarray(row_mask, dtype=bool)]
    ids = df['session'].unique()                                                             # :195
    pieces = []
    for i in range(0, len(ids), chunk_size):                                                 # :198
        first, last = ids[i], ids[min(len(ids) - 1, i + chunk_size - 1)]
        part = df[df['session'].between(first, last)]                                        # :200 (.loc[first:last] on the session index)
        j = part.merge(part, on='session')                                                   # :202
        j = j[j['aid_x'] != j['aid_y']]                                                      # :204
        dt = j['ts_y'].to_numpy(np.int64) - j['ts_x'].to_numpy(np.int64)                     # :206, signed total seconds (see header)
        lab = ((dt > 0) & (dt <= int(round(hour_difference * 3600)))).astype(np.int64)       # :207-211
        pieces.append(pd.DataFrame({'x1': j['aid_x'].to_numpy(), 'x2': j['aid_y'].to_numpy(), 'target': lab}))
    allp = pd.concat(pieces, ignore_index=True) if pieces else pd.DataFrame({'x1': [], 'x2': [], 'target': []})
    g = allp.groupby(['x1', 'x2'])['target']
    if target_aggregation == 'mean':                                                         # :217-220
        out = (g.mean() >= 0.5).astype(np.int64).reset_index()
    elif target_aggregation == 'max':                                                        # :221-223
        out = g.max().reset_index()
    else:
        raise ValueError('Invalid target aggregation')                                       # :225
    return out.astype('int64')


def pairs_diff(df, shuffle_keys):
    """``torch_trainer.py:229-255``. ``shuffle_keys``: one integer per row of the (session, ts)-sorted frame; the
    session's aids ordered by (key, event order) stand for the reference's unseeded ``pl.col('aid').shuffle()``."""
    df = df.assign(ts=ts_seconds(df['ts'])).sort_values(['session', 'ts'], kind='stable').reset_index(drop=True)
    s = df['session'].to_numpy()
    x1 = df['aid'].to_numpy().astype(np.int64)
    has_next = np.r_[s[1:] == s[:-1], False]                        # shift(-1) is null on a session's last row ...
    x2 = np.r_[x1[1:], -1]
    x3 = x1[np.lexsort((np.asarray(shuffle_keys), s))]              # the session's aids in shuffled order, row by row
    keep = has_next                                                 # ... and drop_nulls() removes that row (:236)
    neg = keep & (x2 != x3) & (x1 != x3)                            # :239-241 (the reference repeats x1 != x3)
    pos = keep & (x2 != x3) & (x1 != x2) & (x1 != x3)               # :246-248
    N = pd.DataFrame({'x1': x1[neg], 'x2': x3[neg], 'target': 0}).drop_duplicates(['x1', 'x2'])    # :242-244
    P = pd.DataFrame({'x1': x1[pos], 'x2': x2[pos], 'target': 1}).drop_duplicates(['x1', 'x2'])    # :249-250
    return pd.concat((P, N), ignore_index=True).drop_duplicates(['x1', 'x2'], keep='first').astype('int64')   # :252-254: positives first
