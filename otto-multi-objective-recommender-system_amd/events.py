"""Event frame -> SoA + CSR session offsets (host plumbing before the kernels).

The reference's frames are ``session u32, aid u32, ts u64 ms (pickles) or s (splits), type u8``
(``src/utilities/dataset_writer_pickle.py:57-60``) and consumers sort by ``(session, ts)``
(``src/ranker/aid_feature_engineering.py:40``).  SPEC-COVIS step 1 needs int32 seconds and a
stable (session, ts) order.
"""
import numpy as np

from .synth import Events


def frame_to_events(df, n_aids=None, ts_unit='auto'):
    """pandas frame with columns session, aid, ts, type -> :class:`synth.Events`.

    ``ts_unit``: 's', 'ms' or 'auto' (values above 1e11 are milliseconds, as in train.pkl/test.pkl,
    which consumers divide by 1000: ``src/ranker/aid_feature_engineering.py:37``).
    """
    ts = df['ts'].to_numpy()
    if np.issubdtype(ts.dtype, np.datetime64):
        ts = ts.astype('datetime64[s]').astype(np.int64)
    else:
        ts = ts.astype(np.int64)
        if ts_unit == 'ms' or (ts_unit == 'auto' and len(ts) and ts.max() > 10 ** 11):
            ts = ts // 1000
    session = df['session'].to_numpy().astype(np.int64)
    order = np.lexsort((ts, session))           # stable: ties keep input order
    if not np.array_equal(order, np.arange(len(order))):
        session, ts = session[order], ts[order]
        aid = df['aid'].to_numpy()[order]
        typ = df['type'].to_numpy()[order]
    else:
        aid, typ = df['aid'].to_numpy(), df['type'].to_numpy()
    if typ.dtype.kind in 'OUS':                 # raw jsonl type strings (dataset_writer_pickle.py:29-33)
        typ = np.vectorize({'clicks': 0, 'carts': 1, 'orders': 2}.get)(typ)
    starts = np.flatnonzero(np.r_[True, session[1:] != session[:-1]]) if len(session) else np.zeros(0, dtype=np.int64)
    sess_off = np.r_[starts, len(session)].astype(np.int64)
    aid = aid.astype(np.uint32)
    if n_aids is None:
        n_aids = int(aid.max()) + 1 if len(aid) else 1
    return Events(aid=aid, ts=ts.astype(np.int32), type=typ.astype(np.uint8), sess_off=sess_off, n_aids=int(n_aids)), \
        (session[starts].astype(np.int64) if len(session) else np.zeros(0, dtype=np.int64))
