"""(session, candidate) interaction features on the device -- the computation of the reference's
``src/ranker/interaction_feature_engineering.py:56-113`` (a polars group-by / join chain over the candidate pickle) on the
dense candidate arrays ``covisitation.candidates.candidate_lookup`` returns, without the pickle round trip.

``interaction_features`` returns device tensors; ``to_frame`` lays them out as the reference's feature frame (same column
names and order, :115-125) for ``DATA/feature_engineering/{train,test}_<type>_interaction_features.pkl``.
"""
import ctypes as C

import numpy as np

from .. import _lib

ROW_COLUMNS = ('session_candidate_occurrence_count', 'session_candidate_cumcount_last', 'session_candidate_click_occurrence_count',
               'session_candidate_cart_occurrence_count', 'session_candidate_order_occurrence_count')
SESSION_COLUMNS = ('session_candidate_score_mean', 'session_candidate_score_std', 'session_candidate_score_min',
                   'session_candidate_score_max', 'session_candidate_occurrence_count_mean', 'session_candidate_occurrence_count_sum',
                   'session_candidate_occurrence_count_max', 'session_candidate_cumcount_last_mean',
                   'session_candidate_cumcount_last_sum', 'session_candidate_cumcount_last_max')
AID_COLUMNS = ('aid_candidate_score_mean', 'aid_candidate_score_std', 'aid_candidate_score_max',
               'aid_session_candidate_occurrence_count_mean', 'aid_session_candidate_occurrence_count_sum',
               'aid_session_candidate_occurrence_count_max', 'aid_session_candidate_cumcount_last_mean',
               'aid_session_candidate_cumcount_last_sum', 'aid_session_candidate_cumcount_last_max')


def interaction_features(aid, typ, sess_off, cand, scores, n_aids):
    """``aid`` int32 / ``typ`` uint8 / ``sess_off`` int64: the sorted events of S sessions; ``cand`` int32 [S, C] (-1 padded,
    unique per row), ``scores`` float32 [S, C]. Returns (row uint16 [S, C, 5], session float32 [S, 10], aid float32 [n_aids, 9])
    in the column orders above; NaN = polars' null."""
    import torch
    dev = aid.device
    if dev.type != 'cuda':
        raise _lib.OttoError('interaction_features needs a ROCm device (no CPU fallback)')
    for name, x, dt in (('aid', aid, torch.int32), ('type', typ, torch.uint8), ('sess_off', sess_off, torch.int64),
                        ('cand', cand, torch.int32), ('scores', scores, torch.float32)):
        if x.dtype != dt or not x.is_contiguous():
            raise ValueError(f'{name}: expected contiguous {dt}')
    S, Cn = cand.shape
    if scores.shape != cand.shape or sess_off.numel() != S + 1:
        raise ValueError('cand / scores / sess_off shapes disagree')
    lib = _lib.lib()
    ws_b = lib.otto_inter_workspace(int(n_aids))
    ws = torch.empty(int(ws_b), dtype=torch.uint8, device=dev)
    row = torch.empty((S, Cn, 5), dtype=torch.int16, device=dev)
    sf = torch.empty((S, 10), dtype=torch.float32, device=dev)
    af = torch.empty((int(n_aids), 9), dtype=torch.float32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    with torch.cuda.device(dev):
        _lib.check(lib.otto_inter_features(p(aid), p(typ), p(sess_off), S, p(cand), p(scores), int(Cn), int(n_aids), p(row), p(sf), p(af),
                                           p(ws), int(ws_b), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                   'otto_inter_features')
    return row, sf, af


def interaction_features_rows(aid, typ, sess_off, table, n_aids):
    """The same features over the ranker's candidate table (``covisitation.candidates.ranker_table``: the frame
    ``interaction_feature_engineering.py:25-28`` reads -- the session's own aids followed by the candidates, one row per
    (session, candidate), CSR by session). Returns (row uint16 [R, 5], session float32 [S, 10], aid float32 [n_aids, 9])."""
    import torch
    dev = aid.device
    if dev.type != 'cuda':
        raise _lib.OttoError('interaction_features_rows needs a ROCm device (no CPU fallback)')
    cand, scores, row_off = table['candidates'], table['candidate_scores'], table['row_off']
    S, R = sess_off.numel() - 1, cand.numel()
    for name, x, dt in (('aid', aid, torch.int32), ('type', typ, torch.uint8), ('sess_off', sess_off, torch.int64),
                        ('candidates', cand, torch.int32), ('candidate_scores', scores, torch.float32), ('row_off', row_off, torch.int64)):
        if x.dtype != dt or not x.is_contiguous():
            raise ValueError(f'{name}: expected contiguous {dt}')
    if row_off.numel() != S + 1 or scores.numel() != R:
        raise ValueError('table / sess_off shapes disagree')
    lib = _lib.lib()
    ws_b = lib.otto_inter_workspace(int(n_aids))
    ws = torch.empty(int(ws_b), dtype=torch.uint8, device=dev)
    row = torch.empty((max(R, 1), 5), dtype=torch.int16, device=dev)
    sf = torch.empty((S, 10), dtype=torch.float32, device=dev)
    af = torch.empty((int(n_aids), 9), dtype=torch.float32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr()) if t.numel() else C.c_void_p(0)
    with torch.cuda.device(dev):
        _lib.check(lib.otto_inter_features_rows(p(aid), p(typ), p(sess_off), S, p(row_off), p(cand), p(scores), int(n_aids), C.c_void_p(row.data_ptr()),
                                                p(sf), p(af), p(ws), int(ws_b), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                   'otto_inter_features_rows')
    return row[:R], sf, af


def table_to_frame(table, row, sess_feat, aid_feat):
    """The reference's feature frame from the ranker table + :func:`interaction_features_rows` (same columns as :func:`to_frame`)."""
    import pandas as pd
    import torch
    off = table['row_off'].cpu().numpy()
    s_idx = np.repeat(np.arange(len(off) - 1), np.diff(off))
    cand_h = table['candidates'].cpu().numpy()
    out = {'session': table['session'].cpu().numpy(), 'candidates': cand_h, 'candidate_scores': table['candidate_scores'].cpu().numpy()}
    if table.get('candidate_labels') is not None:
        out['candidate_labels'] = table['candidate_labels'].cpu().numpy()
    r = row.cpu().numpy().view(np.uint16)
    for q, name in enumerate(ROW_COLUMNS):
        col = r[:, q].astype(np.float32) if name.endswith('cumcount_last') else r[:, q]
        if name.endswith('cumcount_last'):
            col[col == 0] = np.nan
        out[name] = col
    sf = sess_feat.cpu().numpy()[s_idx]
    for q, name in enumerate(SESSION_COLUMNS):
        out[name] = sf[:, q]
    af = aid_feat.cpu().numpy()[cand_h]
    for q, name in enumerate(AID_COLUMNS):
        out[name] = af[:, q]
    return pd.DataFrame(out)


def to_frame(session_ids, cand, scores, row, sess_feat, aid_feat, labels=None):
    """The reference's feature frame (one row per (session, candidate)): columns session, candidates, candidate_scores
    [, candidate_labels], the five row features, the ten session features, the nine aid features."""
    import pandas as pd
    cand_h = cand.cpu().numpy()
    keep = cand_h >= 0
    s_idx, c_idx = np.nonzero(keep)
    out = {'session': np.asarray(session_ids)[s_idx].astype(np.int32), 'candidates': cand_h[keep].astype(np.int32),
           'candidate_scores': scores.cpu().numpy()[keep].astype(np.float32)}
    if labels is not None:
        out['candidate_labels'] = np.asarray(labels.cpu().numpy() if hasattr(labels, 'cpu') else labels)[keep].astype(np.uint8)
    r = row.cpu().numpy().view(np.uint16)[s_idx, c_idx]
    for q, name in enumerate(ROW_COLUMNS):
        col = r[:, q].astype(np.float32) if name.endswith('cumcount_last') else r[:, q]
        if name.endswith('cumcount_last'):
            col[col == 0] = np.nan                       # absent from the session: null in the reference
        out[name] = col
    sf = sess_feat.cpu().numpy()[s_idx]
    for q, name in enumerate(SESSION_COLUMNS):
        out[name] = sf[:, q]
    af = aid_feat.cpu().numpy()[cand_h[keep]]
    for q, name in enumerate(AID_COLUMNS):
        out[name] = af[:, q]
    return pd.DataFrame(out)
