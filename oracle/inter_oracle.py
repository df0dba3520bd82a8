"""CPU ORACLE for the interaction feature engineering -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/`` and ``__graft_entry__.smoke()`` may import this module.

PARITY UNPINNED: the reference's ``src/ranker/interaction_feature_engineering.py:56-113`` is written in polars 0.15, which
is not installable here, and the reference holds no fixture of its output; this is a pandas restatement of that code,
statement by statement (line numbers in the comments). Polars semantics assumed: ``std`` is the sample standard deviation
(ddof = 1, null for a single value), aggregations skip nulls, ``sum`` of an all-null group is 0... the feature
``session_candidate_cumcount_last`` is null for candidates the session never saw.
"""
import numpy as np
import pandas as pd

ROW_FEATURES = ('session_candidate_occurrence_count', 'session_candidate_cumcount_last',
                'session_candidate_click_occurrence_count', 'session_candidate_cart_occurrence_count',
                'session_candidate_order_occurrence_count')
SESSION_FEATURES = ('session_candidate_score_mean', 'session_candidate_score_std', 'session_candidate_score_min',
                    'session_candidate_score_max', 'session_candidate_occurrence_count_mean',
                    'session_candidate_occurrence_count_sum', 'session_candidate_occurrence_count_max',
                    'session_candidate_cumcount_last_mean', 'session_candidate_cumcount_last_sum',
                    'session_candidate_cumcount_last_max')
AID_FEATURES = ('aid_candidate_score_mean', 'aid_candidate_score_std', 'aid_candidate_score_max',
                'aid_session_candidate_occurrence_count_mean', 'aid_session_candidate_occurrence_count_sum',
                'aid_session_candidate_occurrence_count_max', 'aid_session_candidate_cumcount_last_mean',
                'aid_session_candidate_cumcount_last_sum', 'aid_session_candidate_cumcount_last_max')


def interaction_features(df_candidate, df):
    """``df_candidate``: session, candidates, candidate_scores; ``df``: session, aid, ts, type (events of those sessions)."""
    cand = df_candidate.drop_duplicates().sort_values('session', kind='stable').reset_index(drop=True)             # :33-34
    df = df[df['session'].isin(cand['session'])].sort_values(['session', 'ts'], kind='stable').reset_index(drop=True)   # :53-54
    df['session_aid_cumcount'] = df.groupby('session').cumcount() + 1                                               # :57-61
    g = df.groupby(['session', 'aid'])
    per_aid = pd.DataFrame({'session_candidate_cumcount_last': g['session_aid_cumcount'].last(),                     # :63-65
                            'session_candidate_occurrence_count': g['aid'].count()}).reset_index()                   # :67
    per_type = df.groupby(['session', 'aid', 'type'])['aid'].count().rename('n').reset_index()                       # :68
    cand = cand.merge(per_aid.rename(columns={'aid': 'candidates'}), on=['session', 'candidates'], how='left')       # :71-76
    cand['session_candidate_occurrence_count'] = cand['session_candidate_occurrence_count'].fillna(0)                # :77
    for value, name in enumerate(('click', 'cart', 'order')):                                                       # :79-85
        col = f'session_candidate_{name}_occurrence_count'
        t = per_type[per_type['type'] == value].rename(columns={'aid': 'candidates', 'n': col})[['session', 'candidates', col]]
        cand = cand.merge(t, on=['session', 'candidates'], how='left')
        cand[col] = cand[col].fillna(0)
    cand = cand.drop_duplicates().reset_index(drop=True)

    def agg(key, prefix, with_min):
        gb = cand.groupby(key)
        out = {}
        sc = 'candidate_scores'
        out[f'{prefix}candidate_score_mean'] = gb[sc].mean()
        out[f'{prefix}candidate_score_std'] = gb[sc].std(ddof=1)
        if with_min:
            out[f'{prefix}candidate_score_min'] = gb[sc].min()
        out[f'{prefix}candidate_score_max'] = gb[sc].max()
        mid = 'session_candidate_' if prefix == 'session_' else 'aid_session_candidate_'
        for c in ('occurrence_count', 'cumcount_last'):
            src = f'session_candidate_{c}'
            out[f'{mid}{c}_mean'] = gb[src].mean()
            out[f'{mid}{c}_sum'] = gb[src].sum(min_count=0)
            out[f'{mid}{c}_max'] = gb[src].max()
        return pd.DataFrame(out).reset_index()
    cand = cand.merge(agg('session', 'session_', True), on='session', how='left')                                     # :87-100
    cand = cand.merge(agg('candidates', 'aid_', False), on='candidates', how='left')                                  # :102-113
    return cand
