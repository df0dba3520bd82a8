"""Synthetic OTTO-shape event streams (SURVEY.md App. C / §8 d).

The laws are calibrated on the dataset statistics the reference's notebooks print
(``notebook/otto-multi-objective-recommender-system-eda.ipynb`` cell 5,
``eda/session_count_distribution.png``, ``eda/aid_count_distribution.png``):

* session length  L = clip(round(LogNormal(ln 6, 1.435)), 2, 500)  (mean 16.7 / median 6)
* aid popularity  LogNormal(median 20, sigma 1.88) normalised to a categorical
* within-session repeat probability 0.25 (copy the base draw of an earlier event)
* type categorical p = (0.8985, 0.0780, 0.0235)
* ts: session start uniform in [1659304800, 1661723999] s, gaps Exp(60 s),
  5 % chance of an extra U(1 h, 3 d) jump; int32 seconds

Output is the event frame of ``src/utilities/dataset_writer_pickle.py:57-60``
as SoA arrays + CSR session offsets, sorted by (session, ts).

``generate_sessions`` (NumPy, seeded) is the generator used by tests and golden
fixtures; ``generate_sessions_torch`` draws from the same laws on a torch device
(used by bench.py at full-OTTO size, where host generation would dominate).
"""
from dataclasses import dataclass
import math
import numpy as np

OTTO_N_AIDS = 1_855_603
OTTO_N_SESSIONS = 14_571_582
TS_LO, TS_HI = 1_659_304_800, 1_661_723_999
TYPE_P = (0.8985, 0.0780, 0.0235)


@dataclass
class Events:
    """SoA event stream. ``sess_off`` has n_sessions+1 entries (CSR)."""
    aid: np.ndarray        # uint32 [E]
    ts: np.ndarray         # int32  [E] seconds
    type: np.ndarray       # uint8  [E] 0 click, 1 cart, 2 order
    sess_off: np.ndarray   # int64  [S+1]
    n_aids: int

    @property
    def n_events(self):
        return int(self.aid.shape[0])

    @property
    def n_sessions(self):
        return int(self.sess_off.shape[0] - 1)

    def session_ids(self):
        return np.repeat(np.arange(self.n_sessions, dtype=np.int64), np.diff(self.sess_off))

    def to_frame(self):
        import pandas as pd
        return pd.DataFrame({
            'session': self.session_ids().astype(np.uint32),
            'aid': self.aid.astype(np.uint32),
            'ts': self.ts.astype(np.int64),
            'type': self.type.astype(np.uint8),
        })


def generate_sessions(n_sessions, n_aids=OTTO_N_AIDS, seed=42, repeat_p=0.25,
                      len_median=6.0, len_sigma=1.435, max_len=500):
    rng = np.random.default_rng(seed)
    L = np.clip(np.rint(rng.lognormal(math.log(len_median), len_sigma, n_sessions)), 2, max_len).astype(np.int64)
    sess_off = np.zeros(n_sessions + 1, dtype=np.int64)
    np.cumsum(L, out=sess_off[1:])
    E = int(sess_off[-1])
    sess = np.repeat(np.arange(n_sessions, dtype=np.int64), L)
    start = sess_off[:-1][sess]
    pos = np.arange(E, dtype=np.int64) - start

    pop = rng.lognormal(math.log(20.0), 1.88, n_aids)
    cdf = np.cumsum(pop)
    cdf /= cdf[-1]
    base = np.minimum(np.searchsorted(cdf, rng.random(E)), n_aids - 1).astype(np.uint32)
    rep = (rng.random(E) < repeat_p) & (pos > 0)
    src = start + np.floor(rng.random(E) * pos).astype(np.int64)
    aid = np.where(rep, base[np.minimum(src, E - 1)], base).astype(np.uint32)

    typ = np.searchsorted(np.cumsum(TYPE_P), rng.random(E)).clip(0, 2).astype(np.uint8)

    t_start = rng.integers(TS_LO, TS_HI + 1, n_sessions)
    gaps = rng.exponential(60.0, E)
    jump = rng.random(E) < 0.05
    gaps = gaps + jump * rng.uniform(3600.0, 259200.0, E)
    gaps[sess_off[:-1]] = 0.0
    c = np.cumsum(np.floor(gaps).astype(np.int64))
    ts = (t_start[sess] + (c - c[sess_off[:-1]][sess])).astype(np.int32)
    return Events(aid=aid, ts=ts, type=typ, sess_off=sess_off, n_aids=int(n_aids))


def generate_sessions_torch(n_sessions, n_aids=OTTO_N_AIDS, seed=42, device='cuda', repeat_p=0.25,
                            len_median=6.0, len_sigma=1.435, max_len=500, pop_seed=None):
    """Same laws as :func:`generate_sessions`, drawn with torch on ``device``.

    Returns a dict of device tensors ``aid (int32 bit pattern of uint32), ts int32,
    type uint8, sess_off int64`` plus ``n_aids``.

    ``pop_seed``: the aid popularity table is the one a call with ``seed=pop_seed`` (and the same ``n_sessions``)
    draws, whatever ``seed`` is -- the ranks of a weak-scaling run (seed = 42 + rank) then sample their sessions
    from ONE catalogue, as the shards of one OTTO stream would, instead of superposing W different rank laws.
    ``pop_seed == seed`` (or None) is the single-stream generator unchanged.
    """
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    f32 = dict(device=device, dtype=torch.float32, generator=g)
    z = torch.randn(n_sessions, **f32)
    L = torch.clamp(torch.round(torch.exp(math.log(len_median) + len_sigma * z)), 2, max_len).to(torch.int64)
    sess_off = torch.zeros(n_sessions + 1, device=device, dtype=torch.int64)
    torch.cumsum(L, 0, out=sess_off[1:])
    E = int(sess_off[-1].item())
    sess = torch.repeat_interleave(torch.arange(n_sessions, device=device), L, output_size=E)
    start = sess_off[:-1][sess]
    pos = torch.arange(E, device=device) - start

    pop = torch.exp(math.log(20.0) + 1.88 * torch.randn(n_aids, device=device, dtype=torch.float64, generator=g))
    if pop_seed is not None and pop_seed != seed:
        g2 = torch.Generator(device=device)                  # replay the draws a seed=pop_seed call makes before its table
        g2.manual_seed(pop_seed)
        torch.randn(n_sessions, device=device, dtype=torch.float32, generator=g2)
        pop = torch.exp(math.log(20.0) + 1.88 * torch.randn(n_aids, device=device, dtype=torch.float64, generator=g2))
    cdf = torch.cumsum(pop, 0)
    cdf = cdf / cdf[-1]
    u = torch.rand(E, device=device, dtype=torch.float64, generator=g)
    base = torch.searchsorted(cdf, u).clamp_(max=n_aids - 1)
    del u
    rep = (torch.rand(E, **f32) < repeat_p) & (pos > 0)
    src = start + torch.floor(torch.rand(E, device=device, dtype=torch.float64, generator=g) * pos).to(torch.int64)
    aid = torch.where(rep, base[src.clamp_(max=E - 1)], base).to(torch.int32)
    del base, rep, src

    tp = torch.tensor(TYPE_P, device=device, dtype=torch.float32).cumsum(0)
    typ = torch.searchsorted(tp, torch.rand(E, **f32)).clamp_(max=2).to(torch.uint8)

    t_start = torch.randint(TS_LO, TS_HI + 1, (n_sessions,), device=device, generator=g)
    gaps = -60.0 * torch.log1p(-torch.rand(E, **f32))
    jump = torch.rand(E, **f32) < 0.05
    gaps = gaps + jump * (3600.0 + (259200.0 - 3600.0) * torch.rand(E, **f32))
    gaps[sess_off[:-1]] = 0.0
    c = torch.cumsum(torch.floor(gaps).to(torch.int64), 0)
    ts = (t_start[sess] + (c - c[sess_off[:-1]][sess])).to(torch.int32)
    return {'aid': aid, 'ts': ts, 'type': typ, 'sess_off': sess_off, 'n_aids': int(n_aids)}
