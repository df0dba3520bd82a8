"""Batch loader and dataset builders of the MF trainer.

* :class:`DeviceBatchLoader` replaces the Merlin GPU loader the reference uses
  (``torch_trainer.py:315-318``: ``Loader(dataset, batch_size, shuffle=True, drop_last=False)``):
  the whole parquet sits in HBM (223.6M rows x 3 x int64 = 5.4 GB of 288 GB) and every epoch
  yields ``(dict_of_int64_device_tensors, None)`` batches in a fresh random order.
* ``build_sessions_aids`` / ``build_aid_pairs_device`` are the dataset builders of
  ``torch_trainer.py:190-260, 286-287`` (App. E defects not reproduced); the aid-pair builders run on the device
  (``csrc/otto_pairs.hip``), their CPU restatement lives in ``oracle/pairs_oracle.py`` (test infrastructure).
"""
import numpy as np
import torch


class DeviceBatchLoader:

    def __init__(self, columns, batch_size, shuffle=True, drop_last=False, device='cuda:0', seed=None):
        self.device = torch.device(device)
        self.columns = {k: (v if torch.is_tensor(v) else torch.from_numpy(np.array(v, dtype=np.int64))).to(
            device=self.device, dtype=torch.int64).contiguous() for k, v in columns.items()}
        self.n = next(iter(self.columns.values())).numel()
        self.batch_size, self.shuffle, self.drop_last = int(batch_size), shuffle, drop_last
        self.gen = torch.Generator(device=self.device)
        if seed is not None:
            self.gen.manual_seed(seed)

    @classmethod
    def from_parquet(cls, path, batch_size, shuffle=True, drop_last=False, device='cuda:0', seed=None):
        import pyarrow.parquet as pq
        table = pq.read_table(str(path))
        return cls({name: table.column(name).to_numpy() for name in table.column_names if not name.startswith('__')},
                   batch_size, shuffle, drop_last, device, seed)

    def check_ranges(self, limits):
        """``limits``: column -> table size. Raises ValueError when a column holds an id outside [0, size): the YAML's
        ``n_sessions`` / ``n_aids`` / ``n_embeddings`` do not cover the parquet (the kernels would skip such rows and the
        epoch would end in ``OttoError``; this names the column up front)."""
        for name, size in limits.items():
            col = self.columns.get(name)
            if col is None or col.numel() == 0:
                continue
            lo, hi = int(col.min()), int(col.max())
            if lo < 0 or hi >= int(size):
                raise ValueError(f"column '{name}' holds ids in [{lo}, {hi}] but its embedding table has {int(size)} rows")

    def __len__(self):
        return self.n // self.batch_size if self.drop_last else (self.n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        perm = torch.randperm(self.n, device=self.device, generator=self.gen) if self.shuffle else None
        for b in range(len(self)):
            lo, hi = b * self.batch_size, min(self.n, (b + 1) * self.batch_size)
            if perm is None:
                yield {k: v[lo:hi] for k, v in self.columns.items()}, None
            else:
                idx = perm[lo:hi]
                yield {k: v[idx] for k, v in self.columns.items()}, None


def _ts_seconds(ts):
    """Event times as int64 SECONDS. ``train.pkl`` / ``test.pkl`` carry uint64 milliseconds
    (``src/utilities/dataset_writer_pickle.py:59``), the split parquets seconds, the reference's 'time' branch expects a
    datetime column (``torch_trainer.py:206`` ``.dt.seconds``): all three are accepted, with the same 'auto' rule as
    ``events.frame_to_events`` (values above 1e11 are milliseconds). Signed, so ``ts_y - ts_x`` cannot wrap."""
    v = np.asarray(ts)
    if np.issubdtype(v.dtype, np.datetime64):
        return v.astype('datetime64[s]').astype(np.int64)
    v = v.astype(np.int64)
    if len(v) and int(v.max()) > 10 ** 11:
        v = v // 1000
    return v


def build_sessions_aids(df):
    """``torch_trainer.py:286-287``: (session, aid, target = type), int64."""
    out = df.rename(columns={'type': 'target'})[['session', 'aid', 'target']].astype('int64')
    return out


def build_aid_pairs_device(ev, sampling_strategy='diff', hour_difference=1, target_aggregation='mean', sample_frac=0.15, seed=42,
                           shuffle_keys=None):
    """Labelled aid pairs (x1, x2, target) of ``torch_trainer.py:190-260`` (``include/otto_pairs.h``)
    over a sorted, resident event stream ``ev`` (:class:`otto_amd.events.DeviceEvents`). Returns device int64 tensors
    ``(x1, x2, target)`` sorted by (x1, x2) -- the same SET of labelled pairs as the CPU restatement
    ``oracle/pairs_oracle.py`` (test infrastructure) for the same sample / the same ``shuffle_keys``.

    'diff' (``:229-255``): per session x1 = aid, x2 = next aid (positive), x3 = a random aid of the same session
    (negative); (x1, x2) target 1 and (x1, x3) target 0 with x2 != x3, x1 != x2 / x1 != x3; de-duplicated, positives win
    over negatives. 'time' (``:190-227``): self-join of a row sample per session, drop aid_x == aid_y, target =
    0 < dt <= hour_difference hours in signed total seconds (the reference's ``.dt.seconds`` drops whole days:
    ``include/otto_pairs.h``), aggregated per pair by mean >= 0.5 or max. The reference's sampling / shuffle is
    unseeded; here it is seeded. There is no host path: the builders need a ROCm device.

    'time': the row sample (``sample_frac`` of the events, Bernoulli per event from a seeded device generator; the reference
    samples each 30,000-session chunk unseeded) is drawn here, the self-join, the time predicate and the per-pair mean / max
    run in ``otto_pairs_time``. 'diff': the per-session permutation is a device radix sort of (session, random key)
    (``otto_events_sort``), pairs and de-duplication in ``otto_pairs_diff``. One call handles up to 2^32 - 1 raw records
    (full OTTO at the reference's 15 % sample: ~4.6e8)."""
    import ctypes as C
    from .. import _lib
    dev = ev.aid.device
    if dev.type != 'cuda':
        raise _lib.OttoError('build_aid_pairs_device needs a ROCm device (no CPU fallback)')
    if sampling_strategy not in ('time', 'diff'):
        raise ValueError('Invalid sampling strategy')
    if target_aggregation not in ('mean', 'max'):
        raise ValueError('Invalid target aggregation')
    lib = _lib.lib()
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None and t.numel() else C.c_void_p(0)
    stream = lambda: C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed))
    aid, ts, off = ev.aid, ev.ts, ev.sess_off
    E, S = aid.numel(), off.numel() - 1
    with torch.cuda.device(dev):
        if sampling_strategy == 'time':
            if sample_frac < 1.0:
                keep = torch.rand(E, device=dev, generator=gen) < sample_frac
                sess = torch.repeat_interleave(torch.arange(S, device=dev), off[1:] - off[:-1], output_size=E)
                cnt = torch.bincount(sess[keep], minlength=S)
                off = torch.zeros(S + 1, dtype=torch.int64, device=dev)
                torch.cumsum(cnt, 0, out=off[1:])
                aid, ts = aid[keep].contiguous(), ts[keep].contiguous()
            second = None
        else:
            # x3: the session's aids in a random order = stable sort of (session, random key)
            keys = (torch.randint(0, 2 ** 31, (E,), device=dev, generator=gen, dtype=torch.int64) if shuffle_keys is None
                    else torch.as_tensor(np.asarray(shuffle_keys, dtype=np.int64), device=dev))
            sess = torch.repeat_interleave(torch.arange(S, device=dev, dtype=torch.int32), off[1:] - off[:-1], output_size=E)
            ws_b = lib.otto_events_sort_workspace(E)
            ws = torch.empty(max(int(ws_b), 8), dtype=torch.uint8, device=dev)
            second = torch.empty(E, dtype=torch.int32, device=dev)
            tmp_ts = torch.empty(E, dtype=torch.int32, device=dev)
            tmp_ty = torch.empty(E, dtype=torch.uint8, device=dev)
            tmp_off = torch.empty(E + 1, dtype=torch.int64, device=dev)
            tmp_id = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
            ns = C.c_int64()
            _lib.check(lib.otto_events_sort(ptr(sess), ptr(keys), ptr(aid), ptr(ev.type), E, 1, ptr(second), ptr(tmp_ts), ptr(tmp_ty),
                                            C.c_void_p(0), C.c_void_p(tmp_off.data_ptr()), ptr(tmp_id), C.byref(ns),
                                            C.c_void_p(ws.data_ptr()), int(ws_b), stream()), 'otto_events_sort')
            del ws, tmp_ts, tmp_ty, tmp_off, tmp_id
        raw = C.c_int64()
        _lib.check(lib.otto_pairs_raw_count(ptr(off), S, int(sampling_strategy == 'time'), C.byref(raw), stream()), 'otto_pairs_raw_count')
        raw = int(raw.value)
        if raw >= 2 ** 32:
            raise ValueError(f'{raw} raw pair records exceed one call (2^32 - 1): lower sample_frac or split the sessions')
        ws_b = lib.otto_pairs_workspace(raw)
        ws = torch.empty(max(int(ws_b), 8), dtype=torch.uint8, device=dev)
        x1 = torch.empty(max(raw, 1), dtype=torch.int64, device=dev)
        x2, tg = torch.empty_like(x1), torch.empty_like(x1)
        n_rows = C.c_int64()
        if sampling_strategy == 'time':
            _lib.check(lib.otto_pairs_time(ptr(aid), ptr(ts), ptr(off), S, raw, int(round(float(hour_difference) * 3600)),
                                           0 if target_aggregation == 'mean' else 1, ptr(x1), ptr(x2), ptr(tg), C.byref(n_rows),
                                           C.c_void_p(ws.data_ptr()), int(ws_b), stream()), 'otto_pairs_time')
        else:
            _lib.check(lib.otto_pairs_diff(ptr(aid), ptr(second), ptr(off), S, raw, ptr(x1), ptr(x2), ptr(tg), C.byref(n_rows),
                                           C.c_void_p(ws.data_ptr()), int(ws_b), stream()), 'otto_pairs_diff')
        n = int(n_rows.value)
        return x1[:n].clone(), x2[:n].clone(), tg[:n].clone()
