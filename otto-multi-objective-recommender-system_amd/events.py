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


class DeviceEvents:
    """Sorted event stream resident on the device: ``aid`` int32 (bit pattern of uint32), ``ts`` int32 seconds, ``type``
    uint8, ``sess_off`` int64 CSR, ``session_ids`` int64 (the session of every CSR row), ``order`` (input row of every
    output row) -- what ``CovisBuilder.feed`` takes."""

    def __init__(self, aid, ts, typ, sess_off, session_ids, order, n_aids):
        self.aid, self.ts, self.type, self.sess_off, self.session_ids, self.order, self.n_aids = aid, ts, typ, sess_off, session_ids, order, n_aids

    @property
    def n_events(self):
        return int(self.aid.numel())

    @property
    def n_sessions(self):
        return int(self.sess_off.numel() - 1)

    def to_host(self):
        """The same stream as a host :class:`synth.Events` (tests, the CPU oracle)."""
        return Events(aid=self.aid.cpu().numpy().view(np.uint32), ts=self.ts.cpu().numpy(), type=self.type.cpu().numpy(),
                      sess_off=self.sess_off.cpu().numpy(), n_aids=self.n_aids)


def _to_device(arr, dtype, device, pinned=True):
    """Host column -> device tensor through a page-locked staging buffer (asynchronous copy on the current stream).
    ``arr``: one array or a list of arrays (the column chunks of several frames / Arrow record batches): the pieces are
    copied back to back into ONE staging buffer -- no host-side concatenation of the frames."""
    import torch
    parts = [np.asarray(a) for a in (arr if isinstance(arr, (list, tuple)) else [arr])]
    total = sum(len(a) for a in parts)
    tdtype = torch.from_numpy(np.empty(0, dtype=dtype)).dtype
    if total == 0:
        return torch.empty(0, dtype=tdtype, device=device)
    host = torch.empty(total, dtype=tdtype, pin_memory=pinned)
    view, o = host.numpy(), 0
    for a in parts:                                   # plain NumPy copies (with the dtype conversion) into the staging buffer
        view[o:o + len(a)] = a
        o += len(a)
    return host.to(device, non_blocking=True)


def _column_chunks(frames, name):
    """The NumPy arrays of column ``name`` of every frame (pandas) / every chunk of every table (pyarrow), in order, zero-copy
    where the column allows it."""
    out = []
    for f in frames:
        if hasattr(f, 'column_names') and not hasattr(f, 'iloc'):
            out += [c for c in f.column(name).chunks if len(c)]
        else:
            out.append(f[name])
    return out


def frame_to_events_device(frame, device='cuda:0', n_aids=None, ts_unit='auto'):
    """Device-side :func:`frame_to_events`: the columns of a pandas frame or a pyarrow table -- or of
    a LIST of them (train + validation / test: the reference concatenates the frames on the host first,
    ``src/ranker/aid_feature_engineering.py:21-36``; here every column chunk is copied straight into one page-locked
    staging buffer) -- cross PCIe once, then the type-string map, the ms -> s division, the stable (session, ts) radix
    sort and the CSR session offsets all run in HIP kernels (``include/otto_events.h``): no ``pd.concat``, no host lexsort
    of 223 M rows. Same semantics as the NumPy path on the concatenated frame: ties keep their input order; ``ts_unit``
    's', 'ms' or 'auto'."""
    import ctypes as C
    import torch
    from . import _lib
    dev = torch.device(device)
    if dev.type != 'cuda':
        raise _lib.OttoError('frame_to_events_device needs a ROCm device (the NumPy path is events.frame_to_events)')
    lib = _lib.lib()
    frames = list(frame) if isinstance(frame, (list, tuple)) else [frame]

    def host(c):                                          # one column chunk -> NumPy
        if hasattr(c, 'to_numpy') and not hasattr(c, 'iloc') and hasattr(c, 'type'):
            return c.to_numpy(zero_copy_only=False)
        return c.to_numpy() if hasattr(c, 'to_numpy') else np.asarray(c)
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None and t.numel() else C.c_void_p(0)
    stream = lambda: C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    with torch.cuda.device(dev):
        ts_parts = []
        for c in _column_chunks(frames, 'ts'):
            t = host(c)
            ts_parts.append(t.astype('datetime64[s]').astype(np.int64) if np.issubdtype(t.dtype, np.datetime64) else t)
        is_dt = any(np.issubdtype(host(c).dtype, np.datetime64) for c in _column_chunks(frames, 'ts')[:1])
        n = sum(len(t) for t in ts_parts)
        div = 1
        d_ts = _to_device(ts_parts, np.int64, dev)
        del ts_parts
        if not is_dt and n:
            if ts_unit == 'ms' or (ts_unit == 'auto' and int(d_ts.max()) > 10 ** 11):
                div = 1000
        d_sess = _to_device([host(c) for c in _column_chunks(frames, 'session')], np.uint32, dev)
        aid_parts = [host(c) for c in _column_chunks(frames, 'aid')]
        if n_aids is None:
            n_aids = max([int(a.max()) for a in aid_parts if len(a)], default=0) + 1
        d_aid = _to_device(aid_parts, np.uint32, dev)
        del aid_parts
        d_type = torch.empty(n, dtype=torch.uint8, device=dev)
        o = 0
        for c in _column_chunks(frames, 'type'):
            is_arrow = hasattr(c, 'type') and not hasattr(c, 'iloc')
            m = len(c)
            is_str = (is_arrow and str(c.type) in ('string', 'large_string')) or (not is_arrow and getattr(c, 'dtype', np.dtype('u1')).kind in 'OUS')
            if is_str and m:
                import pyarrow as pa
                arr = c if is_arrow else pa.array(c.to_numpy(), type=pa.string())
                if isinstance(arr, pa.ChunkedArray):
                    arr = arr.combine_chunks()
                if arr.null_count:
                    # the device kernel reads offsets + bytes only: a null would be decoded from whatever bytes its offsets span
                    raise ValueError(f'type column holds {arr.null_count} null(s): event types must be clicks / carts / orders')
                big = str(arr.type) == 'large_string'
                bufs = arr.buffers()                 # [validity, offsets, data]
                off = np.frombuffer(bufs[1], dtype=np.int64 if big else np.int32)[arr.offset:arr.offset + m + 1]
                data = np.frombuffer(bufs[2], dtype=np.uint8)
                d_off = _to_device(off, off.dtype, dev)
                d_bytes = _to_device(data, np.uint8, dev)
                _lib.check(lib.otto_events_type_from_strings(ptr(d_off), int(big), ptr(d_bytes), m, ptr(d_type[o:o + m]), stream()),
                           'otto_events_type_from_strings')
            elif m:
                d_type[o:o + m].copy_(_to_device(host(c), np.uint8, dev))
            o += m
        ws_bytes = lib.otto_events_sort_workspace(n)
        ws = torch.empty(max(int(ws_bytes), 8), dtype=torch.uint8, device=dev)
        o_aid = torch.empty(n, dtype=torch.int32, device=dev)
        o_ts = torch.empty(n, dtype=torch.int32, device=dev)
        o_type = torch.empty(n, dtype=torch.uint8, device=dev)
        o_order = torch.empty(n, dtype=torch.int32, device=dev)
        sess_off = torch.empty(n + 1, dtype=torch.int64, device=dev)
        sess_id = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        ns = C.c_int64()
        _lib.check(lib.otto_events_sort(ptr(d_sess), ptr(d_ts), ptr(d_aid), ptr(d_type), n, div, ptr(o_aid), ptr(o_ts), ptr(o_type),
                                        ptr(o_order), C.c_void_p(sess_off.data_ptr()), ptr(sess_id), C.byref(ns), C.c_void_p(ws.data_ptr()),
                                        int(ws_bytes), stream()), 'otto_events_sort')
        S = int(ns.value)
        return DeviceEvents(o_aid, o_ts, o_type, sess_off[:S + 1].clone(), sess_id[:S].to(torch.int64) & 0xFFFFFFFF, o_order, int(n_aids))
