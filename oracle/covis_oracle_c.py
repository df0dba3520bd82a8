"""ctypes wrapper of oracle/covis_oracle.c (TEST INFRASTRUCTURE; see that file's header)."""
import ctypes as C
import os

import numpy as np

import covis_oracle as co

_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_build', 'libcovis_oracle.so')
_lib = None


def available():
    return os.path.exists(_LIB)


def _load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_LIB)
        _lib.covis_topk_c.restype = C.c_int64
    return _lib


def covis_topk_c(aid, ts, typ, sess_off, n_aids, kinds, k=20, window=30, max_gap=86400, threads=0, ts_min=None, ts_max=None,
                 rows=True):
    """{kind: (aid_x, aid_y, W) rows} + 'P', same contract as covis_oracle.covis_topk_numpy."""
    lib = _load()
    aid = np.ascontiguousarray(aid, dtype=np.uint32)
    ts = np.ascontiguousarray(ts, dtype=np.int32)
    typ = np.ascontiguousarray(typ, dtype=np.uint8)
    sess_off = np.ascontiguousarray(sess_off, dtype=np.int64)
    t0 = int(ts.min()) if ts_min is None and len(ts) else int(ts_min or 0)
    t1 = int(ts.max()) if ts_max is None and len(ts) else int(ts_max or 0)
    group, param = [], []
    for kd in kinds:
        if kd == 'time_weighted':
            group.append(0); param += [0, 0, 0]
        elif kd in co.TYPE_WEIGHTS:
            group.append(1); param += list(co.TYPE_WEIGHTS[kd])
        else:
            m = co.FILTER_MASKS[kd]
            group.append(2); param += [sum(1 << (tx * 3 + ty) for tx in range(3) for ty in range(3) if m[tx][ty]), 0, 0]
    group = np.array(group, dtype=np.int32)
    param = np.array(param, dtype=np.int32)
    nk = len(kinds)
    oy = np.zeros((nk, n_aids, k), dtype=np.uint32)
    ow = np.zeros((nk, n_aids, k), dtype=np.uint64)
    on = np.zeros((nk, n_aids), dtype=np.int32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    import time
    t_c = time.time()
    P = lib.covis_topk_c(vp(aid), vp(ts), vp(typ), vp(sess_off), C.c_int64(len(sess_off) - 1), C.c_uint32(n_aids),
                         C.c_int(window), C.c_int(max_gap), C.c_int64(t0), C.c_int64(t1), C.c_int(nk), vp(group), vp(param),
                         C.c_int(k), vp(oy), vp(ow), vp(on), C.c_int(threads))
    t_c = time.time() - t_c
    if P < 0:
        raise RuntimeError('covis_topk_c failed')
    out = {'P': int(P), 'seconds_in_c': t_c}     # wall time of the C call alone (what cpu_baseline reports)
    if not rows:
        return out
    ar = np.arange(k)[None, :]
    for j, kd in enumerate(kinds):
        valid = ar < on[j][:, None]
        x = np.broadcast_to(np.arange(n_aids, dtype=np.uint32)[:, None], (n_aids, k))[valid]
        out[kd] = (x, oy[j][valid], ow[j][valid])
    return out
