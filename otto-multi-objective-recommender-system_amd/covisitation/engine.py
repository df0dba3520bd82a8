"""Device engine of the covisitation builder: thin Python over the C-ABI of
``include/otto_covis.h`` (hand-written gfx950 kernels in ``csrc/otto_covis.hip``).

torch is used only to own device buffers and the stream.  No CPU fallback.
"""
import ctypes as C
import numpy as np

from .. import _lib
from .spec import TYPE_WEIGHTS, FILTER_MASKS, TIME_KIND, ALL_KINDS, mask_bits


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class CovisBuilder:
    """One context per device. ``feed`` session chunks, then ``finalize``.

    Parameters mirror SPEC-COVIS: ``window`` (tail window W), ``max_gap`` seconds,
    ``kinds`` (subset of ``spec.ALL_KINDS``), ``ts_min``/``ts_max`` = t0/t1 of the
    time weight (global over every chunk and rank).
    """

    def __init__(self, n_aids, kinds=ALL_KINDS, window=30, max_gap=86400, ts_min=0, ts_max=0, device='cuda:0'):
        import torch
        self.torch = torch
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise _lib.OttoError('CovisBuilder needs a ROCm device (no CPU fallback)')
        unknown = [k for k in kinds if k not in ALL_KINDS]
        if unknown:
            raise ValueError(f'unknown covisitation kinds {unknown}')
        self.kinds = tuple(kinds)
        self.n_aids = int(n_aids)
        self.type_kinds = [k for k in self.kinds if k in TYPE_WEIGHTS]
        self.filter_kinds = [k for k in self.kinds if k in FILTER_MASKS]
        self.want_time = TIME_KIND in self.kinds
        p = _lib.CovisParams()
        p.window, p.max_gap, p.n_aids = int(window), int(max_gap), self.n_aids
        p.ts_min, p.ts_max, p.want_time = int(ts_min), int(ts_max), int(self.want_time)
        p.n_filters = len(self.filter_kinds)
        for i, k in enumerate(self.filter_kinds):
            p.filter_mask[i] = mask_bits(FILTER_MASKS[k])
        p.n_type_weights = len(self.type_kinds)
        for i, k in enumerate(self.type_kinds):
            for t in range(3):
                p.type_weight[i][t] = TYPE_WEIGHTS[k][t]
        self._lib = _lib.lib()
        self._ctx = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.otto_covis_create(C.byref(self._ctx), C.byref(p)), 'otto_covis_create')

    def close(self):
        if getattr(self, '_ctx', None) is not None and self._ctx:
            self._lib.otto_covis_destroy(self._ctx)
            self._ctx = None

    __del__ = close

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def reset(self):
        _lib.check(self._lib.otto_covis_reset(self._ctx), 'otto_covis_reset')

    def feed(self, aid, ts, typ, sess_off):
        """K1 pair-expand over one session chunk (device tensors: aid int32, ts int32,
        type uint8, sess_off int64 CSR with n_sess+1 entries)."""
        t = self.torch
        for name, x, dt in (('aid', aid, t.int32), ('ts', ts, t.int32), ('type', typ, t.uint8), ('sess_off', sess_off, t.int64)):
            if x.dtype != dt or x.device != self.device or not x.is_contiguous():
                raise ValueError(f'{name}: expected contiguous {dt} on {self.device}, got {x.dtype} on {x.device}')
        n_sess = sess_off.numel() - 1
        if aid.numel() != ts.numel() or aid.numel() != typ.numel():
            raise ValueError('aid/ts/type length mismatch')
        with t.cuda.device(self.device):
            _lib.check(self._lib.otto_covis_feed(self._ctx, _ptr(aid), _ptr(ts), _ptr(typ), _ptr(sess_off),
                                                 C.c_int64(n_sess), self._stream()), 'otto_covis_feed')

    def _finalize_group(self, group, n_kinds, k, out=None):
        t = self.torch
        if out is None:
            out = (t.empty((n_kinds, self.n_aids, k), dtype=t.int32, device=self.device),
                   t.empty((n_kinds, self.n_aids, k), dtype=t.int64, device=self.device),
                   t.empty((n_kinds, self.n_aids), dtype=t.int32, device=self.device))
        y, w, n = out
        with t.cuda.device(self.device):
            _lib.check(self._lib.otto_covis_finalize(self._ctx, group, int(k), _ptr(y), _ptr(w), _ptr(n), self._stream()),
                       'otto_covis_finalize')
        return y, w, n

    def finalize(self, k=20, out=None):
        """Reduce + top-k. Returns {kind: (aid_y int32 [n_aids,k], W int64 [n_aids,k], n int32 [n_aids])}.

        ``out`` optionally maps group id -> preallocated (y, w, n) tensors (bench reuse).
        """
        res = {}
        groups = ((_lib.GROUP_TYPE, self.type_kinds), (_lib.GROUP_FILTER, self.filter_kinds),
                  (_lib.GROUP_TIME, [TIME_KIND] if self.want_time else []))
        for group, names in groups:
            if not names:
                continue
            y, w, n = self._finalize_group(group, len(names), k, None if out is None else out.get(group))
            for i, name in enumerate(names):
                res[name] = (y[i], w[i], n[i])
        return res

    def set_option(self, name, value):
        _lib.check(self._lib.otto_covis_set_option(self._ctx, name.encode(), C.c_int64(int(value))), 'otto_covis_set_option')

    def stats(self):
        buf = (C.c_int64 * len(_lib.STAT_NAMES))()
        _lib.check(self._lib.otto_covis_stats(self._ctx, buf), 'otto_covis_stats')
        return dict(zip(_lib.STAT_NAMES, [int(v) for v in buf]))

    def timings(self):
        """Per-kernel device milliseconds (hipEvents on the stream) of the last feed/finalize."""
        buf = (C.c_float * len(_lib.TIMING_NAMES))()
        self._lib.otto_covis_timings(self._ctx, buf)
        return dict(zip(_lib.TIMING_NAMES, [float(v) for v in buf]))

    def kernel_names(self):
        """{timing slot: kernels launched under it since the last reset} as the library itself reports them."""
        out = {}
        for i, name in enumerate(_lib.TIMING_NAMES):
            buf = C.create_string_buffer(1024)
            _lib.check(self._lib.otto_covis_kernel_names(self._ctx, i, buf, 1024), 'otto_covis_kernel_names')
            out[name] = buf.value.decode()
        return out

    def copy_records(self):
        """Test hook: raw K1 output on the host (rec, tw or None, run_x, run_desc)."""
        st = self.stats()
        rec = np.empty(st['pair_slots'], dtype=np.uint32)
        tw = np.empty(st['pair_slots'], dtype=np.uint32) if self.want_time else None
        run_x = np.empty(st['tail_events'], dtype=np.uint32)
        run_desc = np.empty(st['tail_events'], dtype=np.uint64)
        vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else C.c_void_p(0)
        with self.torch.cuda.device(self.device):
            _lib.check(self._lib.otto_covis_copy_records(self._ctx, vp(rec), vp(tw), vp(run_x), vp(run_desc)),
                       'otto_covis_copy_records')
        return rec, tw, run_x, run_desc

    # ---- multi-GPU exchange (SURVEY.md section 8 e) ------------------------------------------------
    def export_runs(self, x_lo, x_hi):
        """Runs with aid_x in [x_lo, x_hi) as device tensors (hdr int32 [n_runs,2], rec int32, tw int32|None)."""
        t = self.torch
        nr, nc = C.c_int64(), C.c_int64()
        with t.cuda.device(self.device):
            _lib.check(self._lib.otto_covis_export_count(self._ctx, int(x_lo), int(x_hi), C.byref(nr), C.byref(nc),
                                                         self._stream()), 'otto_covis_export_count')
            hdr = t.empty((nr.value, 2), dtype=t.int32, device=self.device)
            rec = t.empty(nc.value, dtype=t.int32, device=self.device)
            tw = t.empty(nc.value, dtype=t.int32, device=self.device) if self.want_time else None
            _lib.check(self._lib.otto_covis_export_runs(self._ctx, int(x_lo), int(x_hi), _ptr(hdr), _ptr(rec), _ptr(tw),
                                                        self._stream()), 'otto_covis_export_runs')
        return hdr, rec, tw

    def export_all(self, bounds):
        """Every owner's piece in ONE owner-major buffer (owner o = aid_x in [bounds[o], bounds[o+1])): returns
        (hdr int32 [n_runs,2], rec int32 [n_recs], tw|None, runs_per_owner, recs_per_owner) -- the send buffers and
        split sizes of the all-to-all-v."""
        t = self.torch
        W = len(bounds) - 1
        hb = (C.c_uint32 * (W + 1))(*[int(b) for b in bounds])
        nr, nc = (C.c_int64 * W)(), (C.c_int64 * W)()
        with t.cuda.device(self.device):
            _lib.check(self._lib.otto_covis_export_plan(self._ctx, W, hb, nr, nc, self._stream()), 'otto_covis_export_plan')
            runs, recs = [int(v) for v in nr], [int(v) for v in nc]
            hdr = t.empty((sum(runs), 2), dtype=t.int32, device=self.device)
            rec = t.empty(sum(recs), dtype=t.int32, device=self.device)
            tw = t.empty(sum(recs), dtype=t.int32, device=self.device) if self.want_time else None
            _lib.check(self._lib.otto_covis_export_fill(self._ctx, W, hb, _ptr(hdr), _ptr(rec), _ptr(tw), self._stream()),
                       'otto_covis_export_fill')
        return hdr, rec, tw, runs, recs

    def export_plan_range(self, bounds, slot_lo, slot_hi):
        """(runs_per_owner, recs_per_owner) of the run slots [slot_lo, slot_hi) (chunked exchange; synchronises the stream)."""
        W = len(bounds) - 1
        hb = (C.c_uint32 * (W + 1))(*[int(b) for b in bounds])
        nr, nc = (C.c_int64 * W)(), (C.c_int64 * W)()
        with self.torch.cuda.device(self.device):
            _lib.check(self._lib.otto_covis_export_plan_range(self._ctx, W, hb, C.c_int64(int(slot_lo)), C.c_int64(int(slot_hi)), nr, nc,
                                                              self._stream()), 'otto_covis_export_plan_range')
        return [int(v) for v in nr], [int(v) for v in nc]

    def export_fill_range(self, bounds, slot_lo, slot_hi, runs, recs, hdr, rec, tw=None):
        """Fills the owner-major send buffers of one slot range (``hdr`` int32 [sum(runs), 2], ``rec`` int32 [sum(recs)],
        ``tw`` like rec or None) on the current stream; no synchronisation."""
        W = len(bounds) - 1
        hb = (C.c_uint32 * (W + 1))(*[int(b) for b in bounds])
        nr, nc = (C.c_int64 * W)(*runs), (C.c_int64 * W)(*recs)
        with self.torch.cuda.device(self.device):
            _lib.check(self._lib.otto_covis_export_fill_range(self._ctx, W, hb, C.c_int64(int(slot_lo)), C.c_int64(int(slot_hi)), nr, nc,
                                                              _ptr(hdr), _ptr(rec), _ptr(tw), self._stream()), 'otto_covis_export_fill_range')

    def run_slots(self):
        return self.stats()['tail_events']

    def import_reserve(self, n_recs):
        """Zero-copy receive buffers: int32 tensors (rec, tw|None) that alias the end of the context's own record arrays.
        Fill them (e.g. as the output of the all-to-all-v) and hand them to ``import_runs``: no device copy is made."""
        t = self.torch
        pr, pt = C.c_void_p(), C.c_void_p()
        with t.cuda.device(self.device):
            _lib.check(self._lib.otto_covis_import_reserve(self._ctx, C.c_int64(int(n_recs)), C.byref(pr), C.byref(pt),
                                                           self._stream()), 'otto_covis_import_reserve')

        def wrap(ptr):
            if not ptr or n_recs == 0:
                return None
            holder = type('_DevPtr', (), {'__cuda_array_interface__': {'shape': (int(n_recs),), 'typestr': '<i4',
                                                                       'data': (int(ptr), False), 'version': 2}})()
            return t.as_tensor(holder, device=self.device)
        return wrap(pr.value), wrap(pt.value)

    def import_runs(self, hdr, rec, tw=None):
        t = self.torch
        with t.cuda.device(self.device):
            _lib.check(self._lib.otto_covis_import_runs(self._ctx, _ptr(hdr), C.c_int64(hdr.shape[0]), _ptr(rec), _ptr(tw),
                                                        C.c_int64(rec.numel()), self._stream()), 'otto_covis_import_runs')


def topk_to_rows(y, w, n):
    """Dense device top-k -> host rows (aid_x, aid_y, W) in (aid_x asc, rank asc) order (SPEC-COVIS 9)."""
    import torch
    k = y.shape[1]
    valid = torch.arange(k, device=y.device)[None, :] < n[:, None]
    aid_x = torch.arange(y.shape[0], device=y.device, dtype=torch.int32)[:, None].expand(-1, k)[valid]
    # (views, not astype: the values are non-negative, and a same-width astype is one more host pass over ~16 bytes per row --
    # 7 GB for the 14 row sets of a full-size build)
    def host(t, dt):
        a = t.cpu().numpy()
        return a.view(dt) if a.dtype.itemsize == np.dtype(dt).itemsize else a.astype(dt)
    return host(aid_x, np.uint32), host(y[valid], np.uint32), host(w[valid], np.uint64)
