"""Device engine of the MF path: thin Python over ``include/otto_mf.h``.
torch owns the embedding tables / optimizer state and the stream; every
arithmetic step runs in hand-written gfx950 kernels (``csrc/otto_mf.hip``)."""
import ctypes as C

from .. import _lib

LOSS_MSE, LOSS_BCE = 0, 1
BPR_HOGWILD, BPR_BATCH = 0, 1


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _chk(name, t, dtype, device):
    if t.dtype != dtype or t.device != device or not t.is_contiguous():
        raise ValueError(f'{name}: expected contiguous {dtype} on {device}, got {t.dtype} on {t.device}')


class MFEngine:
    """Workspace (row-owner words, gradient slots) for one pair of embedding tables."""

    def __init__(self, n1, n2, d, max_batch, shared_table=False, device='cuda:0'):
        import torch
        self.torch = torch
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise _lib.OttoError('MFEngine needs a ROCm device (no CPU fallback)')
        self.n1, self.n2, self.d = int(n1), int(n1 if shared_table else n2), int(d)
        self.max_batch, self.shared = int(max_batch), bool(shared_table)
        self._lib = _lib.lib()
        self._ctx = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.otto_mf_create(C.byref(self._ctx), self.n1, self.n2, self.d, self.max_batch,
                                                int(self.shared)), 'otto_mf_create')

    def close(self):
        if getattr(self, '_ctx', None) is not None and self._ctx:
            self._lib.otto_mf_destroy(self._ctx)
            self._ctx = None

    __del__ = close

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def _tables(self, E1, E2):
        t = self.torch
        _chk('E1', E1, t.float32, self.device)
        _chk('E2', E2, t.float32, self.device)
        if E1.shape != (self.n1, self.d) or E2.shape != (self.n2, self.d):
            raise ValueError(f'table shapes {tuple(E1.shape)}, {tuple(E2.shape)} != ({self.n1},{self.d}), ({self.n2},{self.d})')

    def _idx(self, i1, i2, extra=()):
        t = self.torch
        for n, x in (('i1', i1), ('i2', i2)) + tuple(extra):
            _chk(n, x, t.int64, self.device)
        if i1.numel() != i2.numel() or any(x.numel() != i1.numel() for _, x in extra):
            raise ValueError('index / target length mismatch')
        return i1.numel()

    def forward(self, E1, E2, i1, i2, out=None):
        """out[b] = <E1[i1[b]], E2[i2[b]]>  (torch_modules.py:13-19, 32-38)."""
        t = self.torch
        self._tables(E1, E2)
        B = self._idx(i1, i2)
        if out is None:
            out = t.empty(B, dtype=t.float32, device=self.device)
        with t.cuda.device(self.device):
            _lib.check(self._lib.otto_mf_forward(self._ctx, _ptr(E1), _ptr(E2), _ptr(i1), _ptr(i2), B, _ptr(out),
                                                 self._stream()), 'otto_mf_forward')
        return out

    def eval(self, E1, E2, i1, i2, target, loss_kind, loss_out, pred=None):
        """validate() batch body: mean loss into ``loss_out`` (1-element device view), optional predictions."""
        t = self.torch
        self._tables(E1, E2)
        B = self._idx(i1, i2, (('target', target),))
        with t.cuda.device(self.device):
            _lib.check(self._lib.otto_mf_eval(self._ctx, _ptr(E1), _ptr(E2), _ptr(i1), _ptr(i2), _ptr(target), B,
                                              int(loss_kind), _ptr(pred), _ptr(loss_out), self._stream()), 'otto_mf_eval')

    def eval_sums(self, E1, E2, i1, i2, target, loss_kind, loss_out, pred=None):
        """``eval`` + the context's running score sums (sum |p - t|, sum (p - t)^2, hits at 0.5, count) grow by this batch:
        validate() reads four doubles per epoch instead of every prediction (torch_trainer.py:144-158)."""
        t = self.torch
        self._tables(E1, E2)
        B = self._idx(i1, i2, (('target', target),))
        with t.cuda.device(self.device):
            _lib.check(self._lib.otto_mf_eval_sums(self._ctx, _ptr(E1), _ptr(E2), _ptr(i1), _ptr(i2), _ptr(target), B,
                                                   int(loss_kind), _ptr(pred), _ptr(loss_out), self._stream()), 'otto_mf_eval_sums')

    def read_sums(self, reset=True):
        """(sum |p - t|, sum (p - t)^2, hits, count) accumulated by ``eval_sums``; synchronises."""
        buf = (C.c_double * 4)()
        with self.torch.cuda.device(self.device):
            _lib.check(self._lib.otto_mf_read_sums(self._ctx, buf, int(reset), self._stream()), 'otto_mf_read_sums')
        return tuple(float(v) for v in buf)

    def check(self):
        """Raise ``OttoError`` if any kernel since the last call skipped a sample whose row id was outside its table
        (the kernels range-check instead of faulting; ``nn.Embedding`` would raise IndexError in the reference)."""
        with self.torch.cuda.device(self.device):
            _lib.check(self._lib.otto_mf_check(self._ctx, None, self._stream()), 'otto_mf_check')

    def step_sparse_adam(self, E1, m1, v1, E2, m2, v2, i1, i2, target, loss_kind, lr, betas, eps, t_step, loss_out):
        """train() batch body with SparseAdam semantics; mean pre-update loss into ``loss_out``."""
        t = self.torch
        self._tables(E1, E2)
        B = self._idx(i1, i2, (('target', target),))
        with t.cuda.device(self.device):
            _lib.check(self._lib.otto_mf_step_sparse_adam(
                self._ctx, _ptr(E1), _ptr(m1), _ptr(v1), _ptr(E2), _ptr(m2), _ptr(v2), _ptr(i1), _ptr(i2), _ptr(target),
                B, int(loss_kind), float(lr), float(betas[0]), float(betas[1]), float(eps), int(t_step), _ptr(loss_out),
                self._stream()), 'otto_mf_step_sparse_adam')

    def bpr_step(self, U, V, u, i, seed, epoch, row0, lr, l2=0.0, mode=BPR_HOGWILD, loss_sum=None, neg_out=None):
        t = self.torch
        self._tables(U, V)
        B = self._idx(u, i)
        if loss_sum is None:
            loss_sum = t.empty(1, dtype=t.float32, device=self.device)
        with t.cuda.device(self.device):
            _lib.check(self._lib.otto_mf_bpr_step(self._ctx, _ptr(U), _ptr(V), _ptr(u), _ptr(i), B, int(seed), int(epoch),
                                                  int(row0), float(lr), float(l2), int(mode), _ptr(loss_sum),
                                                  _ptr(neg_out), self._stream()), 'otto_mf_bpr_step')
        return loss_sum


def score_topk(U, V, k=20, pad_col=-1):
    """Full-sort scoring: top-k of U @ V.T per row without materialising it
    (recbole/inference.py:76-80). Returns (ids int32 [B,k], scores float32 [B,k])."""
    import torch
    if U.device.type != 'cuda':
        raise _lib.OttoError('score_topk needs a ROCm device (no CPU fallback)')
    for n, x in (('U', U), ('V', V)):
        _chk(n, x, torch.float32, U.device)
    B, d = U.shape
    N = V.shape[0]
    if V.shape[1] != d:
        raise ValueError('factor dimension mismatch')
    lib = _lib.lib()
    ws_bytes = lib.otto_mf_score_workspace(B, N, int(k))
    ws = torch.empty(max(ws_bytes, 8), dtype=torch.uint8, device=U.device)
    ids = torch.empty((B, k), dtype=torch.int32, device=U.device)
    scores = torch.empty((B, k), dtype=torch.float32, device=U.device)
    with torch.cuda.device(U.device):
        _lib.check(lib.otto_mf_score_topk(_ptr(U), _ptr(V), B, N, int(d), int(k), int(pad_col), _ptr(ids), _ptr(scores),
                                          _ptr(ws), ws_bytes, C.c_void_p(torch.cuda.current_stream(U.device).cuda_stream)),
                   'otto_mf_score_topk')
    return ids, scores


def topk_merge(part_scores, part_ids, k):
    """Exact merge of W partial top-k lists per row: ``part_scores`` float32 / ``part_ids`` int32 [W, B, k] (id -1 = empty)
    -> (ids int32 [B, k], scores float32 [B, k]) ordered by (score desc, id asc). Used by the item-sharded scoring."""
    import torch
    if part_scores.device.type != 'cuda':
        raise _lib.OttoError('topk_merge needs a ROCm device (no CPU fallback)')
    _chk('part_scores', part_scores, torch.float32, part_scores.device)
    _chk('part_ids', part_ids, torch.int32, part_scores.device)
    W, B, kk = part_scores.shape
    if part_ids.shape != part_scores.shape or kk != k:
        raise ValueError('partial lists must be [W, B, k]')
    ids = torch.empty((B, k), dtype=torch.int32, device=part_scores.device)
    scores = torch.empty((B, k), dtype=torch.float32, device=part_scores.device)
    with torch.cuda.device(part_scores.device):
        _lib.check(_lib.lib().otto_mf_topk_merge(_ptr(part_scores), _ptr(part_ids), int(W), int(B), int(k), _ptr(ids), _ptr(scores),
                                                 C.c_void_p(torch.cuda.current_stream(part_scores.device).cuda_stream)),
                   'otto_mf_topk_merge')
    return ids, scores
