"""``SparseAdam`` with the constructor and state layout of ``torch.optim.SparseAdam`` (what the
reference builds by reflection, ``src/matrix_factorization/torch_trainer.py:352``), but whose step
is fused with forward/backward in the HIP kernels: ``fused_step`` = the reference's
``model(...)``, ``criterion``, ``zero_grad``, ``backward``, ``optimizer.step()``
(``torch_trainer.py:62-75``).  Subclassing ``torch.optim.Optimizer`` keeps torch's LR schedulers
(``StepLR`` per batch, ``torch_trainer.py:76-77``) working unchanged.
"""
import torch

from .engine import LOSS_MSE, LOSS_BCE


def loss_kind(criterion):
    if isinstance(criterion, torch.nn.MSELoss):
        kind = LOSS_MSE
    elif isinstance(criterion, torch.nn.BCEWithLogitsLoss):
        kind = LOSS_BCE
    else:
        raise ValueError(f'unsupported loss function {type(criterion).__name__} (MSELoss, BCEWithLogitsLoss)')
    if getattr(criterion, 'reduction', 'mean') != 'mean' or getattr(criterion, 'pos_weight', None) is not None \
            or getattr(criterion, 'weight', None) is not None:
        raise ValueError("only reduction='mean' without weights is supported")
    return kind


class SparseAdam(torch.optim.Optimizer):

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, maximize=False):
        if maximize:
            raise ValueError('maximize is not supported')
        if not 0.0 < lr or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or not 0.0 < eps:
            raise ValueError('invalid SparseAdam hyper-parameters')
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps))

    def _state(self, p):
        st = self.state[p]
        if len(st) == 0:
            st['step'] = 0
            st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    def step(self, closure=None):
        raise RuntimeError('SparseAdam here is fused with forward/backward: call fused_step (torch_trainer.train does)')

    def fused_step(self, model, i1, i2, targets, criterion, loss_out):
        """One training batch; the mean pre-update loss lands in ``loss_out`` (1-element device tensor)."""
        group = self.param_groups[0]
        self._opt_called = True     # torch's LR schedulers only check that an optimizer step preceded scheduler.step()
        E1, E2, shared = model._tables()
        s1 = self._state(E1)
        s2 = s1 if shared else self._state(E2)
        s1['step'] += 1
        if not shared:
            s2['step'] += 1
        model.engine(i1.numel()).step_sparse_adam(
            E1.data, s1['exp_avg'], s1['exp_avg_sq'], E2.data, s2['exp_avg'], s2['exp_avg_sq'],
            i1, i2, targets, loss_kind(criterion), group['lr'], group['betas'], group['eps'], s1['step'], loss_out)
