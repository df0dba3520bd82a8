"""Same classes, constructor arguments, parameter names and forward semantics as the reference's
``src/matrix_factorization/torch_modules.py`` (CollaborativeFiltering ``:4-19``,
MatrixFactorization ``:22-38``), so ``state_dict()`` checkpoints are interchangeable
(``embeddings.weight`` / ``session_embeddings.weight`` + ``aid_embeddings.weight``).
``nn.Embedding`` only owns the tables; the gather + dot runs in the HIP kernel
(``otto_mf_forward``) and training goes through ``torch_optim.SparseAdam.fused_step``.
"""
import torch.nn as nn

from .engine import MFEngine


class _FusedMixin:
    _engine = None

    def _tables(self):
        raise NotImplementedError

    def engine(self, batch):
        """Workspace sized for ``batch`` (recreated when a larger batch arrives)."""
        E1, E2, shared = self._tables()
        if self._engine is None or self._engine.max_batch < batch or self._engine.device != E1.device:
            if self._engine is not None:
                self._engine.close()
            self._engine = MFEngine(E1.shape[0], E2.shape[0], E1.shape[1], max(int(batch), 1), shared_table=shared,
                                    device=E1.device)
        return self._engine

    def _forward(self, i1, i2):
        E1, E2, _ = self._tables()
        return self.engine(i1.numel()).forward(E1.detach(), E2.detach(), i1.contiguous(), i2.contiguous())


class CollaborativeFiltering(nn.Module, _FusedMixin):

    def __init__(self, n_embeddings, n_factors, sparse=False, dropout_probability=0):
        super(CollaborativeFiltering, self).__init__()
        if dropout_probability > 0:
            raise NotImplementedError('dropout_probability > 0 is not supported by the fused kernels (reference configs use 0.)')
        self.embeddings = nn.Embedding(num_embeddings=n_embeddings, embedding_dim=n_factors, sparse=sparse)
        self.dropout = nn.Identity()

    def _tables(self):
        return self.embeddings.weight, self.embeddings.weight, True

    def forward(self, x1, x2):
        return self._forward(x1, x2)


class MatrixFactorization(nn.Module, _FusedMixin):

    def __init__(self, n_sessions, n_aids, n_factors, sparse=False, dropout_probability=0):
        super(MatrixFactorization, self).__init__()
        if dropout_probability > 0:
            raise NotImplementedError('dropout_probability > 0 is not supported by the fused kernels (reference configs use 0.)')
        self.session_embeddings = nn.Embedding(num_embeddings=n_sessions, embedding_dim=n_factors, sparse=sparse)
        self.aid_embeddings = nn.Embedding(num_embeddings=n_aids, embedding_dim=n_factors, sparse=sparse)
        self.dropout = nn.Identity()

    def _tables(self):
        return self.session_embeddings.weight, self.aid_embeddings.weight, False

    def forward(self, sessions, aids):
        return self._forward(sessions, aids)
