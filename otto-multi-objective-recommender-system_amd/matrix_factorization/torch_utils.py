"""Seeding helper with the call signature of the reference's ``src/matrix_factorization/torch_utils.py:7-30``
(``set_seed(seed, deterministic_cudnn)``), as the trainer script calls it (``torch_trainer.py:325``)."""
import os
import random

import numpy as np
import torch


def _seed_everything(seed):
    for seeder in (random.seed, np.random.seed, torch.manual_seed):
        seeder(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)      # every visible device, the current one included


def set_seed(seed, deterministic_cudnn=False):
    """Seed Python, NumPy and torch (host + every device). ``deterministic_cudnn`` is accepted for config compatibility:
    the fused HIP step has no cuDNN / MIOpen autotuned path, so it only pins torch's own backend switches."""
    seed = int(seed)
    os.environ['PYTHONHASHSEED'] = str(seed)
    _seed_everything(seed)
    backend = torch.backends.cudnn
    flag = bool(deterministic_cudnn)
    backend.deterministic = backend.deterministic or flag
    backend.benchmark = backend.benchmark and not flag
