"""``set_seed`` as in the reference's ``src/matrix_factorization/torch_utils.py:7-30``."""
import os
import random

import numpy as np
import torch


def set_seed(seed, deterministic_cudnn=False):
    if deterministic_cudnn:
        torch.backends.cudnn.deterministic = True
        torch.backends.cudnn.benchmark = False
    os.environ['PYTHONHASHSEED'] = str(seed)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
