"""Diagnostic (not a test): 1-rank BPR loss curves of the planted data of tests/test_integration_gpu.py for several
launch sizes, to separate launch-size effects from data-parallel effects in the 2-rank band."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')]
import numpy as np, torch
from test_integration_gpu import _planted
from otto_amd.matrix_factorization.bpr import BPR, train_epoch
dev = torch.device('cuda:0')
u, i, held, n_users, n_items, d = _planted(128)
du, di = torch.from_numpy(u).to(dev), torch.from_numpy(i).to(dev)
for rpl in (36000, 16384, 8192, 4096, 2048):
    for perm in (False, True):
        torch.manual_seed(0)
        m = BPR(n_users, n_items, d)
        with torch.no_grad():
            m.user_embedding.weight.normal_(0, 0.1); m.item_embedding.weight.normal_(0, 0.1)
        m.to(dev)
        if perm:
            p = torch.randperm(du.numel(), device=dev)
            uu, ii = du[p].contiguous(), di[p].contiguous()
        else:
            uu, ii = du, di
        L = [train_epoch(m, uu, ii, lr=0.2, seed=1, epoch=e, rows_per_launch=rpl) for e in range(40)]
        print(f'rows_per_launch {rpl:6d} shuffled {perm}: ' + ' '.join(f'{x:.4f}' for x in L[::5]) + f' final {L[-1]:.4f}', flush=True)
