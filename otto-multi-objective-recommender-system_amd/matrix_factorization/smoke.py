"""MF leg of ``__graft_entry__.smoke()``: one fused SparseAdam step, one BPR batch step and one
scoring call on cuda:0, each checked against the CPU oracle."""
import numpy as np


def run(dev):
    import torch
    import mf_oracle as mo
    from .engine import MFEngine, score_topk, BPR_BATCH
    rng = np.random.default_rng(3)
    n1, n2, d, B = 2000, 500, 32, 4096
    E1 = (rng.standard_normal((n1, d)) * 0.3).astype(np.float32)
    E2 = (rng.standard_normal((n2, d)) * 0.3).astype(np.float32)
    i1, i2, tg = rng.integers(0, n1, B), np.minimum(rng.zipf(1.5, B) - 1, n2 - 1), rng.integers(0, 3, B)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    dE1, dE2 = t(E1), t(E2)
    st = [torch.zeros_like(dE1), torch.zeros_like(dE1), torch.zeros_like(dE2), torch.zeros_like(dE2)]
    eng = MFEngine(n1, n2, d, B, device=dev)
    loss = torch.zeros(1, device=dev)
    eng.step_sparse_adam(dE1, st[0], st[1], dE2, st[2], st[3], t(i1), t(i2), t(tg), 0, 0.05, (0.9, 0.999), 1e-8, 1, loss)
    m1, v1, m2, v2 = (np.zeros_like(E1), np.zeros_like(E1), np.zeros_like(E2), np.zeros_like(E2))
    want, _ = mo.sparse_adam_step(E1, m1, v1, E2, m2, v2, i1, i2, tg, 'MSELoss', 0.05, step=1)
    assert abs(loss.item() - want) <= 1e-4 * abs(want), (loss.item(), want)
    assert np.linalg.norm(dE2.cpu().numpy() - E2) <= 1e-4 * np.linalg.norm(E2)
    ls = eng.bpr_step(dE1, dE2, t(i1), t(i2), seed=1, epoch=0, row0=0, lr=0.05, mode=BPR_BATCH)
    j = mo.bpr_negatives(1, 0, 0, i2, n2)
    wl = mo.bpr_step_batch(E1, E2, i1, i2, j, 0.05)
    assert abs(ls.item() - wl) <= 1e-4 * abs(wl), (ls.item(), wl)
    ids, sc = score_topk(dE1[:200].contiguous(), dE2, k=20, pad_col=0)
    wi, ws = mo.score_topk(dE1[:200].cpu().numpy(), dE2.cpu().numpy(), k=20, pad_col=0)
    assert np.allclose(sc.cpu().numpy(), ws, rtol=1e-4, atol=1e-5)
    print('smoke: MF OK (SparseAdam step, BPR batch step, MFMA scoring top-20 within 1e-4 of the oracle)')
