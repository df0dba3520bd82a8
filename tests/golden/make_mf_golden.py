"""Generates tests/golden/mf_golden.npz by RUNNING THE REFERENCE'S OWN CODE in this container:
``/root/reference/src/matrix_factorization/torch_modules.py`` (MatrixFactorization,
CollaborativeFiltering) and ``torch_trainer.py``'s ``train()`` / ``validate()`` on CPU, with
torch's SparseAdam + StepLR exactly as ``torch_trainer.py:352-353`` builds them.

Modules the reference imports at file top but this path never calls (``settings``, ``polars``,
``merlin``, ``seaborn``) are absent here (ordinary ModuleNotFoundError / hard-coded log path) and
are replaced by empty stand-in modules so the import succeeds; nothing of theirs is executed.
Only data (inputs and outputs) is written to the fixture -- no reference source.

Fixture contents, per model (prefix ``mf_`` / ``cf_``):
  w1_0, w2_0        initial embedding tables (mf: session, aid; cf: w1_0 only)
  i1, i2, target    [n_batches, B] int64 batches (with duplicate rows inside a batch)
  step_loss         [n_epochs * n_batches] per-step loss (our loop with the same calls as train())
  step_w1/w2, step_m1/m2, step_v1/v2   tables + SparseAdam state after the FIRST step
  epoch_train_loss  [n_epochs] value returned by the reference train()
  epoch_val_loss, epoch_val_s0, epoch_val_s1   reference validate() loss and scores
  w1_T, w2_T, m*_T, v*_T     tables + state after the last epoch
  lr_trace          lr after every scheduler step
"""
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
REF = '/root/reference/src/matrix_factorization'
HERE = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_stub('settings')
_stub('polars')
_stub('seaborn')
_stub('merlin')
_stub('merlin.loader')
_stub('merlin.loader.torch', Loader=object)
_stub('merlin.io', Dataset=object)
sys.path.insert(0, REF)
import torch  # noqa: E402
import torch_modules  # noqa: E402  (reference)
import torch_trainer  # noqa: E402  (reference)

torch.set_num_threads(1)


def run(model_class, loss_name, n1, n2, d, B, n_batches, n_epochs, lr, step_size, seed):
    rng = np.random.default_rng(seed)
    out = {}
    if model_class == 'MatrixFactorization':
        model = torch_modules.MatrixFactorization(n_sessions=n1, n_aids=n2, n_factors=d, sparse=True, dropout_probability=0.)
        w1 = rng.standard_normal((n1, d)).astype(np.float32) * 0.5
        w2 = rng.standard_normal((n2, d)).astype(np.float32) * 0.5
        with torch.no_grad():
            model.session_embeddings.weight.copy_(torch.from_numpy(w1))
            model.aid_embeddings.weight.copy_(torch.from_numpy(w2))
        out['w1_0'], out['w2_0'] = w1, w2
        i1 = rng.integers(0, n1, (n_batches, B))
        i2 = np.minimum(rng.zipf(1.6, (n_batches, B)) - 1, n2 - 1)      # popular aids repeat inside a batch
        target = rng.choice(3, (n_batches, B), p=[0.9, 0.08, 0.02])
        keys = ('session', 'aid')
        params = lambda: (model.session_embeddings.weight, model.aid_embeddings.weight)
    else:
        model = torch_modules.CollaborativeFiltering(n_embeddings=n1, n_factors=d, sparse=True, dropout_probability=0.)
        w1 = rng.standard_normal((n1, d)).astype(np.float32) * 0.5
        with torch.no_grad():
            model.embeddings.weight.copy_(torch.from_numpy(w1))
        out['w1_0'] = w1
        i1 = np.minimum(rng.zipf(1.5, (n_batches, B)) - 1, n1 - 1)
        i2 = rng.integers(0, n1, (n_batches, B))
        target = rng.integers(0, 2, (n_batches, B))
        keys = ('x1', 'x2')
        params = lambda: (model.embeddings.weight,)
    out['i1'], out['i2'], out['target'] = i1.astype(np.int64), i2.astype(np.int64), target.astype(np.int64)
    loader = [({keys[0]: torch.from_numpy(out['i1'][b]), keys[1]: torch.from_numpy(out['i2'][b]),
                'target': torch.from_numpy(out['target'][b])}, None) for b in range(n_batches)]

    criterion = getattr(torch.nn, loss_name)()
    optimizer = torch.optim.SparseAdam(model.parameters(), lr=lr, betas=(0.9, 0.999))
    scheduler = torch.optim.lr_scheduler.StepLR(optimizer, step_size=step_size, gamma=0.5, last_epoch=-1)
    device = torch.device('cpu')

    # --- first step alone, with the exact calls of train() (torch_trainer.py:59-78), to capture state ---
    import copy
    m2 = copy.deepcopy(model)
    o2 = torch.optim.SparseAdam(m2.parameters(), lr=lr, betas=(0.9, 0.999))
    inputs = loader[0][0]
    outputs = m2(inputs[keys[0]], inputs[keys[1]])
    loss = criterion(outputs, inputs['target'].float())
    o2.zero_grad()
    loss.backward()
    o2.step()
    p2 = list(m2.parameters())
    for n, p in zip(('1', '2'), p2):
        out[f'step_w{n}'] = p.detach().numpy().copy()
        out[f'step_m{n}'] = o2.state[p]['exp_avg'].numpy().copy()
        out[f'step_v{n}'] = o2.state[p]['exp_avg_sq'].numpy().copy()
    out['step0_pred'] = outputs.detach().numpy().copy()

    # --- per-step losses with the same calls, on a second copy (train() only returns the mean) ---
    m3 = copy.deepcopy(model)
    o3 = torch.optim.SparseAdam(m3.parameters(), lr=lr, betas=(0.9, 0.999))
    s3 = torch.optim.lr_scheduler.StepLR(o3, step_size=step_size, gamma=0.5, last_epoch=-1)
    step_loss, lr_trace = [], []
    for _ in range(n_epochs):
        for inputs, _n in loader:
            outputs = m3(inputs[keys[0]], inputs[keys[1]])
            loss = criterion(outputs, inputs['target'].float())
            o3.zero_grad()
            loss.backward()
            o3.step()
            s3.step()
            step_loss.append(loss.detach().item())
            lr_trace.append(s3.get_last_lr()[0])
    out['step_loss'] = np.array(step_loss, dtype=np.float64)
    out['lr_trace'] = np.array(lr_trace, dtype=np.float64)

    # --- the reference's train() / validate() themselves ---
    tl, vl, s0, s1 = [], [], [], []
    for _ in range(n_epochs):
        tl.append(torch_trainer.train(loader, model, criterion, optimizer, device, scheduler))
        v, sc = torch_trainer.validate(loader, model, criterion, device, scores=True)
        vl.append(v)
        vals = list(sc.values())
        s0.append(vals[0])
        s1.append(vals[1])
        score_names = list(sc.keys())
    out['epoch_train_loss'] = np.array(tl, dtype=np.float64)
    out['epoch_val_loss'] = np.array(vl, dtype=np.float64)
    out['epoch_val_s0'] = np.array(s0, dtype=np.float64)
    out['epoch_val_s1'] = np.array(s1, dtype=np.float64)
    out['score_names'] = np.array(score_names)
    for n, p in zip(('1', '2'), params()):
        out[f'w{n}_T'] = p.detach().numpy().copy()
        out[f'm{n}_T'] = optimizer.state[p]['exp_avg'].numpy().copy()
        out[f'v{n}_T'] = optimizer.state[p]['exp_avg_sq'].numpy().copy()
    # the two loops must agree (same arithmetic), else the fixture is inconsistent
    assert np.allclose(np.mean(out['step_loss'].reshape(n_epochs, -1), axis=1), out['epoch_train_loss'], rtol=1e-6)
    out['hyper'] = np.array([n1, n2, d, B, n_batches, n_epochs, step_size], dtype=np.int64)
    out['lr'] = np.float64(lr)
    return out


if __name__ == '__main__':
    res = {}
    for k, v in run('MatrixFactorization', 'MSELoss', 300, 120, 32, 256, 6, 3, 0.05, 5, 11).items():
        res['mf_' + k] = v
    for k, v in run('CollaborativeFiltering', 'BCEWithLogitsLoss', 150, 150, 16, 128, 5, 3, 0.0005, 4, 12).items():
        res['cf_' + k] = v
    np.savez_compressed(os.path.join(HERE, 'mf_golden.npz'), **res)
    print('wrote mf_golden.npz:', {k: getattr(v, 'shape', None) for k, v in res.items() if k.endswith('loss')})
    print('mf epoch losses', res['mf_epoch_train_loss'], res['mf_epoch_val_loss'], res['mf_score_names'])
    print('cf epoch losses', res['cf_epoch_train_loss'], res['cf_epoch_val_loss'], res['cf_score_names'])
