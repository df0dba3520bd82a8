"""Regenerates tests/golden/covis_golden.npz: a small seeded synthetic stream and the
top-20 rows of all 8 kinds from the NumPy oracle (cross-checked against the pure-Python
oracle before writing). The reference has no covisitation builder to generate vectors
from -- these pin the oracle (and through it the HIP path) against drift."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'oracle')]
import covis_oracle as co  # noqa: E402
from otto_amd.synth import generate_sessions  # noqa: E402

ev = generate_sessions(400, n_aids=250, seed=2024)
sp = co.CovisSpec()
py = co.covis_pairs_python(ev.aid, ev.ts, ev.type, ev.sess_off, sp)
nu = co.covis_pairs_numpy(ev.aid, ev.ts, ev.type, ev.sess_off, sp)
out = {'aid': ev.aid, 'ts': ev.ts, 'type': ev.type, 'sess_off': ev.sess_off, 'n_aids': np.int64(ev.n_aids)}
for k in co.ALL_KINDS:
    a = co.pairs_dict_to_arrays(py[k])
    assert all(np.array_equal(p, q) for p, q in zip(a, nu[k])), k
    x, y, W = co.topk_rows(*nu[k], k=20)
    out[f'{k}_x'], out[f'{k}_y'], out[f'{k}_w'] = x, y, W
np.savez_compressed(os.path.join(HERE, 'covis_golden.npz'), **out)
print('wrote covis_golden.npz', {k: len(out[f'{k}_x']) for k in co.ALL_KINDS})
