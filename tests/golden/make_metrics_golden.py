"""Generates tests/golden/metrics_golden.json by calling the reference's own ``src/metrics.py``
(click_recall / cart_order_recall, pure NumPy, importable as is) on seeded random cases plus the
worked example of the reference's EDA notebook (cells 35-45: weighted recall 0.7)."""
import json
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, '/root/reference/src')
import metrics as ref  # noqa: E402

rng = np.random.default_rng(5)
cases = []
for _ in range(60):
    gt = rng.integers(0, 40, rng.integers(0, 30)).tolist()
    pred = rng.permutation(40)[:rng.integers(1, 21)].tolist()
    c = ref.click_recall(gt[:1], pred)
    o = ref.cart_order_recall(gt, pred)
    cases.append({'gt': gt, 'pred': pred, 'click': None if isinstance(c, float) and np.isnan(c) else c,
                  'cart_order': None if isinstance(o, float) and np.isnan(o) else o})
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'metrics_golden.json'), 'w') as f:
    json.dump(cases, f)
print(len(cases), 'cases')
