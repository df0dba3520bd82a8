"""Timing of the candidate lookup (section 8 f1) on one GPU: build the five matrices the recipes read, then run the
click / cart / order recipes over a validation-sized session set (not the bench contract)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from otto_amd.synth import generate_sessions_torch, OTTO_N_AIDS
from otto_amd.covisitation.engine import CovisBuilder
from otto_amd.covisitation import candidates as cd

ap = argparse.ArgumentParser()
ap.add_argument('--train-sessions', type=int, default=3_000_000)
ap.add_argument('--sessions', type=int, default=1_800_000)
ap.add_argument('--k', type=int, default=15)
ap.add_argument('--reps', type=int, default=3)
a = ap.parse_args()
dev = torch.device('cuda:0')
# ONE stream (train and validation sessions share the aid popularity law, as in the dataset): matrices from the first
# train_sessions sessions, lookups over the last `sessions`
d = generate_sessions_torch(a.train_sessions + a.sessions, device=dev)
kinds = ('time_weighted', 'click_weighted', 'cart_weighted', 'click_cart', 'cart_order')
off = d['sess_off'][:a.train_sessions + 1].contiguous()
e = int(off[-1])
b = CovisBuilder(OTTO_N_AIDS, kinds=kinds, ts_min=int(d['ts'].min()), ts_max=int(d['ts'].max()), device=dev)
b.feed(d['aid'][:e].contiguous(), d['ts'][:e].contiguous(), d['type'][:e].contiguous(), off)
mats = b.finalize(k=a.k)
del b
v = {'aid': d['aid'][e:].contiguous(), 'type': d['type'][e:].contiguous(),
     'sess_off': (d['sess_off'][a.train_sessions:] - e).contiguous()}
del d
E = v['aid'].numel()
print(f'sessions {a.sessions}  events {E}  k {a.k}', flush=True)
for name, recipe in (('click', cd.CLICK_RECIPE), ('cart', cd.CART_RECIPE), ('order', cd.ORDER_RECIPE)):
    for r in range(a.reps):
        torch.cuda.synchronize(); t0 = time.time()
        cand, cnt, n = cd.candidate_lookup(v['aid'], v['type'], v['sess_off'], mats, recipe)
        torch.cuda.synchronize(); t1 = time.time()
        print(f'  {name} rep {r}: {1e3*(t1-t0):.2f} ms', flush=True)
    print(f'{name}: {1e3*(t1-t0):.2f} ms  {a.sessions/(t1-t0):.3e} sessions/s  mean candidates {n.float().mean().item():.1f}', flush=True)

for r in range(a.reps):
    torch.cuda.synchronize(); t0 = time.time()
    rc, rw, rn = cd.recency_candidates(v['aid'], v['type'], v['sess_off'])
    torch.cuda.synchronize(); t1 = time.time()
print(f'recency: {1e3*(t1-t0):.2f} ms  {a.sessions/(t1-t0):.3e} sessions/s  mean unique aids {rn.float().mean().item():.1f}', flush=True)
