"""PCIe-inclusive rate of the covisitation build (section 8 f2): host SoA -> pinned staging -> device -> feed, then
finalize. Not the bench contract (bench.py times device-resident inputs)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from otto_amd.synth import generate_sessions_torch, OTTO_N_AIDS, Events
from otto_amd.covisitation.engine import CovisBuilder
from otto_amd.ingest import feed_host_events

ap = argparse.ArgumentParser()
ap.add_argument('--sessions', type=int, default=14_571_582)
ap.add_argument('--chunk', type=int, default=2_000_000)
a = ap.parse_args()
dev = torch.device('cuda:0')
d = generate_sessions_torch(a.sessions, device=dev)
ev = Events(aid=d['aid'].cpu().numpy().view(np.uint32), ts=d['ts'].cpu().numpy(), type=d['type'].cpu().numpy(),
            sess_off=d['sess_off'].cpu().numpy(), n_aids=OTTO_N_AIDS)
del d
torch.cuda.empty_cache()
kinds = ('click_weighted', 'cart_weighted', 'order_weighted')
b = CovisBuilder(OTTO_N_AIDS, kinds=kinds, ts_min=int(ev.ts.min()), ts_max=int(ev.ts.max()), device=dev)
for pinned in (True, False, True):
    b.reset()
    with torch.cuda.device(dev):
        sec, nbytes = feed_host_events(b, ev, dev, chunk_sessions=a.chunk, pinned=pinned)
    torch.cuda.synchronize(); t0 = time.time()
    b.finalize(k=20)
    torch.cuda.synchronize(); t1 = time.time()
    pairs = b.stats()['pairs']
    print(f'pinned={pinned}: ingest+expand {1e3*sec:.1f} ms ({nbytes/1e9:.2f} GB, {nbytes/sec/1e9:.1f} GB/s)  finalize {1e3*(t1-t0):.1f} ms  '
          f'PCIe-inclusive {pairs/(sec+t1-t0):.3e} pairs/s', flush=True)
