"""Wall time of `covisitation/builder.py validation` on a synthetic full-size event set (SURVEY.md section 8 f2 remainder):
writes DATA/splits/{train,val}.parquet (seconds + type codes, the splits' schema) into a scratch directory, then times the
script's phases: parquet decode (pyarrow, host), H2D + device sort, the 7-kind build, writing the 2 x 7 part sets.
Not the bench contract."""
import argparse, logging, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument('--sessions', type=int, default=14_571_582)
ap.add_argument('--keep', action='store_true')
a = ap.parse_args()
tmp = tempfile.mkdtemp(prefix='otto_builder_', dir=os.environ.get('TMPDIR', '/tmp'))
os.environ['OTTO_DATA'] = tmp
import numpy as np, torch, pyarrow as pa, pyarrow.parquet as pq
from otto_amd.synth import generate_sessions_torch
from otto_amd.covisitation import builder
from otto_amd.events import frame_to_events_device
dev = torch.device('cuda:0')
d = generate_sessions_torch(a.sessions, device=dev)
S, E = a.sessions, d['aid'].numel()
sess = torch.repeat_interleave(torch.arange(S, device=dev, dtype=torch.int32), d['sess_off'][1:] - d['sess_off'][:-1], output_size=E)
cols = {'session': sess.cpu().numpy(), 'aid': d['aid'].cpu().numpy(), 'ts': d['ts'].cpu().numpy().astype(np.int64), 'type': d['type'].cpu().numpy()}
del d, sess
torch.cuda.empty_cache()
os.makedirs(os.path.join(tmp, 'splits'))
cut = int(E * 0.93)                                   # train / val split of the event rows
t0 = time.time()
for name, sl in (('train', slice(0, cut)), ('val', slice(cut, E))):
    pq.write_table(pa.table({k: v[sl] for k, v in cols.items()}), os.path.join(tmp, 'splits', f'{name}.parquet'), compression='snappy')
print(f'wrote synthetic splits: {E} events, {sum(os.path.getsize(os.path.join(tmp, "splits", f)) for f in os.listdir(os.path.join(tmp, "splits")))/1e9:.2f} GB of parquet in {time.time()-t0:.1f} s', flush=True)
del cols
logging.getLogger().setLevel(logging.WARNING)
for rep in range(2):
    t0 = time.time()
    frames = builder.load_events('validation')
    t1 = time.time()
    ev = frame_to_events_device(frames)
    torch.cuda.synchronize(); t2 = time.time()
    res = builder.build_matrices(ev)
    torch.cuda.synchronize(); t3 = time.time()
    out = os.path.join(tmp, 'covisitation', 'validation')
    os.makedirs(out, exist_ok=True)
    import pathlib
    builder.write_jobs(builder.all_part_jobs(pathlib.Path(out), 'validation', res, ev.n_aids))
    t4 = time.time()
    print(f'rep {rep}: parquet decode (pyarrow, {os.cpu_count()} host threads) {t1-t0:.2f} s | columns -> pinned staging -> device + sort {t2-t1:.2f} s | '
          f'7-kind build incl. top-k rows to host {t3-t2:.2f} s | write 2 x 7 part sets {t4-t3:.2f} s | total {t4-t0:.2f} s', flush=True)
    del frames, ev, res
if not a.keep:
    shutil.rmtree(tmp, ignore_errors=True)
