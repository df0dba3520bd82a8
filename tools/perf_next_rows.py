"""Timing of the rows section 8 marks "next" on one GPU with device-resident inputs (not the bench contract):
f2 event ingest (stable (session, ts) radix sort of a shuffled full-OTTO-shape frame + CSR), a6 aid-pair builders ('diff' over
the whole stream, 'time' over the reference's 15 % row sample) and f4 interaction features over a validation-sized
candidate table."""
import argparse, ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from otto_amd import _lib
from otto_amd.events import DeviceEvents
from otto_amd.synth import generate_sessions_torch, OTTO_N_AIDS, OTTO_N_SESSIONS
from otto_amd.matrix_factorization.data import build_aid_pairs_device
from otto_amd.ranker.interaction_feature_engineering import interaction_features

ap = argparse.ArgumentParser()
ap.add_argument('--sessions', type=int, default=OTTO_N_SESSIONS)
ap.add_argument('--cand-sessions', type=int, default=1_800_000)
ap.add_argument('--reps', type=int, default=3)
a = ap.parse_args()
dev = torch.device('cuda:0')
lib = _lib.lib()
p = lambda t: C.c_void_p(t.data_ptr())
stream = lambda: C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def timed(fn, reps=a.reps):
    best = 1e30
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.time()
        out = fn()
        torch.cuda.synchronize(); best = min(best, time.time() - t0)
    return best, out


d = generate_sessions_torch(a.sessions, device=dev)
E, S = d['aid'].numel(), a.sessions
print(f'sessions {S}  events {E}', flush=True)

# ---- f2: a frame in arbitrary row order -> sorted SoA + CSR ----
sess = torch.repeat_interleave(torch.arange(S, device=dev, dtype=torch.int32), d['sess_off'][1:] - d['sess_off'][:-1], output_size=E)
perm = torch.randperm(E, device=dev)
f_sess, f_aid, f_type = sess[perm].contiguous(), d['aid'][perm].contiguous(), d['type'][perm].contiguous()
f_ts = (d['ts'][perm].to(torch.int64) * 1000 + 7).contiguous()          # milliseconds, as the raw dataset stores them
del perm, sess
ws_b = lib.otto_events_sort_workspace(E)
ws = torch.empty(int(ws_b), dtype=torch.uint8, device=dev)
o_aid = torch.empty(E, dtype=torch.int32, device=dev); o_ts = torch.empty(E, dtype=torch.int32, device=dev)
o_type = torch.empty(E, dtype=torch.uint8, device=dev); o_order = torch.empty(E, dtype=torch.int32, device=dev)
o_off = torch.empty(E + 1, dtype=torch.int64, device=dev); o_id = torch.empty(E, dtype=torch.int32, device=dev)
ns = C.c_int64()


def sort_events():
    _lib.check(lib.otto_events_sort(p(f_sess), p(f_ts), p(f_aid), p(f_type), E, 1000, p(o_aid), p(o_ts), p(o_type), p(o_order), p(o_off),
                                    p(o_id), C.byref(ns), p(ws), int(ws_b), stream()), 'otto_events_sort')


t, _ = timed(sort_events)
# events of one session with equal seconds keep their (shuffled) input order, so only ts is comparable element-wise
assert ns.value == S and bool((o_ts == d['ts']).all()) and int(o_aid.sum()) == int(d['aid'].sum())
in_bytes = E * (4 + 8 + 4 + 1)
print(f'f2 otto_events_sort: {1e3*t:.2f} ms  {E/t:.3e} events/s  ({in_bytes/t/1e9:.0f} GB/s of input columns; workspace {ws_b/1e9:.2f} GB)', flush=True)
del f_sess, f_aid, f_type, f_ts, ws, o_aid, o_ts, o_type, o_order, o_off, o_id

# ---- a6 ----
ev = DeviceEvents(d['aid'], d['ts'], d['type'], d['sess_off'], torch.arange(S, device=dev), None, OTTO_N_AIDS)
for strat in ('diff', 'time'):
    t, out = timed(lambda: build_aid_pairs_device(ev, strat, hour_difference=1, target_aggregation='mean', sample_frac=0.15, seed=42), reps=2)
    print(f'a6 build_aid_pairs_device({strat!r}): {1e3*t:.1f} ms  {out[0].numel()} labelled pairs  ({int(out[2].sum())} positive)', flush=True)
    del out

# ---- f4 ----
Sv = a.cand_sessions
off = (d['sess_off'][S - Sv:] - d['sess_off'][S - Sv]).contiguous()
e0 = int(d['sess_off'][S - Sv])
v_aid, v_type = d['aid'][e0:].contiguous(), d['type'][e0:].contiguous()
Cn = 100
g = torch.Generator(device=dev); g.manual_seed(1)
# candidate rows: the session's own aids first (what the generators put on top), then random aids; unique per row
cand = torch.randint(0, OTTO_N_AIDS, (Sv, Cn), device=dev, generator=g, dtype=torch.int32)
cand, _ = torch.sort(cand, dim=1)
dup = torch.zeros_like(cand, dtype=torch.bool); dup[:, 1:] = cand[:, 1:] == cand[:, :-1]
cand[dup] = -1
cand, _ = torch.sort(cand, dim=1, descending=True)
first = torch.minimum(off[:-1] + 0, off[1:] - 1)
cand[:, 0] = v_aid[first]                                            # one own aid per row (collisions with col>0 are negligible)
scores = torch.rand((Sv, Cn), device=dev, generator=g)
t, out = timed(lambda: interaction_features(v_aid, v_type, off, cand.contiguous(), scores, OTTO_N_AIDS))
rows = int((cand >= 0).sum())
print(f'f4 interaction_features: {1e3*t:.2f} ms  {Sv} sessions x {Cn} candidates = {rows} rows  {rows/t:.3e} rows/s', flush=True)
