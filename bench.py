#!/usr/bin/env python
"""bench.py -- headline benchmark of the hot path (BASELINE.json metric).

`python bench.py --gpus N --steps K --warmup W`. N > 1 runs one process per GPU: either the driver launches them
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`: WORLD_SIZE is set and this process is
one rank), or a plain `python bench.py --gpus N` starts them itself -- from a parent that never touches the GPU -- and
relays rank 0's JSON line.

A "step" is one full covisitation build over one batch of synthetic OTTO-shape
sessions that are already resident in HBM: pair-expand -> inverted index ->
LDS hash reduce -> top-20 per aid for the 3 type-weighted matrices
(BASELINE.json configs[1]).  value = deduped ordered aid-pairs expanded and
reduced per second, whole job (sum over ranks).  N > 1: every rank holds its own
`--sessions` sessions (weak scaling) and the expanded runs are exchanged by
aid_x owner over RCCL before the reduce (covisitation/distributed.py).

The JSON line also carries
  roofline     : the dominant kernel (by device time) against the HBM peak, from
                 HIP events recorded on the kernel's own stream inside the library
  roofline_expand : same for the pair-expand kernel (the kernel the north star's
                 40 % target names)
  cpu_baseline : the CPU oracle timed on a bounded sample of the same stream
  mf           : BPR-MF training throughput (second half of the metric), when built
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BENCH_KINDS = ('click_weighted', 'cart_weighted', 'order_weighted')


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--sessions', type=int, default=14_571_582, help='sessions per rank (full OTTO = 14,571,582)')
    ap.add_argument('--k', type=int, default=20)
    ap.add_argument('--cpu-sessions', type=int, default=4_000_000, help='sample size of the cpu_baseline leg (0 = skip)')
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak',
                    help='N > 1: weak = --sessions per rank (default); strong = ONE --sessions stream split by session chunk '
                         '(BASELINE.json config 4: full OTTO across the GPUs)')
    ap.add_argument('--no-mf', action='store_true')
    ap.add_argument('--no-dropin', action='store_true', help='skip the 7-kind drop-in build leg')
    ap.add_argument('--mf-rows', type=int, default=200_000_000)
    ap.add_argument('--mf-factors', type=int, default=64)
    return ap.parse_args()


def algorithmic_bytes(st, k, nk):
    """Algorithmic HBM bytes per launch of each kernel (DESIGN.md 'Measurement')."""
    S, Et, P = st['sessions'], st['tail_events'], st['pairs']
    # pair-expand: read aid 4 + ts 4 + type 1 per window event, 3 x 8 B per session (CSR offset + two slot bases); write one
    # (aid_x u32, descriptor u64) per window event and the records: with the component lists (k_expand_lists) one 4-byte
    # list word per run that reads a shared list + the records of the private rows; one 4-byte record per PAIR with the
    # round-2 layout ('expand_rows': the bytes SURVEY.md section 8 d prices, kept for comparison)
    words = st.get('shared_runs', 0) + st.get('row_records', 0)
    out = {
        'expand': 9 * Et + 24 * S + 4 * (words if st.get('shared_runs', 0) else P) + 12 * Et,
        'expand_rows': 9 * Et + 24 * S + 4 * P + 12 * Et,
    }
    # index: read (aid_x u32, descriptor u64) per window event once, write one descriptor per run grouped by aid_x (the
    # bucket split's intermediate copy and the second reads are overhead, not algorithmic)
    out['index'] = 12 * Et + 8 * st['runs']
    # partition: the heavy aids' records are read once and written once into buckets (the count pass and the
    # second read of the scatter pass are overhead, not algorithmic), plus their run descriptors
    out['partition'] = 8 * st['pairs_l'] + 8 * st['runs_l']
    for b, name in (('s', 'reduce_s'), ('m', 'reduce_m'), ('l', 'reduce_l')):
        aids = st[f'items_{b}'] if b != 'l' else 0
        # read every record (4 B) and run descriptor (8 B) of the bin once, 24 B of item/run_start
        # lookups per item, write nk top-k lists per aid (k x 12 B + 4 B)
        out[name] = 4 * st[f'pairs_{b}'] + 8 * st[f'runs_{b}'] + 24 * st[f'items_{b}'] + aids * nk * (12 * k + 4)
    return out


TRAFFIC_FILE = os.path.join(ROOT, 'profiles', 'round3', 'traffic.json')


SOURCES = {   # the files a kernel family is compiled from: its traffic numbers hold while THESE are unchanged
    'covis': ('csrc/otto_covis.hip', 'csrc/common.h', 'csrc/topk.h', 'csrc/scan.h', 'include/otto_covis.h'),
    'mf': ('csrc/otto_mf.hip', 'csrc/common.h', 'include/otto_mf.h'),
}


def source_hash(family=None):
    """sha256 over the kernel sources the traffic numbers of a kernel family belong to ({family: hash} without argument)."""
    import hashlib
    if family is None:
        return {f: source_hash(f) for f in SOURCES}
    h = hashlib.sha256()
    for rel in SOURCES[family]:
        path = os.path.join(ROOT, 'include', os.path.basename(rel)) if rel.startswith('include/') else \
            os.path.join(ROOT, 'otto-multi-objective-recommender-system_amd', rel)
        h.update(os.path.basename(rel).encode())
        h.update(open(path, 'rb').read())
    return h.hexdigest()


def pmc_traffic(names, full_otto):
    """HBM bytes per launch of the kernels `names` (template instantiations as the library reports them) from the committed
    PMC passes (tools/pmc_traffic.py under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs, calibrated:
    tools/pmc_summarize.py). PMC counters cannot be collected inside this run, so the file carries the hash of the kernel
    sources it was measured on: a different hash (any kernel edit since) or a different workload gives null, never stale
    numbers."""
    if not full_otto or not os.path.exists(TRAFFIC_FILE):
        return None
    try:
        doc = json.load(open(TRAFFIC_FILE))
    except Exception:
        return None
    family = 'mf' if any(n.startswith(('k_bpr', 'k_rmf', 'k_mf', 'k_score')) for n in names) else 'covis'
    if not isinstance(doc.get('source_hash'), dict) or doc['source_hash'].get(family) != source_hash(family):
        return None
    tot = sum(v['traffic_bytes_per_launch'] for k, v in doc['kernels'].items() if any(n in k for n in names))
    return int(tot) if tot else None


def roofline_obj(name, ms, nbytes, traffic=None):
    gbs = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    return {'kernel': name, 'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': round(gbs / HBM_PEAK_GBS, 4), 'traffic': traffic, 'avg_ms': round(ms, 4), 'algorithmic_bytes': int(nbytes)}


def candidates_leg(data, dev, n_aids, train_sessions=3_000_000, sessions=1_800_000):
    """Candidate lookup (click recipe, most_common(100)) and recency-weighted candidates over a validation-sized session
    set; the five matrices come from a build over the first ``train_sessions`` sessions of the bench stream."""
    import torch
    from otto_amd.covisitation.engine import CovisBuilder
    from otto_amd.covisitation import candidates as cd
    S = data['sess_off'].numel() - 1
    train_sessions, sessions = min(train_sessions, S), min(sessions, S)
    off = data['sess_off'][:train_sessions + 1].contiguous()
    e = int(off[-1])
    kinds = ('time_weighted', 'click_weighted', 'cart_weighted', 'click_cart', 'cart_order')
    b = CovisBuilder(n_aids, kinds=kinds, ts_min=int(data['ts'].min()), ts_max=int(data['ts'].max()), device=dev)
    b.feed(data['aid'][:e].contiguous(), data['ts'][:e].contiguous(), data['type'][:e].contiguous(), off)
    mats = b.finalize(k=15)
    del b
    lo = S - sessions                                            # the last sessions of the stream stand in for validation
    voff = (data['sess_off'][lo:] - data['sess_off'][lo]).contiguous()
    e0 = int(data['sess_off'][lo])
    vaid, vtyp = data['aid'][e0:].contiguous(), data['type'][e0:].contiguous()
    out = {}
    # algorithmic bytes of the lookup: every list row the recipe concatenates (4 B per entry: sum over the source aids of a
    # session of the lists' valid lengths, per term), the events (aid 4 + type 1), the [S, 100] candidate / count rows
    E = int(vaid.numel())
    sess = torch.repeat_interleave(torch.arange(sessions, device=dev), voff[1:] - voff[:-1], output_size=E)
    key = sess * (1 << 22) + vaid.long()
    src = {'U': torch.unique(key) % (1 << 22), 'CC': torch.unique(key[vtyp <= 1]) % (1 << 22)}
    entries = sum(int(mats[kind][2][src[s_]].sum().item()) for kind, s_ in cd.CLICK_RECIPE)
    lookup_bytes = 4 * entries + 5 * E + sessions * (100 * 8 + 4)
    for name, fn in (('lookup_click_recipe', lambda: cd.candidate_lookup(vaid, vtyp, voff, mats, cd.CLICK_RECIPE)),
                     ('recency', lambda: cd.recency_candidates(vaid, vtyp, voff))):
        fn()
        torch.cuda.synchronize(dev)
        t0 = time.time()
        for _ in range(3):
            fn()
        torch.cuda.synchronize(dev)
        ms = 1e3 * (time.time() - t0) / 3
        out[name] = {'ms': round(ms, 2), 'sessions_per_s': round(sessions / (ms * 1e-3), 1)}
        if name == 'lookup_click_recipe':
            out[name]['roofline'] = {**roofline_obj('k_cand<32, 10, 128> + k_cand<512, 12, 512> (host wall time of both launches)', ms, lookup_bytes),
                                     'list_entries': entries,
                                     'note': 'one workgroup per session, LDS hash + selection: latency bound, not an HBM stream'}
    out['config'] = f'{sessions} sessions, {int(vaid.numel())} events, top-15 matrices from {train_sessions} sessions'
    return out


def next_rows_leg(data, dev, n_aids, cand_sessions=1_800_000):
    """The rows SURVEY.md section 8 marks "next", device-resident inputs, each with a roofline object (algorithmic bytes =
    inputs read once + outputs written once; the radix sort and the pair sorts move several times that by construction):
    f2 event ingest (shuffled full-size frame, ms timestamps -> sorted SoA + CSR), a6 aid-pair builders, f4 interaction
    features over a validation-sized [S, 100] candidate table."""
    import ctypes as C
    import torch
    from otto_amd import _lib
    from otto_amd.events import DeviceEvents
    from otto_amd.matrix_factorization.data import build_aid_pairs_device
    from otto_amd.ranker.interaction_feature_engineering import interaction_features
    lib = _lib.lib()
    p = lambda t: C.c_void_p(t.data_ptr())
    stream = lambda: C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    S, E = data['sess_off'].numel() - 1, data['aid'].numel()

    def timed(fn, reps=2):
        best, out = 1e30, None
        for _ in range(reps):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            out = fn()
            torch.cuda.synchronize(dev)
            best = min(best, time.perf_counter() - t0)
        return 1e3 * best, out
    res = {}
    sess = torch.repeat_interleave(torch.arange(S, device=dev, dtype=torch.int32), data['sess_off'][1:] - data['sess_off'][:-1], output_size=E)
    perm = torch.randperm(E, device=dev)
    f_sess, f_aid, f_type = sess[perm].contiguous(), data['aid'][perm].contiguous(), data['type'][perm].contiguous()
    f_ts = (data['ts'][perm].to(torch.int64) * 1000 + 7).contiguous()
    del perm, sess
    ws_b = int(lib.otto_events_sort_workspace(E))
    ws = torch.empty(ws_b, dtype=torch.uint8, device=dev)
    o_aid, o_ts = torch.empty(E, dtype=torch.int32, device=dev), torch.empty(E, dtype=torch.int32, device=dev)
    o_type, o_order = torch.empty(E, dtype=torch.uint8, device=dev), torch.empty(E, dtype=torch.int32, device=dev)
    o_off, o_id = torch.empty(E + 1, dtype=torch.int64, device=dev), torch.empty(E, dtype=torch.int32, device=dev)
    ns = C.c_int64()

    def sort_events():
        _lib.check(lib.otto_events_sort(p(f_sess), p(f_ts), p(f_aid), p(f_type), E, 1000, p(o_aid), p(o_ts), p(o_type), p(o_order), p(o_off),
                                        p(o_id), C.byref(ns), p(ws), ws_b, stream()), 'otto_events_sort')
    ms, _ = timed(sort_events, 3)
    ok = ns.value == S and bool((o_ts == data['ts']).all())
    res['f2_events_sort'] = {'ms': round(ms, 2), 'events_per_s': round(E / (ms * 1e-3), 1), 'sorted_correctly': ok,
                             'roofline': roofline_obj('otto_events_sort: k_rs_hist + k_rs_scatter x digit passes + gather (host wall time)', ms,
                                                      E * (17 + 13) + S * 12)}
    del f_sess, f_aid, f_type, f_ts, ws, o_aid, o_ts, o_type, o_order, o_off, o_id
    ev = DeviceEvents(data['aid'], data['ts'], data['type'], data['sess_off'], torch.arange(S, device=dev), None, n_aids)
    for strat in ('diff', 'time'):
        ms, out = timed(lambda: build_aid_pairs_device(ev, strat, hour_difference=1, target_aggregation='mean', sample_frac=0.15, seed=42))
        rows = int(out[0].numel())
        read = E * 8 + S * 8 if strat == 'diff' else int(0.15 * E) * 8 + S * 8
        res[f'a6_pairs_{strat}'] = {'ms': round(ms, 1), 'labelled_pairs': rows,
                                    'roofline': roofline_obj(f'otto_pairs_{strat} (emit + radix sort by pair + aggregate; host wall time incl. workspace allocation)',
                                                             ms, read + rows * 24)}
        del out
    Sv = min(cand_sessions, S)
    off = (data['sess_off'][S - Sv:] - data['sess_off'][S - Sv]).contiguous()
    e0 = int(data['sess_off'][S - Sv])
    v_aid, v_type = data['aid'][e0:].contiguous(), data['type'][e0:].contiguous()
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    cand = torch.randint(0, n_aids, (Sv, 100), device=dev, generator=g, dtype=torch.int32)
    cand, _ = torch.sort(cand, dim=1)
    dup = torch.zeros_like(cand, dtype=torch.bool)
    dup[:, 1:] = cand[:, 1:] == cand[:, :-1]
    cand[dup] = -1
    cand, _ = torch.sort(cand, dim=1, descending=True)
    cand[:, 0] = v_aid[torch.minimum(off[:-1] + 0, off[1:] - 1)]
    cand = cand.contiguous()
    scores = torch.rand((Sv, 100), device=dev, generator=g)
    ms, _ = timed(lambda: interaction_features(v_aid, v_type, off, cand, scores, n_aids), 3)
    rows = int((cand >= 0).sum())
    res['f4_interaction_features'] = {'ms': round(ms, 2), 'rows': rows, 'rows_per_s': round(rows / (ms * 1e-3), 1),
                                      'roofline': roofline_obj('k_inter_rows + k_inter_aids (host wall time)', ms,
                                                               Sv * 100 * (4 + 4 + 10) + int(v_aid.numel()) * 5 + Sv * 40 + n_aids * 36)}
    return res


def dropin_leg(data, dev, n_aids, ts_min, ts_max, k, reps=3):
    """What `covisitation/builder.py <mode>` runs on the device for the reference's consumers: the 7 matrix kinds
    (time_weighted, three type-weighted, three filter kinds) of one event stream, top-k each: class-sorted pair-expand
    (filter kinds), time channel, three reduce groups. Device-resident inputs, like the headline value."""
    import torch
    from otto_amd.covisitation.engine import CovisBuilder
    from otto_amd.covisitation.spec import REFERENCE_KINDS
    b = CovisBuilder(n_aids, kinds=REFERENCE_KINDS, ts_min=ts_min, ts_max=ts_max, device=dev)
    tm = {}

    def step():
        b.reset()
        b.feed(data['aid'], data['ts'], data['type'], data['sess_off'])
        return b.finalize(k=k)
    step()
    torch.cuda.synchronize(dev)
    each = []
    for _ in range(reps):
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize(dev)
        each.append(1e3 * (time.perf_counter() - t0))
    ms = sum(each) / len(each)
    st = b.stats()
    return {'ms_per_build': round(ms, 2), 'ms_each_build': [round(v, 2) for v in each], 'aid_pairs_per_s': round(st['pairs'] / (ms * 1e-3), 1), 'kinds': list(REFERENCE_KINDS), 'k': k,
            'note': 'pairs counted once (the all-ones expansion feeds every kind); kernel_ms of the last reduce group below',
            'kernel_ms_last_group': {n: round(v, 3) for n, v in b.timings().items()}, 'kernels': b.kernel_names()}


def cpu_baseline(dev_data, n_sessions, k):
    """Time the CPU restatement (oracle/, kind 'port') on the first n_sessions of the same stream: the best of a few
    thread counts (an all-core run can lose to a quarter of the cores on a 256-thread host; every run is reported)."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import numpy as np
    off = dev_data['sess_off'][:n_sessions + 1].cpu().numpy()
    e = int(off[-1])
    aid = dev_data['aid'][:e].cpu().numpy().astype(np.uint32)
    ts = dev_data['ts'][:e].cpu().numpy()
    typ = dev_data['type'][:e].cpu().numpy()
    try:
        import covis_oracle_c as coc
    except Exception:
        coc = None
    if coc is not None and coc.available():
        host = os.cpu_count() or 1
        tried = {}
        for cores_try in sorted({min(host, 64), min(host, 128), host}):
            r = coc.covis_topk_c(aid, ts, typ, off, dev_data['n_aids'], BENCH_KINDS, k=k, threads=cores_try, rows=False)
            tried[cores_try] = (r['P'], r['seconds_in_c'])
        cores = min(tried, key=lambda c_: tried[c_][1])
        pairs, dt = tried[cores]
        impl = (f'oracle/covis_oracle.c (gcc -O3, OpenMP), best of threads '
                + ', '.join(f'{c_}: {tried[c_][0] / tried[c_][1]:.3g} pairs/s' for c_ in sorted(tried)))
    else:
        import covis_oracle as co
        cores = 1
        st = {}
        t0 = time.time()
        co.covis_topk_numpy(aid, ts, typ, off, co.CovisSpec(kinds=BENCH_KINDS), k=k, stats=st)
        dt = time.time() - t0
        pairs = st['P']
        impl = 'oracle/covis_oracle.py (NumPy)'
    return {'value': round(pairs / dt, 1), 'unit': 'aid-pairs/s', 'cores': cores, 'kind': 'port',
            'sample': f'first {n_sessions} sessions of the same synthetic stream ({e} events, {pairs} pairs), '
                      f'{impl}, {dt:.1f} s wall', 'host_cores_available': os.cpu_count()}


def mf_cpu_baseline(U, V, users, items, n_items, sample):
    """PyTorch-CPU restatement of the BPR batch step (oracle/mf_oracle.py arithmetic) on a bounded sample, at the best of
    a few thread counts (a 256-thread host loses to 8 - 32 threads on these memory-bound passes; every count tried is
    reported)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import mf_oracle as mo
    host = os.cpu_count() or 1
    u = users[:sample].cpu()
    i = items[:sample].cpu()
    uu, inv = torch.unique(u, return_inverse=True)          # only the touched user rows travel to the host
    Uc = U[uu.to(U.device)].cpu()
    Vc = V.cpu()
    t0 = time.time()
    j = torch.from_numpy(mo.bpr_negatives(42, 0, 0, i.numpy()[:20000], n_items))       # sampler rate measured on 20k rows
    t_neg = (time.time() - t0) / 20000 * sample
    j = torch.randint(0, n_items, (sample,))

    def step(n):
        t0 = time.time()
        eu, ei, ej = Uc[inv[:n]], Vc[i[:n]], Vc[j[:n]]
        x = (eu * (ei - ej)).sum(1)
        s = torch.sigmoid(-x)[:, None]
        Uc.index_add_(0, inv[:n], 0.05 * s * (ei - ej))
        Vc.index_add_(0, i[:n], 0.05 * s * eu)
        Vc.index_add_(0, j[:n], -0.05 * s * eu)
        return time.time() - t0
    probe = {}
    for th in sorted({min(host, 8), min(host, 32), min(host, 128), host}):
        torch.set_num_threads(th)
        probe[th] = step(sample // 4)
        if probe[th] > 3 * min(probe.values()):
            break
    cores = min(probe, key=probe.get)
    torch.set_num_threads(cores)
    dt = step(sample)
    return {'value': round(sample / dt, 1), 'unit': 'triplets/s', 'cores': cores, 'kind': 'port', 'host_cores_available': host,
            'sample': f'{sample} triplets of the same stream, PyTorch-CPU gather/dot/sigmoid/index_add_ at {cores} threads '
                      f'({dt:.2f} s; probe on {sample // 4} triplets, s by threads: ' + ', '.join(f'{k}: {v:.2f}' for k, v in sorted(probe.items()))
                      + f'; pure-Python counter RNG of the oracle excluded: {t_neg:.1f} s extrapolated)'}



def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks with torch.distributed.run (one process per GPU,
    rendezvous on 127.0.0.1) from THIS process, which has made no GPU call (torch is not even imported here), pass
    their output through and return the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    from otto_amd.synth import generate_sessions_torch, OTTO_N_AIDS
    from otto_amd.covisitation.engine import CovisBuilder
    from otto_amd.covisitation.distributed import ShardedCovisBuilder, global_ts_range
    from otto_amd import _lib

    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(self_launch(a.gpus))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != a.gpus:
        raise SystemExit(f'--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}')
    # rehearsal hooks (not used by the driver): OTTO_BENCH_BACKEND=gloo stages the exchange through host
    # memory and OTTO_BENCH_DEVICE pins every rank to one GPU, so the N > 1 code path can be exercised on a
    # one-GPU box. The real run is nccl (= RCCL), one GPU per rank.
    backend = os.environ.get('OTTO_BENCH_BACKEND', 'nccl')
    stage = 'cpu' if backend == 'gloo' else None
    dev = torch.device(f"cuda:{os.environ.get('OTTO_BENCH_DEVICE', local_rank)}")
    torch.cuda.set_device(dev)
    if world > 1:
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    if world > 1 and a.scaling == 'strong':
        # ONE stream of --sessions sessions (same seed on every rank), rank r expands the session chunk [S r/W, S (r+1)/W)
        full = generate_sessions_torch(a.sessions, n_aids=OTTO_N_AIDS, seed=42, device=dev)
        s_lo, s_hi = a.sessions * rank // world, a.sessions * (rank + 1) // world
        e_lo, e_hi = int(full['sess_off'][s_lo]), int(full['sess_off'][s_hi])
        data = {'aid': full['aid'][e_lo:e_hi].contiguous(), 'ts': full['ts'][e_lo:e_hi].contiguous(),
                'type': full['type'][e_lo:e_hi].contiguous(), 'sess_off': (full['sess_off'][s_lo:s_hi + 1] - e_lo).contiguous(),
                'n_aids': full['n_aids']}
        del full
        torch.cuda.empty_cache()
    else:
        # weak scaling: every rank draws its own sessions (seed 42 + rank) from ONE aid popularity table (rank 0's), as the
        # shards of one stream would -- rank 0 of N ranks is exactly the N = 1 workload
        data = generate_sessions_torch(a.sessions, n_aids=OTTO_N_AIDS, seed=42 + rank, device=dev, pop_seed=42)
    n_aids = data['n_aids']
    if world > 1:
        ts_min, ts_max = global_ts_range(data['ts'].cpu() if stage else data['ts'])
        builder = ShardedCovisBuilder(n_aids, BENCH_KINDS, ts_min, ts_max, dev, stage_device=stage)
        eng = builder.owner
    else:
        ts_min, ts_max = int(data['ts'].min()), int(data['ts'].max())
        builder = CovisBuilder(n_aids, kinds=BENCH_KINDS, ts_min=ts_min, ts_max=ts_max, device=dev)
        eng = builder
    nk = len(BENCH_KINDS)
    out = {_lib.GROUP_TYPE: (torch.empty((nk, n_aids, a.k), dtype=torch.int32, device=dev),
                             torch.empty((nk, n_aids, a.k), dtype=torch.int64, device=dev),
                             torch.empty((nk, n_aids), dtype=torch.int32, device=dev))}

    def step():
        builder.reset()
        builder.feed(data['aid'], data['ts'], data['type'], data['sess_off'])
        builder.finalize(k=a.k, out=out)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    kernel_ms = {}
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
        for src in ((builder.local, eng) if world > 1 else (eng,)):   # finalize() has already synchronised
            for name, ms in src.timings().items():
                kernel_ms[name] = kernel_ms.get(name, 0.0) + ms
    barrier()
    dt = time.perf_counter() - t0
    st = eng.stats()
    st_expand = builder.local.stats() if world > 1 else st
    pairs = st['pairs']
    if world > 1:
        red = torch.tensor([dt, float(pairs)], dtype=torch.float64, device='cpu' if stage else dev)
        mx = red.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = red.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt, pairs = float(mx[0].item()), int(sm[1].item())

    result = None
    if rank == 0:
        kernel_ms = {k_: v / a.steps for k_, v in kernel_ms.items()}
        if world > 1:   # the expand kernel ran on the local engine; its P = records this rank exported
            st_k = {**st, 'sessions': st_expand['sessions'], 'tail_events': st_expand['tail_events'],
                    'pairs': builder.last_exchange[1]}
        else:
            st_k = st
        ab = algorithmic_bytes(st_k, a.k, nk)
        cand = {n: kernel_ms.get(n, 0.0) for n in ('expand', 'index', 'partition', 'reduce_s', 'reduce_m', 'reduce_l')}
        dom = max(cand, key=cand.get)
        full_otto = world == 1 and a.sessions == 14_571_582 and a.k == 20
        # kernel names as the library reports them for the launches of the last step (what rocprofv3 lists)
        knames = eng.kernel_names()
        if world > 1:
            knames['expand'] = builder.local.kernel_names()['expand']
        # the traffic file is keyed by rocprofv3's kernel names: exact instantiations where a slot is one template, prefixes
        # for the index / partition slots (several small kernels)
        ksub = {slot: tuple(n for n in names.split(' + ') if '<' in n and '/' not in n and '*' not in n) for slot, names in knames.items()}
        ksub['index'] = ('k_bkt_split', 'k_bkt_fused')
        ksub['partition'] = ('k_partition',)
        result = {
            'metric': 'aid-pairs/sec covisitation build',
            'value': round(pairs * a.steps / dt, 1),
            'unit': 'aid-pairs/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': round(1e3 * dt / a.steps, 3),
            'higher_is_better': True,
            'scaling': a.scaling if world > 1 else 'weak',
            'vs_baseline': None,
            'dtype': 'u64',
            'data': 'synthetic',
            'config': {
                'workload': f'full-OTTO-shape covisitation: {a.sessions} synthetic sessions {"per GPU" if (world == 1 or a.scaling == "weak") else "in all, split by session chunk over the GPUs"}, '
                            f'{data["aid"].numel()} events, {n_aids} aids, window=30, max_gap=86400 s, '
                            f'3 type-weighted matrices (click/cart/order_weighted), top-{a.k}/aid',
                'sessions_per_gpu': int(data['sess_off'].numel() - 1), 'events_per_gpu': int(data['aid'].numel()),
                'pairs_total': int(pairs), 'kinds': list(BENCH_KINDS), 'k': a.k,
                'parallelism': 'single GPU' if world == 1 else f'session-chunk x{world}, RCCL all-to-all-v of expanded runs by aid_x owner',
            },
            'roofline': roofline_obj(knames[dom], cand[dom], ab[dom], pmc_traffic(ksub[dom], full_otto)),
            'roofline_expand': {**roofline_obj(knames['expand'], cand['expand'], ab['expand'], pmc_traffic(ksub['expand'], full_otto)),
                                'frac_on_round2_bytes': round(ab['expand_rows'] / (cand['expand'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if cand['expand'] > 0 else None,
                                'round2_algorithmic_bytes': int(ab['expand_rows']),
                                'note': 'algorithmic_bytes = what the component-list layout moves (one list word per run + private rows); '
                                        'frac_on_round2_bytes prices the same time on the one-record-per-pair bytes of SURVEY.md section 8 d'},
            'kernel_ms': {k_: round(v, 3) for k_, v in kernel_ms.items()},
            'stats': st,
        }
    # ---- the real drop-in workload (covisitation/builder.py:build_matrices): all 7 reference kinds, k = 20, N = 1 ------
    if rank == 0 and world == 1 and not a.no_dropin:
        try:
            result['dropin_7kinds'] = dropin_leg(data, dev, n_aids, ts_min, ts_max, a.k)
        except Exception as e:                                   # a reported extra, never a reason to lose the bench line
            result['dropin_7kinds'] = {'error': repr(e)}
    # ---- second half of the metric: BPR-MF triplets/s ----------------------------------------------
    if not a.no_mf:
        import bench_mf
        if bench_mf is not None:
            del builder, eng, out
            torch.cuda.empty_cache()
            mf = bench_mf.run(a, dev, rank, world, mf_cpu_baseline if (rank == 0 and world == 1 and a.cpu_sessions > 0) else None,
                              (lambda subs: pmc_traffic(subs, world == 1 and a.sessions == 14_571_582)))
            if rank == 0:
                result['mf'] = mf
    # ---- the callers either side of the path, N = 1 only: candidate lookup + recency ----
    if rank == 0 and world == 1 and not a.no_mf:
        try:
            result['candidates'] = candidates_leg(data, dev, n_aids)
        except Exception as e:                                   # a reported extra, never a reason to lose the bench line
            result['candidates'] = {'error': repr(e)}
    if rank == 0 and world == 1 and not a.no_mf:
        try:
            result['next_rows'] = next_rows_leg(data, dev, n_aids)
        except Exception as e:                                   # a reported extra, never a reason to lose the bench line
            result['next_rows'] = {'error': repr(e)}
    if rank == 0 and world == 1 and a.cpu_sessions > 0:
        result['cpu_baseline'] = cpu_baseline(data, min(a.cpu_sessions, a.sessions), a.k)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
