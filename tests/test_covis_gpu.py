"""GPU parity tests of the covisitation builder: HIP path (through the C-ABI) vs the
CPU oracle (oracle/covis_oracle.py) on the same seeded inputs. Integer work: bit-exact."""
import itertools

import numpy as np
import pytest

import covis_oracle as co
from otto_amd.synth import generate_sessions, Events
from otto_amd.covisitation import spec as cs

pytestmark = pytest.mark.gpu

# kind sets: with filter kinds K1 runs the class-sorted M-matrix kernels, without them the fused register kernel
NOFILT = ('time_weighted', 'click_weighted', 'cart_weighted', 'order_weighted')


def _to_dev(ev, dev):
    import torch
    return (torch.from_numpy(ev.aid.astype(np.int32)).to(dev), torch.from_numpy(ev.ts).to(dev),
            torch.from_numpy(ev.type).to(dev), torch.from_numpy(ev.sess_off).to(dev))


def _build(ev, dev, kinds=cs.ALL_KINDS, k=20, window=30, max_gap=86400, chunks=1, l_cap=None, options=None):
    from otto_amd.covisitation.engine import CovisBuilder, topk_to_rows
    ts_min, ts_max = (int(ev.ts.min()), int(ev.ts.max())) if ev.n_events else (0, 0)
    b = CovisBuilder(ev.n_aids, kinds=kinds, window=window, max_gap=max_gap, ts_min=ts_min, ts_max=ts_max, device=dev)
    if l_cap is not None:
        b.set_option('l_cap', l_cap)
    for name, value in (options or {}).items():
        b.set_option(name, value)
    aid, ts, typ, off = _to_dev(ev, dev)
    S = ev.n_sessions
    bounds = np.linspace(0, S, chunks + 1).astype(np.int64)
    for c in range(chunks):
        lo, hi = int(bounds[c]), int(bounds[c + 1])
        e0, e1 = int(ev.sess_off[lo]), int(ev.sess_off[hi])
        sub_off = (off[lo:hi + 1] - e0).contiguous()
        b.feed(aid[e0:e1].contiguous(), ts[e0:e1].contiguous(), typ[e0:e1].contiguous(), sub_off)
    out = b.finalize(k=k)
    rows = {kind: topk_to_rows(*out[kind]) for kind in kinds}
    return b, rows


def _oracle_rows(ev, kinds, k=20, window=30, max_gap=86400, stats=None):
    sp = co.CovisSpec(window=window, max_gap=max_gap, kinds=tuple(kinds))
    return co.covis_topk_numpy(ev.aid, ev.ts, ev.type, ev.sess_off, sp, k=k, stats=stats)


def _assert_rows_equal(got, want, kinds):
    for kind in kinds:
        gx, gy, gw = got[kind]
        wx, wy, ww = want[kind]
        assert len(gx) == len(wx), f'{kind}: {len(gx)} rows vs oracle {len(wx)}'
        assert np.array_equal(gx, wx), f'{kind}: aid_x differs'
        assert np.array_equal(gw, ww), f'{kind}: W differs'
        assert np.array_equal(gy, wy), f'{kind}: aid_y differs'


def test_product_spec_matches_oracle_spec():
    assert cs.TYPE_WEIGHTS == co.TYPE_WEIGHTS and cs.FILTER_MASKS == co.FILTER_MASKS
    assert cs.ALL_KINDS == co.ALL_KINDS and cs.Q16 == co.Q16


@pytest.mark.parametrize('kinds,options', [(cs.ALL_KINDS, None), (('time_weighted', 'click_weighted'), None), (('cart_weighted',), None),
                                           (('time_weighted', 'click_weighted'), {'fused': 1}), (('cart_weighted',), {'fused': 1})],
                         ids=['filters-class-kernels', 'lists-time', 'lists', 'fused-rows-time', 'fused-rows'])
def test_pair_expand_records_match_oracle(gpu_device, kinds, options):
    """K1 alone: every window's records equal the oracle's per-window expansion. A run is (aid_x, a list of records);
    a run descriptor with sp != 63 points at a SHARED component list that holds the run's own aid at position sp (not a
    pair; its slot of the time channel carries the run's time extra). The records of a window are compared as a set of
    (x, y, type_y, filter bits, time extra): each ordered pair at most once, whatever runs it is spread over."""
    ev = generate_sessions(1500, n_aids=400, seed=3)
    b, _ = _build(ev, gpu_device, kinds=kinds, options=options)
    rec, tw, run_x, run_desc = b.copy_records()
    sp = co.CovisSpec()
    t0, t1 = int(ev.ts.min()), int(ev.ts.max())
    fk = b.filter_kinds
    L = np.diff(ev.sess_off)
    nwin = np.where(np.minimum(L, 30) >= 2, np.minimum(L, 30), 0)
    ev_base = np.r_[0, np.cumsum(nwin)]
    pair_base = np.r_[0, np.cumsum(nwin * (nwin - 1))]
    lists = len(rec) == pair_base[-1] + ev_base[-1]          # component lists: n list slots per window behind the pair slots
    assert b.stats()['tail_events'] == ev_base[-1] and b.stats()['pair_slots'] == pair_base[-1] + (ev_base[-1] if lists else 0)
    total = shared_runs = 0
    for s in range(ev.n_sessions):
        want = co.expand_window_python(ev.aid, ev.ts, ev.type, int(ev.sess_off[s]), int(ev.sess_off[s + 1]), sp, fk, t0, t1)
        got = []
        used = set()
        for r in range(int(ev_base[s]), int(ev_base[s + 1])):
            d = int(run_desc[r])
            ln, off, spos = d & 0x3F, (d >> 8) & ((1 << 40) - 1), (d >> 48) & 63
            if not ln:
                assert d == 0 and run_x[r] == 0xFFFFFFFF
                continue
            x = int(run_x[r])
            if spos != 63:
                assert lists and pair_base[-1] + ev_base[s] <= off and off + ln <= pair_base[-1] + ev_base[s + 1]
                assert spos < ln and int(rec[off + spos]) & 0x3FFFFFF == x
                shared_runs += 1
            else:
                assert pair_base[s] <= off and off + ln <= pair_base[s + 1]
            for t in range(ln):
                if t == spos:
                    continue
                rc = int(rec[off + t])
                if spos == 63:
                    assert off + t not in used
                    used.add(off + t)
                e = 0 if tw is None else int(tw[off + (spos if spos != 63 else t)])
                got.append((x, rc & 0x3FFFFFF, (rc >> 26) & 3, rc >> 28, e))
        if tw is None:
            want = [w[:4] + (0,) for w in want]
        assert len(set(g[:2] for g in got)) == len(got), f'session {s}: a pair occurs twice'
        assert sorted(got) == sorted(want), f'session {s}'
        total += len(want)
    assert b.stats()['pairs'] == total
    assert (shared_runs > 0) == lists


@pytest.mark.parametrize('kinds', [cs.ALL_KINDS, NOFILT], ids=['all-kinds', 'fused'])
@pytest.mark.parametrize('n_sessions,n_aids,seed', [(3000, 2000, 11), (20000, 60, 12), (800, 1855603, 13)])
def test_topk_all_kinds_match_oracle(gpu_device, n_sessions, n_aids, seed, kinds):
    """Full pipeline, k=20: small/medium/heavy aids (n_aids=60 makes every aid heavy); all 8 kinds (class-sorted
    K1) and the four kinds without a filter mask (fused K1)."""
    ev = generate_sessions(n_sessions, n_aids=n_aids, seed=seed)
    st = {}
    want = _oracle_rows(ev, kinds, stats=st)
    b, got = _build(ev, gpu_device, kinds=kinds)
    _assert_rows_equal(got, want, kinds)
    assert b.stats()['pairs'] == st['P']


def _hub_events(n_hub_sessions, per_session, repeated, n_aids, hub_type=0):
    """Sessions `hub, a, b, c, ...` (clicks, fresh aids every session, seconds apart) plus `repeated` partners that meet
    the hub in two sessions each: the hub has n_hub_sessions * per_session keys of ONE click record and `repeated` keys of
    two records."""
    aid, typ, off = [], [], [0]
    nxt = 1 + repeated
    for s in range(n_hub_sessions):
        row = [0] + list(range(nxt, nxt + per_session))
        nxt += per_session
        if s < 2 * repeated:
            row.append(1 + s // 2)
        aid += row
        typ += [hub_type] + [0] * (len(row) - 1)
        off.append(len(aid))
    assert nxt <= n_aids
    ts = (1_660_000_000 + np.arange(len(aid))).astype(np.int32)
    return Events(aid=np.array(aid, dtype=np.uint32), ts=ts, type=np.array(typ, dtype=np.uint8),
                  sess_off=np.array(off, dtype=np.int64), n_aids=n_aids)


@pytest.mark.parametrize('n_hub,per,repeated', [(40, 28, 5), (40, 28, 25), (300, 28, 7), (300, 28, 40)],
                         ids=['M-5-heavy', 'M-25-heavy', 'L-7-heavy', 'L-40-heavy'])
def test_topk_walks_with_fewer_heavy_keys_than_k(gpu_device, n_hub, per, repeated):
    """The multi-wave bins rank the keys of more than one click record first and fall back to every key when fewer
    than k of them exist: a hub aid of the M bin (1,120 records) / the L bin (8,400 records: partitions, merge) whose
    partners are almost all single clicks, with fewer and with more than k = 20 repeated partners; the hub's light
    keys tie on weight, so the rows below the repeated partners are the smallest aid_y."""
    ev = _hub_events(n_hub, per, repeated, n_aids=1 + repeated + n_hub * per + 5)
    kinds = ('click_weighted', 'cart_weighted', 'order_weighted')
    want = _oracle_rows(ev, kinds)
    for options in (None, {'hot': 0}):
        b, got = _build(ev, gpu_device, kinds=kinds, options=options)
        _assert_rows_equal(got, want, kinds)
    st = b.stats()
    assert st['items_m'] + st['items_l'] >= 1


def test_topk_type_weights_that_rank_a_single_click_above_other_keys(gpu_device, monkeypatch):
    """Type weights (5, 1, 2): one click record outweighs two cart records, so "more than one click record" says
    nothing about the rank of a key -- the heavy-first walks must switch themselves off (`hot_ok` on the host)."""
    w = {'click_weighted': (5, 1, 2), 'cart_weighted': (1, 9, 6), 'order_weighted': (3, 3, 1)}
    for mod in (cs, co):
        monkeypatch.setattr(mod, 'TYPE_WEIGHTS', {**mod.TYPE_WEIGHTS, **w})
    import otto_amd.covisitation.engine as eng
    monkeypatch.setattr(eng, 'TYPE_WEIGHTS', cs.TYPE_WEIGHTS)
    kinds = ('click_weighted', 'cart_weighted', 'order_weighted')
    ev = generate_sessions(20000, n_aids=60, seed=21)
    want = _oracle_rows(ev, kinds)
    _, got = _build(ev, gpu_device, kinds=kinds)
    _assert_rows_equal(got, want, kinds)


@pytest.mark.parametrize('options', [{'fused': 1}, {'fused': 1, 'fast_path': 0}, {'fused': 0}, {'fused': 0, 'fast_path': 0}],
                         ids=['fused-rows', 'fused-rows-general-only', 'class-kernels', 'class-kernels-general-only'])
def test_expand_variants_agree(gpu_device, options):
    """The fused K1 with its gap-free shortcut switched off (every window through the general row loop), and the
    class-sorted kernels forced for the same kinds, give the oracle's rows too."""
    ev = generate_sessions(4000, n_aids=700, seed=17)
    want = _oracle_rows(ev, NOFILT)
    _, got = _build(ev, gpu_device, kinds=NOFILT, options=options, chunks=2)
    _assert_rows_equal(got, want, NOFILT)


def test_heavy_aid_partitions_and_overflow_retry(gpu_device):
    """One aid co-occurs with ~30k distinct aids: exercises hash partitions (R > 1), the partial
    top-k merge and -- with l_cap raised so partitions overflow the LDS table -- the re-partition rounds."""
    rng = np.random.default_rng(5)
    S, n_aids = 4000, 40000
    aid = rng.integers(1, n_aids, size=(S, 30)).astype(np.uint32)
    aid[:, rng.integers(0, 30, S)[0]] = 0
    aid[np.arange(S), rng.integers(0, 30, S)] = 0
    ts = (1_660_000_000 + np.cumsum(rng.integers(1, 50, size=(S, 30)), axis=1)).astype(np.int32)
    typ = rng.integers(0, 3, size=(S, 30)).astype(np.uint8)
    ev = Events(aid=aid.ravel(), ts=ts.ravel(), type=typ.ravel(), sess_off=np.arange(S + 1, dtype=np.int64) * 30, n_aids=n_aids)
    kinds = ('click_weighted', 'cart_weighted', 'order_weighted', 'time_weighted', 'click_cart', 'cart_order')
    want = _oracle_rows(ev, kinds)
    b, got = _build(ev, gpu_device, kinds=kinds)
    assert b.stats()['items_l'] > 1 and b.stats()['retries'] == 0
    _assert_rows_equal(got, want, kinds)
    b2, got2 = _build(ev, gpu_device, kinds=kinds, l_cap=200000)
    assert b2.stats()['retries'] >= 1, 'l_cap=200000 should overflow the 8192-slot table and re-partition'
    _assert_rows_equal(got2, want, kinds)


@pytest.mark.parametrize('l_cap', [64, 256, 1024])
def test_partition_scatter_paths_by_partition_count(gpu_device, l_cap):
    """A hub aid of ~58 k pairs cut into partitions of l_cap pairs: l_cap = 64 -> 1024 partitions, above what the partition pass
    stages in LDS (512: its direct scatter path with one global cursor bump per record); 256 -> 256 partitions and 1024 -> 64, the
    staged path with many / few partitions per chunk. With and without the time channel travelling along."""
    rng = np.random.default_rng(11)
    S, n_aids = 2000, 30000
    aid = rng.integers(1, n_aids, size=(S, 30)).astype(np.uint32)
    aid[np.arange(S), rng.integers(0, 30, S)] = 0
    ts = (1_660_000_000 + np.cumsum(rng.integers(1, 50, size=(S, 30)), axis=1)).astype(np.int32)
    typ = rng.integers(0, 3, size=(S, 30)).astype(np.uint8)
    ev = Events(aid=aid.ravel(), ts=ts.ravel(), type=typ.ravel(), sess_off=np.arange(S + 1, dtype=np.int64) * 30, n_aids=n_aids)
    for kinds in (('click_weighted', 'cart_weighted', 'order_weighted'), ('time_weighted', 'click_weighted')):
        want = _oracle_rows(ev, kinds)
        b, got = _build(ev, gpu_device, kinds=kinds, l_cap=l_cap)
        assert b.stats()['items_l'] >= 58000 // (2 * l_cap)
        _assert_rows_equal(got, want, kinds)


def test_context_reuse_across_streams_and_groups(gpu_device):
    """One context, two different event streams one after the other (reset between them), kinds from all three reduce groups, hub
    aids that are partitioned: the partition buckets are filled once per heavy item list and reused by the later passes and groups
    -- they must not survive into the next stream's build, nor a second finalize with another k."""
    from otto_amd.covisitation.engine import CovisBuilder, topk_to_rows
    kinds = ('click_weighted', 'time_weighted', 'click_cart', 'cart_weighted')
    rng = np.random.default_rng(31)

    def hub_stream(S, n_aids, hub):
        aid = rng.integers(1, n_aids, size=(S, 30)).astype(np.uint32)
        aid[np.arange(S), rng.integers(0, 30, S)] = hub
        ts = (1_660_000_000 + np.cumsum(rng.integers(1, 50, size=(S, 30)), axis=1)).astype(np.int32)
        typ = rng.integers(0, 3, size=(S, 30)).astype(np.uint8)
        return Events(aid=aid.ravel(), ts=ts.ravel(), type=typ.ravel(), sess_off=np.arange(S + 1, dtype=np.int64) * 30, n_aids=n_aids)
    ev_a, ev_b = hub_stream(1500, 20000, 0), hub_stream(1100, 20000, 7)
    ts_min = int(min(ev_a.ts.min(), ev_b.ts.min()))
    ts_max = int(max(ev_a.ts.max(), ev_b.ts.max()))
    b = CovisBuilder(20000, kinds=kinds, ts_min=ts_min, ts_max=ts_max, device=gpu_device)
    b.set_option('l_cap', 512)
    for ev, ks in ((ev_a, (20,)), (ev_b, (20, 15))):
        b.reset()
        b.feed(*_to_dev(ev, gpu_device))
        fresh = CovisBuilder(20000, kinds=kinds, ts_min=ts_min, ts_max=ts_max, device=gpu_device)
        fresh.set_option('l_cap', 512)
        fresh.set_option('partition', 0)                  # a build that never uses the buckets
        fresh.feed(*_to_dev(ev, gpu_device))
        for k in ks:
            out, ref = b.finalize(k=k), fresh.finalize(k=k)
            assert b.stats()['items_l'] >= 2
            got = {kind: topk_to_rows(*out[kind]) for kind in kinds}
            want = {kind: topk_to_rows(*ref[kind]) for kind in kinds}
            _assert_rows_equal(got, want, kinds)


@pytest.mark.parametrize('n_sess', [4095, 4096, 9000])
def test_packed_heavy_layout_counter_limit(gpu_device, n_sess):
    """Heavy aids with fewer than 4096 runs use 12-bit packed counters (a pair gains at most one record per session
    holding x). 4095 sessions that all hold (x, y) drive one counter to its maximum; with 4096 sessions the aid must
    fall back to the wide layout. Both against the oracle, all kinds without a filter mask plus the filter kinds."""
    rng = np.random.default_rng(n_sess)
    n_aids = 30000
    x, y = 7, 11
    z = rng.integers(100, n_aids, size=(n_sess, 3)).astype(np.uint32)
    aid = np.concatenate([np.full((n_sess, 1), x, np.uint32), np.full((n_sess, 1), y, np.uint32), z], axis=1)
    ts = (1_660_000_000 + np.cumsum(rng.integers(1, 30, size=aid.shape), axis=1)).astype(np.int32)
    typ = np.zeros(aid.shape, dtype=np.uint8)
    typ[:, 2:] = rng.integers(0, 3, size=(n_sess, 3))
    ev = Events(aid=aid.ravel(), ts=ts.ravel(), type=typ.ravel(), sess_off=np.arange(n_sess + 1, dtype=np.int64) * 5, n_aids=n_aids)
    for kinds in (NOFILT, ('click_weighted', 'click_click', 'cart_order')):
        want = _oracle_rows(ev, kinds)
        b, got = _build(ev, gpu_device, kinds=kinds)
        assert b.stats()['items_l'] >= 2          # x and y are heavy aids
        _assert_rows_equal(got, want, kinds)
        wx, wy, ww = want['click_weighted']
        assert ww[(wx == x) & (wy == y)][0] == n_sess * 65536


@pytest.mark.parametrize('options', [{'bucket_index': 0}, {'packed_heavy': 0}, {'guess': 0}, {'partition': 0},
                                     {'packed_heavy': 0, 'guess': 0, 'bucket_index': 0, 'fused': 0}],
                         ids=['atomic-index', 'wide-heavy', 'no-guess', 'no-partition', 'all-fallbacks'])
def test_pipeline_options_agree(gpu_device, options):
    """Every A/B switch of the pipeline (global-atomic index, wide layout for all heavy aids, no threshold guessing,
    filtered re-reads instead of the partition pass) gives the oracle's rows on data with partitioned heavy aids."""
    ev = generate_sessions(20000, n_aids=300, seed=23)
    kinds = ('click_weighted', 'order_weighted', 'time_weighted')
    want = _oracle_rows(ev, kinds)
    b, got = _build(ev, gpu_device, kinds=kinds, options=options)
    st = b.stats()
    assert st['items_l'] > st['items_m'] + st['items_s'] or st['items_l'] > 100     # heavy aids dominate
    _assert_rows_equal(got, want, kinds)


def test_chunked_feed_equals_single_feed(gpu_device):
    ev = generate_sessions(2500, n_aids=900, seed=21)
    _, one = _build(ev, gpu_device, chunks=1)
    _, five = _build(ev, gpu_device, chunks=5)
    _assert_rows_equal(five, one, cs.ALL_KINDS)


def test_window_gap_and_k_parameters(gpu_device):
    ev = generate_sessions(1200, n_aids=300, seed=31)
    for kinds in (('click_click', 'time_weighted', 'cart_weighted'), NOFILT):
        for window, gap, k in ((5, 600, 3), (32, 10_000_000, 32), (2, 86400, 1), (17, 0, 20), (9, 86400, 20)):
            want = _oracle_rows(ev, kinds, k=k, window=window, max_gap=gap)
            _, got = _build(ev, gpu_device, kinds=kinds, k=k, window=window, max_gap=gap)
            _assert_rows_equal(got, want, kinds)


def test_edge_cases(gpu_device):
    """Sessions of length 1 (no window), a session of one repeated aid (no pairs), an empty
    event stream, and gaps that exclude every pair."""
    aid = np.array([5, 7, 7, 7, 1, 2, 1, 9], dtype=np.uint32)
    ts = np.array([10, 20, 21, 22, 100, 100, 100000, 5], dtype=np.int32) + 1_660_000_000
    typ = np.array([0, 1, 2, 0, 0, 1, 2, 0], dtype=np.uint8)
    off = np.array([0, 1, 4, 7, 8], dtype=np.int64)
    ev = Events(aid=aid, ts=ts, type=typ, sess_off=off, n_aids=10)
    for kinds in (cs.ALL_KINDS, NOFILT):
        want = _oracle_rows(ev, kinds)
        b, got = _build(ev, gpu_device, kinds=kinds)
        _assert_rows_equal(got, want, kinds)
        assert b.stats()['pairs'] == 2   # only (1,2),(2,1) at ts 100; the third event is > 1 day later
        empty = Events(aid=aid[:0], ts=ts[:0], type=typ[:0], sess_off=np.zeros(1, dtype=np.int64), n_aids=10)
        b, got = _build(empty, gpu_device, kinds=kinds)
        assert all(len(got[k][0]) == 0 for k in kinds)


def test_multi_gpu_exchange_roundtrip(gpu_device):
    """Two session shards expanded in two contexts, runs exchanged by aid_x owner, owners reduce:
    the union equals the single-context build (the exchange the 8-GPU path does over RCCL)."""
    from otto_amd.covisitation.engine import CovisBuilder, topk_to_rows
    ev = generate_sessions(3000, n_aids=700, seed=41)
    kinds = cs.ALL_KINDS
    _, want = _build(ev, gpu_device, kinds=kinds)
    ts_min, ts_max = int(ev.ts.min()), int(ev.ts.max())
    aid, ts, typ, off = _to_dev(ev, gpu_device)
    half = ev.n_sessions // 2
    shards = []
    for lo, hi in ((0, half), (half, ev.n_sessions)):
        b = CovisBuilder(ev.n_aids, kinds=kinds, ts_min=ts_min, ts_max=ts_max, device=gpu_device)
        e0, e1 = int(ev.sess_off[lo]), int(ev.sess_off[hi])
        b.feed(aid[e0:e1].contiguous(), ts[e0:e1].contiguous(), typ[e0:e1].contiguous(), (off[lo:hi + 1] - e0).contiguous())
        shards.append(b)
    cut = ev.n_aids // 2
    got = {k: [] for k in kinds}
    for x_lo, x_hi in ((0, cut), (cut, ev.n_aids)):
        owner = CovisBuilder(ev.n_aids, kinds=kinds, ts_min=ts_min, ts_max=ts_max, device=gpu_device)
        for si, b in enumerate(shards):
            hdr, rec, tw = b.export_runs(x_lo, x_hi)
            if si == 1:
                # zero-copy receive: the records land in the owner's own arrays (what the RCCL all-to-all-v writes
                # into), import_runs then only registers the runs
                rrec, rtw = owner.import_reserve(rec.numel())
                rrec.copy_(rec)
                rtw.copy_(tw)
                rec, tw = rrec, rtw
            owner.import_runs(hdr, rec, tw)
        out = owner.finalize(k=20)
        for k in kinds:
            gx, gy, gw = topk_to_rows(*out[k])
            assert ((gx >= x_lo) & (gx < x_hi)).all()
            got[k].append((gx, gy, gw))
    merged = {k: tuple(np.concatenate([p[i] for p in got[k]]) for i in range(3)) for k in kinds}
    _assert_rows_equal(merged, want, kinds)


@pytest.mark.parametrize('kinds', [NOFILT, cs.ALL_KINDS], ids=['lists', 'class-kernels'])
def test_chunked_export_by_run_slot_ranges(gpu_device, kinds):
    """The pipelined exchange cuts the export into ranges of run slots (plan_range / fill_range: the fill of range c + 1
    runs while range c is on the links): 5 ranges x 3 owners by hand -- planned counts, filled buffers imported range
    after range into owner contexts -- equal the single-context build; the planned counts of the ranges add up to the
    one-piece plan. With the component lists a run reads a shared list: the export materialises private rows (the run's
    own entry dropped, one time extra per record)."""
    import torch
    from otto_amd.covisitation.engine import CovisBuilder, topk_to_rows
    ev = generate_sessions(2500, n_aids=500, seed=43)
    b, want = _build(ev, gpu_device, kinds=kinds)
    ts_min, ts_max = int(ev.ts.min()), int(ev.ts.max())
    bounds = [0, 150, 151, ev.n_aids]                                   # a one-aid owner in the middle
    slots = b.run_slots()
    cuts = [slots * c // 5 for c in range(6)]
    whole = b.export_all(bounds)
    owners = [CovisBuilder(ev.n_aids, kinds=kinds, ts_min=ts_min, ts_max=ts_max, device=gpu_device) for _ in range(3)]
    tot_runs, tot_recs = [0, 0, 0], [0, 0, 0]
    for c in range(5):
        runs, recs = b.export_plan_range(bounds, cuts[c], cuts[c + 1])
        hdr = torch.empty((sum(runs), 2), dtype=torch.int32, device=gpu_device)
        rec = torch.empty(sum(recs), dtype=torch.int32, device=gpu_device)
        tw = torch.empty(sum(recs), dtype=torch.int32, device=gpu_device) if b.want_time else None
        b.export_fill_range(bounds, cuts[c], cuts[c + 1], runs, recs, hdr, rec, tw)
        r0 = c0 = 0
        for o in range(3):
            if runs[o]:
                h_o = hdr[r0:r0 + runs[o]].contiguous()
                assert ((h_o[:, 0] >= bounds[o]) & (h_o[:, 0] < bounds[o + 1])).all() and int(h_o[:, 1].sum()) == recs[o]
                owners[o].import_runs(h_o, rec[c0:c0 + recs[o]].contiguous(), None if tw is None else tw[c0:c0 + recs[o]].contiguous())
            tot_runs[o] += runs[o]
            tot_recs[o] += recs[o]
            r0 += runs[o]
            c0 += recs[o]
    assert tot_runs == whole[3] and tot_recs == whole[4]
    got = {k: [] for k in kinds}
    for o in range(3):
        out = owners[o].finalize(k=20)
        for k in kinds:
            got[k].append(topk_to_rows(*out[k]))
    merged = {k: tuple(np.concatenate([p[i] for p in got[k]]) for i in range(3)) for k in kinds}
    _assert_rows_equal(merged, want, kinds)
    with pytest.raises(Exception):
        b.export_plan_range(bounds, 10, slots + 1)


def test_sharded_builder_over_rccl_world_of_one(gpu_device):
    """The N > 1 code path of bench.py over the real RCCL backend with a world of one process: the all-to-all-v writes the
    records straight into the owner context (otto_covis_import_reserve), no staging. Rows == single-context build."""
    import socket
    import torch.multiprocessing as mp
    from conftest import ROOT
    ev = generate_sessions(3000, n_aids=900, seed=61)
    _, want = _build(ev, gpu_device)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_sharded_worker, args=(0, 1, port, q, ROOT, 'nccl'))
    p.start()
    import queue
    res = None
    for _ in range(120):                       # poll: a crashed worker must not stall the suite
        try:
            res = q.get(timeout=1)
            break
        except queue.Empty:
            if not p.is_alive():
                break
    p.join(60)
    assert res is not None and p.exitcode == 0, f'RCCL worker failed (exit code {p.exitcode})'
    _assert_rows_equal(res[2], want, cs.ALL_KINDS)


def test_deterministic_across_runs(gpu_device):
    ev = generate_sessions(4000, n_aids=1500, seed=51)
    _, a = _build(ev, gpu_device)
    _, b = _build(ev, gpu_device)
    _assert_rows_equal(a, b, cs.ALL_KINDS)


def test_golden_fixture_and_hand_computed_sessions(gpu_device):
    """The committed golden rows (tests/golden/covis_golden.npz) and the hand-derived micro-sessions
    of tests/test_covis_oracle.py, through the HIP path."""
    import os
    from conftest import GOLDEN
    from test_covis_oracle import _micro, HAND, HAND_TIME
    g = np.load(os.path.join(GOLDEN, 'covis_golden.npz'))
    ev = Events(aid=g['aid'], ts=g['ts'], type=g['type'], sess_off=g['sess_off'], n_aids=int(g['n_aids']))
    _, got = _build(ev, gpu_device)
    _assert_rows_equal(got, {k: (g[f'{k}_x'], g[f'{k}_y'], g[f'{k}_w']) for k in cs.ALL_KINDS}, cs.ALL_KINDS)
    aid, ts, typ, off = _micro()
    _, got = _build(Events(aid=aid, ts=ts, type=typ, sess_off=off, n_aids=31), gpu_device)
    for kind, want in list(HAND.items()) + [('time_weighted', None)]:
        gx, gy, gw = got[kind]
        d = dict(zip(zip(gx.tolist(), gy.tolist()), gw.tolist()))
        assert d == ({p: w * cs.Q16 for p, w in want.items()} if want is not None else HAND_TIME), kind


def _sharded_worker(rank, world, port, q, root, backend='gloo'):
    import os
    import sys
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path[:0] = [root, os.path.join(root, 'oracle')]
    import torch
    import torch.distributed as dist
    from otto_amd.covisitation.distributed import ShardedCovisBuilder, global_ts_range
    from otto_amd.covisitation.engine import topk_to_rows
    dist.init_process_group(backend, rank=rank, world_size=world)
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    ev = generate_sessions(3000, n_aids=900, seed=61)
    per = ev.n_sessions // world
    lo, hi = rank * per, ev.n_sessions if rank == world - 1 else (rank + 1) * per
    e0, e1 = int(ev.sess_off[lo]), int(ev.sess_off[hi])
    aid = torch.from_numpy(ev.aid[e0:e1].astype(np.int32)).to(dev)
    ts = torch.from_numpy(ev.ts[e0:e1]).to(dev)
    typ = torch.from_numpy(ev.type[e0:e1]).to(dev)
    off = torch.from_numpy(ev.sess_off[lo:hi + 1] - e0).to(dev)
    ts_min, ts_max = global_ts_range(ts.cpu() if backend == 'gloo' else ts)
    # gloo: pieces staged through host memory; nccl (= RCCL): device buffers, records received in place
    b = ShardedCovisBuilder(ev.n_aids, cs.ALL_KINDS, ts_min, ts_max, dev, stage_device='cpu' if backend == 'gloo' else None)
    out_rows = None
    for _ in range(2):                      # twice: reset() must leave no state behind (bench loops like this)
        b.reset()
        b.feed(aid, ts, typ, off)
        out = b.finalize(k=20)
        out_rows = {k: topk_to_rows(*out[k]) for k in cs.ALL_KINDS}
    q.put((rank, b.bounds, out_rows))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_builder_two_ranks_on_one_gpu(gpu_device):
    """The real ShardedCovisBuilder (the N > 1 path of bench.py) with 2 processes sharing this GPU; the
    all-to-all-v runs over gloo with the pieces staged through host memory. Union of the two owners'
    rows == single-context build."""
    import socket
    import torch.multiprocessing as mp
    from conftest import ROOT
    ev = generate_sessions(3000, n_aids=900, seed=61)
    _, want = _build(ev, gpu_device)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q, ROOT)) for r in range(2)]
    for p in ps:
        p.start()
    res = {}
    for _ in ps:
        r, bounds, rows = q.get(timeout=300)
        res[r] = (bounds, rows)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    merged = {}
    for kind in cs.ALL_KINDS:
        parts = []
        for r in range(2):
            bounds, rows = res[r]
            gx = rows[kind][0]
            assert ((gx >= bounds[r]) & (gx < bounds[r + 1])).all()
            parts.append(rows[kind])
        merged[kind] = tuple(np.concatenate([p[i] for p in parts]) for i in range(3))
    _assert_rows_equal(merged, want, cs.ALL_KINDS)


@pytest.fixture(scope='module')
def full_otto(gpu_device):
    """BASELINE.json configs[1] input: 14,571,582 synthetic sessions (~243 M events) over 1,855,603 aids, resident in HBM."""
    import torch
    from otto_amd.synth import generate_sessions_torch, OTTO_N_AIDS, OTTO_N_SESSIONS
    d = generate_sessions_torch(OTTO_N_SESSIONS, n_aids=OTTO_N_AIDS, seed=42, device=gpu_device)
    d['ts_min'], d['ts_max'] = int(d['ts'].min()), int(d['ts'].max())
    yield d
    del d
    torch.cuda.empty_cache()


def _full_run(d, dev, kinds, k, chunks, options):
    from otto_amd.synth import OTTO_N_AIDS, OTTO_N_SESSIONS
    from otto_amd.covisitation.engine import CovisBuilder
    b = CovisBuilder(OTTO_N_AIDS, kinds=kinds, ts_min=d['ts_min'], ts_max=d['ts_max'], device=dev)
    for name, value in options.items():
        b.set_option(name, value)
    S = OTTO_N_SESSIONS
    cuts = [S * c // chunks for c in range(chunks + 1)]
    for c in range(chunks):
        lo, hi = cuts[c], cuts[c + 1]
        e0, e1 = int(d['sess_off'][lo]), int(d['sess_off'][hi])
        b.feed(d['aid'][e0:e1], d['ts'][e0:e1], d['type'][e0:e1], (d['sess_off'][lo:hi + 1] - e0).contiguous())
    out = b.finalize(k=k)
    st = b.stats()
    b.close()
    return out, st


def _assert_outputs_identical(out1, out2, kinds, k, dev):
    import torch
    for kind in kinds:
        n1, n2 = out1[kind][2], out2[kind][2]
        assert torch.equal(n1, n2), f'{kind}: list lengths differ'
        nz = torch.arange(k, device=dev)[None, :] < n1[:, None]          # entries beyond n are unspecified
        assert torch.equal(out1[kind][0][nz], out2[kind][0][nz]), f'{kind}: aid_y differs'
        assert torch.equal(out1[kind][1][nz], out2[kind][1][nz]), f'{kind}: W differs'


def _assert_list_properties(out, kinds, k, dev, n_aids):
    import torch
    for kind in kinds:
        y, w, n = out[kind]
        valid = torch.arange(k, device=dev)[None, :] < n[:, None]
        assert int(n.max()) <= k and int((n > 0).sum()) > 1_000_000
        both = valid[:, 1:] & valid[:, :-1]
        w0, w1, y0, y1 = w[:, :-1], w[:, 1:], y[:, :-1], y[:, 1:]
        assert bool(((w0 > w1) | ((w0 == w1) & (y0 < y1)))[both].all()), f'{kind}: rows not sorted by (W desc, aid_y asc)'
        xs = torch.arange(n_aids, device=dev, dtype=torch.int32)[:, None]
        assert not bool(((y == xs) & valid).any()), 'aid_x listed as its own neighbour'
        assert bool((w[valid] % 65536 == 0).all()) and bool((w[valid] > 0).all())


BENCH_KINDS = ('click_weighted', 'cart_weighted', 'order_weighted')


def test_full_otto_bench_path_bit_exact_and_sampled_oracle(gpu_device, full_otto):
    """The configuration ``bench.py`` times, at the size it times it (BASELINE.json configs[1]: 14,571,582 sessions,
    ~243 M events, 1,855,603 aids, the three type-weighted kinds, top-20), through the DEFAULT options: component-list
    pair-expand (`k_expand_lists<false>`: shared lists, one run per (aid, component)), bucketed index, packed heavy-aid
    layouts, threshold guessing. Checks:

    1. bit for bit against a run that shares none of those code paths: 7 session chunks, class-sorted pair-expand
       kernels (`fused` 0, `fast_path` 0), global-atomic index (`bucket_index` 0), wide heavy-aid tables
       (`packed_heavy` 0), two-pass top-k (`guess` 0); and against the register-row pair-expand of round 2 (`fused` 1,
       one private row per aid and window: one run per distinct aid of a window);
    2. a SAMPLED ORACLE check: ~2,000 `aid_x` stratified over the three size bins of the reduce (the five heaviest aids,
       100 of the heaviest 3 % ~ L bin, 500 of the next 26 % ~ M bin, 1,400 of the rest ~ S bin). The row of `aid_x`
       depends only on the sessions that hold `aid_x`, so the oracle run on exactly those sessions must reproduce the
       rows of the sampled aids exactly (aid_y, W and n, all three kinds)."""
    import torch
    from otto_amd.synth import OTTO_N_AIDS, OTTO_N_SESSIONS
    d, dev, k = full_otto, gpu_device, 20
    out1, st1 = _full_run(d, dev, BENCH_KINDS, k, 1, {})
    assert st1['pairs'] > 1_000_000_000 and st1['items_l'] > 0 and st1['items_m'] > 0 and st1['items_s'] > 0
    assert st1['retries'] == 0
    out2, st2 = _full_run(d, dev, BENCH_KINDS, k, 7, {'fused': 0, 'fast_path': 0, 'bucket_index': 0, 'packed_heavy': 0, 'guess': 0})
    # an aid has one run per COMPONENT of a window with the lists, one per window with private rows
    assert st1['pairs'] == st2['pairs'] and st2['runs'] <= st1['runs'] <= 1.1 * st2['runs']
    _assert_outputs_identical(out1, out2, BENCH_KINDS, k, dev)
    del out2
    out3, st3 = _full_run(d, dev, BENCH_KINDS, k, 1, {'fused': 1})
    assert st3['pairs'] == st2['pairs'] and st3['runs'] == st2['runs'] and st3['pair_slots'] < st1['pair_slots']
    _assert_outputs_identical(out1, out3, BENCH_KINDS, k, dev)
    del out3
    _assert_list_properties(out1, BENCH_KINDS, k, dev, OTTO_N_AIDS)
    # cart_weighted (1,9,6) dominates click_weighted (1,6,3) pair by pair, so its best weight per aid is >= too
    some = out1['click_weighted'][2] > 0
    assert bool((out1['cart_weighted'][1][:, 0] >= out1['click_weighted'][1][:, 0])[some].all())

    _sampled_oracle_check(d, dev, out1, BENCH_KINDS, k)


def _sampled_oracle_check(d, dev, out, kinds, k, sizes=(100, 500, 1400), min_rows=15_000):
    """~sum(sizes) `aid_x` stratified over the three size bins of the reduce (the five heaviest aids, sizes[0] of the heaviest 3 % ~ L
    bin, sizes[1] of the next 26 % ~ M bin, sizes[2] of the rest ~ S bin). The row of `aid_x` depends only on the sessions that hold
    `aid_x`, so the oracle run on exactly those sessions must reproduce the rows of the sampled aids exactly (aid_y, W and n)."""
    import torch
    from otto_amd.synth import OTTO_N_AIDS, OTTO_N_SESSIONS
    S, off, aid = OTTO_N_SESSIONS, d['sess_off'], d['aid'].long()
    E = aid.numel()
    L = off[1:] - off[:-1]
    sess = torch.repeat_interleave(torch.arange(S, device=dev), L, output_size=E)
    in_win = (off[1:][sess] - torch.arange(E, device=dev)) <= 30                     # tail window of 30 events
    cnt = torch.bincount(aid[in_win], minlength=OTTO_N_AIDS)
    order = torch.argsort(cnt, descending=True)
    n_pos = int((cnt > 0).sum())
    g = torch.Generator(device='cpu')
    g.manual_seed(7)
    b1, b2 = int(0.03 * n_pos), int(0.29 * n_pos)

    def pick(lo, hi, m):
        return order[lo + torch.randperm(hi - lo, generator=g)[:m].to(dev)]
    sample = torch.unique(torch.cat((order[:5], pick(5, b1, sizes[0]), pick(b1, b2, sizes[1]), pick(b2, n_pos, sizes[2]))))
    marked = torch.zeros(OTTO_N_AIDS, dtype=torch.bool, device=dev)
    marked[sample] = True
    sel = torch.zeros(S, dtype=torch.bool, device=dev)
    sel[sess[in_win & marked[aid]]] = True                                           # sessions whose window holds a sampled aid
    ev_sel = sel[sess]
    sub_off = np.r_[0, np.cumsum(L[sel].cpu().numpy())].astype(np.int64)
    sub_aid = d['aid'][ev_sel].cpu().numpy().astype(np.uint32)
    sub_ts, sub_typ = d['ts'][ev_sel].cpu().numpy(), d['type'][ev_sel].cpu().numpy()
    assert 20_000 < len(sub_off) - 1 < 3_000_000, 'sample of sessions out of the intended range'
    del sess, in_win, ev_sel, sel
    import covis_oracle_c as coc
    if coc.available():
        want = coc.covis_topk_c(sub_aid, sub_ts, sub_typ, sub_off, OTTO_N_AIDS, kinds, k=k,
                                ts_min=d['ts_min'], ts_max=d['ts_max'])
    else:
        want = co.covis_topk_numpy(sub_aid, sub_ts, sub_typ, sub_off,
                                   co.CovisSpec(kinds=kinds, ts_min=d['ts_min'], ts_max=d['ts_max']), k=k)
    smp = np.sort(sample.cpu().numpy())
    heavy = int(cnt.max())
    n_rows = 0
    for kind in kinds:
        wx, wy, ww = want[kind]
        keep = np.isin(wx, smp)
        wx, wy, ww = wx[keep], wy[keep], ww[keep]
        y, w, n = (t[sample.sort().values].cpu().numpy() for t in out[kind])
        valid = np.arange(k)[None, :] < n[:, None]
        gx = np.broadcast_to(smp[:, None], valid.shape)[valid].astype(np.uint32)
        assert len(gx) == len(wx), f'{kind}: {len(gx)} rows for the sampled aids vs oracle {len(wx)}'
        assert np.array_equal(gx, wx) and np.array_equal(y[valid].astype(np.uint32), wy), f'{kind}: sampled rows differ (aid)'
        assert np.array_equal(w[valid].astype(np.uint64), ww), f'{kind}: sampled rows differ (W)'
        n_rows += len(gx)
    assert n_rows > len(kinds) * min_rows and heavy > 20_000, 'sample too thin to mean anything'



def test_full_otto_time_weighted_sampled_oracle(gpu_device, full_otto):
    """The time-weighted kind at full OTTO size through the default options: component-list pair-expand with the time channel
    (`k_expand_lists<true>`), the partition pass carrying the extras, and the PACKED time sums of round 3 in all three bins (one sum of
    65536 + extra per key: the heaviest aids' keys come closest to its 2^30 bound) -- against the oracle on ~900 sampled `aid_x`."""
    from otto_amd.synth import OTTO_N_AIDS
    d, dev, k = full_otto, gpu_device, 20
    kinds = ('time_weighted',)
    out, st = _full_run(d, dev, kinds, k, 1, {})
    assert st['pairs'] > 1_000_000_000 and st['items_l'] > 0 and st['retries'] == 0
    _sampled_oracle_check(d, dev, out, kinds, k, sizes=(60, 240, 600), min_rows=6_000)


def test_full_otto_filter_kinds_properties(gpu_device, full_otto):
    """Same input with a filter kind configured (`click_click`, BASELINE config 1's mask): K1 then runs the class-sorted
    M-matrix kernels and the FILTER reduce group. Checked through size-independent properties:
      * 1 feed + partitioned heavy aids  ==  7 session chunks + the filter / re-read path for heavy aids (`partition` 0);
      * every list sorted by (W desc, aid_y asc), no aid_x in its own list, n <= k, W a positive multiple of 65536;
      * `click_click` (symmetric mask, unit weight) is symmetric: if y is listed for x and x for y the weights match."""
    import torch
    from otto_amd.synth import OTTO_N_AIDS
    d, dev, k = full_otto, gpu_device, 20
    kinds = ('click_weighted', 'cart_weighted', 'click_click')
    out1, st1 = _full_run(d, dev, kinds, k, 1, {})
    out2, st2 = _full_run(d, dev, kinds, k, 7, {'partition': 0})
    assert st1['pairs'] == st2['pairs'] > 1_000_000_000 and st1['runs'] == st2['runs']
    assert st1['items_l'] > 0 and st1['retries'] == 0
    _assert_outputs_identical(out1, out2, kinds, k, dev)
    del out2
    _assert_list_properties(out1, kinds, k, dev, OTTO_N_AIDS)
    # symmetry of click_click on a sample of aids
    y, w, n = out1['click_click']
    sample = torch.randperm(OTTO_N_AIDS, device=dev)[:200_000]
    ys, ws, ns = y[sample].long(), w[sample], n[sample]
    valid = torch.arange(k, device=dev)[None, :] < ns[:, None]
    ys = torch.where(valid, ys, torch.zeros_like(ys))                 # entries beyond n are unspecified
    back_y, back_w, back_n = y[ys], w[ys], n[ys]                      # [m, k, k] lists of the neighbours
    hit = (back_y == sample[:, None, None].to(torch.int32)) & (torch.arange(k, device=dev)[None, None, :] < back_n[..., None])
    has = hit.any(-1) & valid
    wb = (back_w * hit).sum(-1)
    assert int(has.sum()) > 10_000, 'too few mutual pairs in the sample to test symmetry'
    assert bool((wb[has] == ws[has]).all()), 'click_click is not symmetric'


def test_config1_click_click_100k_sessions_vs_numpy_oracle(gpu_device):
    """BASELINE.json configs[0] at its stated size: click->click covisitation on a 100,000-session OTTO-shape sample over
    the full aid space (1,855,603 aids), window 30, top-20 -- bit-exact against the NumPy oracle."""
    from otto_amd.synth import OTTO_N_AIDS
    ev = generate_sessions(100_000, n_aids=OTTO_N_AIDS, seed=42)
    kinds = ('click_click',)
    b, got = _build(ev, gpu_device, kinds=kinds, k=20)
    st = {}
    want = _oracle_rows(ev, kinds, k=20, stats=st)
    assert b.stats()['pairs'] == st['P'] > 5_000_000
    _assert_rows_equal(got, want, kinds)


def test_export_all_equals_per_owner_export(gpu_device):
    """The one-pass all-owner export (send buffer of the all-to-all-v) carries, per owner, exactly the runs the
    per-range export does (as multisets: run order inside a piece is unspecified)."""
    from otto_amd.covisitation.engine import CovisBuilder
    ev = generate_sessions(2500, n_aids=800, seed=71)
    b = CovisBuilder(ev.n_aids, kinds=cs.ALL_KINDS, ts_min=int(ev.ts.min()), ts_max=int(ev.ts.max()), device=gpu_device)
    b.feed(*_to_dev(ev, gpu_device))
    bounds = [0, 100, 100, 555, 800]            # includes an empty owner
    hdr, rec, tw, runs, recs = b.export_all(bounds)
    assert len(runs) == 4 and hdr.shape[0] == sum(runs) and rec.numel() == sum(recs) and runs[1] == 0
    hdr, rec, tw = hdr.cpu().numpy(), rec.cpu().numpy(), tw.cpu().numpy()

    def runs_of(h, r, t):
        out, p = [], 0
        for x, n in h.tolist():
            out.append((x, tuple(r[p:p + n].tolist()), tuple(t[p:p + n].tolist())))
            p += n
        return sorted(out)
    ro = co_ = 0
    for o in range(4):
        h1, r1, t1 = b.export_runs(bounds[o], bounds[o + 1])
        want = runs_of(h1.cpu().numpy(), r1.cpu().numpy(), t1.cpu().numpy())
        got = runs_of(hdr[ro:ro + runs[o]], rec[co_:co_ + recs[o]], tw[co_:co_ + recs[o]])
        assert got == want, f'owner {o}'
        ro += runs[o]
        co_ += recs[o]


def test_randomized_small_configurations_vs_python_oracle(gpu_device):
    """60 tiny adversarial streams (few aids -> heavy repeats and ties, equal timestamps, gaps right at the
    threshold, window / k sweeps) against the literal pure-Python transcription of SPEC-COVIS."""
    rng = np.random.default_rng(2024)
    for trial in range(60):
        kinds = cs.ALL_KINDS if trial % 3 == 0 else NOFILT      # class-sorted K1 / fused K1
        n_aids = int(rng.choice([2, 3, 5, 9, 40]))
        S = int(rng.integers(1, 60))
        window = int(rng.choice([2, 3, 7, 16, 30, 32]))
        gap = int(rng.choice([0, 1, 50, 86400]))
        k = int(rng.choice([1, 2, 5, 20, 32]))
        L = rng.integers(1, 45, S)
        off = np.r_[0, np.cumsum(L)].astype(np.int64)
        E = int(off[-1])
        aid = rng.integers(0, n_aids, E).astype(np.uint32)
        typ = rng.integers(0, 3, E).astype(np.uint8)
        step = rng.choice([0, 0, 1, 25, 51, 90000], E)           # many equal timestamps, some gaps at / over the threshold
        ts = np.zeros(E, dtype=np.int64)
        for s in range(S):
            seg = slice(int(off[s]), int(off[s + 1]))
            ts[seg] = 1_660_000_000 + np.cumsum(step[seg])
        ev = Events(aid=aid, ts=ts.astype(np.int32), type=typ, sess_off=off, n_aids=n_aids)
        sp = co.CovisSpec(window=window, max_gap=gap, kinds=kinds)
        want_pairs = co.covis_pairs_python(ev.aid, ev.ts, ev.type, ev.sess_off, sp)
        want = {kd: co.topk_rows(*co.pairs_dict_to_arrays(want_pairs[kd]), k=k) for kd in kinds}
        options = {'fast_path': 0} if trial % 3 == 2 else None
        _, got = _build(ev, gpu_device, kinds=kinds, k=k, window=window, max_gap=gap, chunks=int(rng.integers(1, 4)), options=options)
        try:
            _assert_rows_equal(got, want, kinds)
        except AssertionError as e:
            raise AssertionError(f'trial {trial}: n_aids={n_aids} S={S} window={window} gap={gap} k={k}: {e}')
