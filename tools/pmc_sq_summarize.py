"""Per-kernel SQ activity ratios from one `rocprofv3 --pmc SQ_...` pass over tools/perf_covis.py (counter_collection.csv).
Ratios are per SQ_WAVE_CYCLES (cycles a wave is resident): with w waves per SIMD a pipe is busy about w x the ratio."""
import collections, csv, glob, sys
f = (glob.glob(sys.argv[1] + '/*_counter_collection.csv') + glob.glob(sys.argv[1] + '/*/*_counter_collection.csv'))[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
order = []
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if 'otto' not in k:
        continue
    if k not in acc:
        order.append(k)
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
print('ratios per SQ_WAVE_CYCLES; counters: ' + ' '.join(sorted({c for v in acc.values() for c in v})))
for k in order:
    v = acc[k]
    wc = v.get('SQ_WAVE_CYCLES', 0.0)
    if wc <= 0:
        continue
    g = lambda n: v.get(n, 0.0) / wc
    print(f"{k[:78]:78s} inst_valu {g('SQ_ACTIVE_INST_VALU'):.3f} inst_any {g('SQ_ACTIVE_INST_ANY'):.3f} inst_lds {g('SQ_ACTIVE_INST_LDS'):.3f} "
          f"wait_any {g('SQ_WAIT_ANY'):.3f} wait_inst {g('SQ_WAIT_INST_ANY'):.3f} wait_lds {g('SQ_WAIT_INST_LDS'):.3f}")
