"""Per-kernel instruction totals from a `rocprofv3 --pmc SQ_INSTS_... SQ_WAVES` pass (counter_collection.csv): wave-instructions
per kind, summed over the dispatches of a kernel, and per wave."""
import collections, csv, glob, sys
f = (glob.glob(sys.argv[1] + '/*_counter_collection.csv') + glob.glob(sys.argv[1] + '/*/*_counter_collection.csv'))[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if 'otto' not in k:
        continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    disp[k].add(r['Dispatch_Id'])
for k, v in acc.items():
    w = v.get('SQ_WAVES', 0.0) or 1.0
    print(f"{k[:70]:70s} dispatches {len(disp[k]):3d} waves {w:.3g} | " + ' '.join(f"{c[8:] if c.startswith('SQ_INSTS_') else c} {x:.4g} ({x / w:.0f}/wave)" for c, x in sorted(v.items()) if c != 'SQ_WAVES'))
