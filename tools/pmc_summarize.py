"""Turns the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic.py into per-kernel HBM traffic (bytes per launch).

Method (MI355X_MICROARCH.md section HBM): the two counters are collected in SEPARATE rocprofv3 --pmc passes; they count in
KiB at the L2's memory-side interface; FETCH_SIZE under-reports some access widths on gfx950, so both counters are first
scaled by a calibration launch of KNOWN size with the same 4-byte-per-lane access shape (otto_calib_*_u32, 4 GiB each)."""
import collections, csv, glob, json, sys

def load(d):
    f = (glob.glob(d + '/*_counter_collection.csv') + glob.glob(d + '/*/*_counter_collection.csv'))[0]
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        per[r['Kernel_Name']].append(float(r['Counter_Value']))
    return per

fetch, write = load(sys.argv[1]), load(sys.argv[2])
KNOWN = 4.0 * (1 << 30)
def find(per, key):
    return [k for k in per if key in k]
cal_r = sum(fetch[find(fetch, 'otto_calib_read_u32')[0]]) * 1024 / KNOWN     # reported / true for 4 B/lane loads
cal_w = sum(write[find(write, 'otto_calib_write_u32')[0]]) * 1024 / KNOWN
out = {'method': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; KiB units; scaled by calibration launches of 4 GiB '
                 'with 4-byte-per-lane loads / stores (reported/true ratios below)',
       'calibration': {'fetch_reported_over_true': round(cal_r, 4), 'write_reported_over_true': round(cal_w, 4),
                       'raw_read_launch_write_counter_KiB': sum(write[find(write, 'otto_calib_read_u32')[0]]),
                       'raw_write_launch_fetch_counter_KiB': sum(fetch[find(fetch, 'otto_calib_write_u32')[0]])},
       'kernels': {}}
for k in sorted(set(fetch) | set(write)):
    if 'otto' not in k or 'calib' in k:
        continue
    fv, wv = fetch.get(k, []), write.get(k, [])
    n = max(len(fv), len(wv), 1)
    rd = sum(fv) * 1024 / cal_r / n
    wr = sum(wv) * 1024 / cal_w / n
    if rd + wr < 1e6:
        continue
    out['kernels'][k[:90]] = {'launches': n, 'read_bytes_per_launch': int(rd), 'write_bytes_per_launch': int(wr),
                              'traffic_bytes_per_launch': int(rd + wr)}
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
out['source_hash'] = bench.source_hash()      # bench.py attaches these numbers only while the kernel sources are unchanged
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(json.dumps(out['calibration']))
for k, v in out['kernels'].items():
    print(k[:70].ljust(70), v['launches'], f"rd {v['read_bytes_per_launch']/1e9:.2f} GB  wr {v['write_bytes_per_launch']/1e9:.2f} GB")
