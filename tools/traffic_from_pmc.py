"""HBM traffic per kernel of a bench.py run from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, one pass each).

    python tools/traffic_from_pmc.py <dir with the counter_collection CSVs> <config id> <iterations of the run> <out.json>

Bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KB, and FETCH_SIZE counts 64-byte units as
32 on gfx950 (MI355X_MICROARCH.md, "HBM / rocprofv3"; calibrated on this repo's access shapes in round 2,
profiles/r02_rocprof_pmc_summary.txt: 0.500 of the known bytes with whole-line rows, WRITE_SIZE exact).
`launches_per_iteration` = launches seen / iterations of the run (warm-up included); the handful of set-up launches of
bench.py (synthetic data through the reconstruct kernels, once) are in there: an upper bound by a few percent.
"""
import collections
import csv
import glob
import json
import os
import sys

root, cfg, iters, out = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), sys.argv[4]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
per = {}
total = 0.0
for name, cs in sorted(agg.items()):
    if 'FETCH_SIZE' not in cs or 'WRITE_SIZE' not in cs or '::k_' not in name:   # (this library's kernels)
        continue
    n = len(cs['FETCH_SIZE'])
    fetch = sum(cs['FETCH_SIZE']) / n
    write = sum(cs['WRITE_SIZE']) / len(cs['WRITE_SIZE'])
    b = (2 * fetch + write) * 1024
    per[name] = {'launches': n, 'launches_per_iteration': round(n / iters, 3), 'FETCH_SIZE_KB': round(fetch, 1),
                 'WRITE_SIZE_KB': round(write, 1), 'bytes_per_launch': b, 'bytes_per_iteration': b * n / iters}
    total += b * n / iters
json.dump({'config': cfg, 'iterations_of_the_run': iters,
           'method': __doc__.split('\n\n')[1].replace('\n', ' '),
           'iteration_bytes': total, 'per_kernel': per}, open(out, 'w'), indent=1)
print(f'config {cfg}: {total / 1e9:.3f} GB per iteration')
for k, v in sorted(per.items(), key=lambda kv: -kv[1]['bytes_per_iteration'])[:8]:
    print(f"  {k[:70]:70s} {v['bytes_per_launch'] / 1e9:8.3f} GB x {v['launches_per_iteration']}")
