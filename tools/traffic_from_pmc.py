"""HBM traffic per kernel and per MU iteration of a bench.py run from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, one pass each).

    python tools/traffic_from_pmc.py <dir of the long run> <config id> <steps of the long run> <out.json> [<dir of the short run> <its steps>]

Bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KB, and FETCH_SIZE counts 64-byte units as
32 on gfx950 (MI355X_MICROARCH.md, "HBM / rocprofv3"; calibrated on this repo's access shapes: whole lines in round 2,
profiles/r02_rocprof_pmc_summary.txt, half lines in round 4, profiles/r04_fetch_calib_cols.txt; WRITE_SIZE is exact).

Per ITERATION (round 4): the difference of two runs that differ ONLY in their number of timed steps, divided by the
difference of the step counts -- the set-up launches of bench.py (the synthetic data goes through the product's own
reconstruct kernels once, the warm-up, the energy evaluation at the end) cancel exactly.  Without a short run the old
estimate is used: everything the long run launched / its steps + warm-up (an upper bound: at the config-5 shard the set-up
is a whole extra transform pass over H, +8 %).
`bytes_per_launch`: the LARGEST dispatch of the kernel in the long run (the launch on the activations; the same kernel also
runs on the few planes of V, R and W) -- per-name averages over launches of different sizes say nothing.
"""
import collections
import csv
import glob
import json
import os
import sys

root, cfg, steps, out = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), sys.argv[4]
short_root, short_steps = (sys.argv[5], float(sys.argv[6])) if len(sys.argv) > 6 else (None, None)


def collect(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    return agg


def total_bytes(cs):
    return (2 * sum(cs.get('FETCH_SIZE', [])) + sum(cs.get('WRITE_SIZE', []))) * 1024


long_run = collect(root)
short_run = collect(short_root) if short_root else None
warmup = 2.0    # tools/final_measure.sh: --warmup 2
per = {}
total = 0.0
for name, cs in sorted(long_run.items()):
    if 'FETCH_SIZE' not in cs or 'WRITE_SIZE' not in cs or '::k_' not in name:   # (this library's kernels)
        continue
    n = len(cs['FETCH_SIZE'])
    if short_run is not None:
        scs = short_run.get(name, {})
        per_it = (total_bytes(cs) - total_bytes(scs)) / (steps - short_steps)
        launches_it = (n - len(scs.get('FETCH_SIZE', []))) / (steps - short_steps)
    else:
        per_it = total_bytes(cs) / (steps + warmup)
        launches_it = n / (steps + warmup)
    big_f, big_w = max(cs['FETCH_SIZE']), max(cs['WRITE_SIZE'])
    per[name] = {'launches': n, 'launches_per_iteration': round(launches_it, 3),
                 'FETCH_SIZE_KB': round(big_f, 1), 'WRITE_SIZE_KB': round(big_w, 1),
                 'bytes_per_launch': (2 * big_f + big_w) * 1024, 'bytes_per_iteration': per_it}
    total += per_it
method = ('difference of two runs (%g and %g timed steps) / difference of the steps: set-up launches cancel' % (steps, short_steps)
          if short_run is not None else 'all launches of one run / (steps + warm-up): an upper bound (set-up launches included)')
json.dump({'config': cfg, 'steps_of_the_long_run': steps, 'steps_of_the_short_run': short_steps,
           'method': '(2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes (gfx950: FETCH_SIZE counts 64-byte units as 32); per iteration: ' + method
                     + '; bytes_per_launch = the largest dispatch of the kernel',
           'iteration_bytes': total, 'per_kernel': per}, open(out, 'w'), indent=1)
print(f'config {cfg}: {total / 1e9:.3f} GB per iteration ({method})')
for k, v in sorted(per.items(), key=lambda kv: -kv[1]['bytes_per_iteration'])[:8]:
    print(f"  {k[:70]:70s} {v['bytes_per_iteration'] / 1e9:8.3f} GB per iteration, largest launch {v['bytes_per_launch'] / 1e9:.3f} GB x {v['launches_per_iteration']}")
