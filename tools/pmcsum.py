"""Averages per kernel (name substring argv[2], default all) of the rocprofv3 --pmc CSVs under a directory (argv[1])."""
import collections
import csv
import glob
import os
import sys
root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ''
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f'   {c:45s} n={len(v):3d}  avg={sum(v) / len(v):.4g}')
# durations of the same dispatches (kernel trace written beside the counters): average ns per kernel
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            dur[r['Kernel_Name'][:60]].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
for k, v in dur.items():
    print(f'{k}   avg duration {sum(v) / len(v) / 1e3:.1f} us over {len(v)} dispatches')
