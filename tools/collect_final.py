"""Copies the results of tools/final_measure.sh <tag> (gpurun_out/final_<tag>/) into profiles/: bench lines, rocprof kernel
stats, the FETCH_SIZE / WRITE_SIZE section of the PMC summary and the per-group traffic of profiles/r<NN>_traffic.json.
    python tools/collect_final.py <tag> <round>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], int(sys.argv[2])
src = os.path.join(ROOT, 'gpurun_out', 'final_' + tag)
pre = os.path.join(ROOT, 'profiles', 'r%02d_' % rnd)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(src + '/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))


def avg(sub, c):
    k = [k for k in agg if sub in k][0]
    v = agg[k][c]
    return sum(v) / len(v), len(v)


mr_f, _ = avg('k_mix_reconstruct', 'FETCH_SIZE')
mr_w, _ = avg('k_mix_reconstruct', 'WRITE_SIZE')
rf_f, nrf = avg('k_fft_rows_fwd', 'FETCH_SIZE')
rf_w, _ = avg('k_fft_rows_fwd', 'WRITE_SIZE')
sp_f, nsp = avg('k_split_corr_W', 'FETCH_SIZE')
sp_w, _ = avg('k_split_corr_W', 'WRITE_SIZE')
gw_f, _ = avg('k_mix_grad_W2', 'FETCH_SIZE')
gw_w, _ = avg('k_mix_grad_W2', 'WRITE_SIZE')
per_iter = nrf / nsp
d = json.load(open(pre + 'traffic.json'))
k = d['kernels']
k['reconstruct'].update({
    'mix_reconstruct_FETCH_SIZE_KB': round(mr_f, 1), 'mix_reconstruct_WRITE_SIZE_KB': round(mr_w, 1),
    'rows_fwd_avg_FETCH_SIZE_KB': round(rf_f, 1), 'rows_fwd_avg_WRITE_SIZE_KB': round(rf_w, 1),
    'rows_fwd_launches_per_iteration': round(per_iter, 2),
    'traffic_bytes': (2 * mr_f + mr_w) * 1024 + 0.5 * per_iter * (2 * rf_f + rf_w) * 1024})
k['update_H'].update({'FETCH_SIZE_KB': round(sp_f, 1), 'WRITE_SIZE_KB': round(sp_w, 1),
                      'traffic_bytes': (2 * sp_f + sp_w) * 1024})
k['grad_W'].update({'FETCH_SIZE_KB': round(gw_f, 1), 'WRITE_SIZE_KB': round(gw_w, 1),
                    'traffic_bytes': (2 * gw_f + gw_w) * 1024})
json.dump(d, open(pre + 'traffic.json', 'w'), indent=1)
print({n: round(k[n]['traffic_bytes'] / 1e9, 3) for n in k})
for c in ('n1', 'config2', 'config4', 'config5'):
    shutil.copy(f'{src}/bench_{c}.json', pre + f'bench_{c}.json')
shutil.copy(glob.glob(src + '/stats/**/*kernel_stats.csv', recursive=True)[0], pre + 'rocprof_kernel_stats.csv')
lines = []
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    for kk, v in agg.items():
        if c in v and any(s in kk for s in ('k_fft_rows_fwd', 'k_mix_reconstruct', 'k_fft_rows_inv', 'k_split_corr_W',
                                            'k_mix_grad_W2', 'k_fft_sum_groups')):
            lines.append(f'{c:10s}  {kk[:62]:62s} n={len(v[c]):3d} avg={sum(v[c]) / len(v[c]):12.1f} KB')
pp = pre + 'rocprof_pmc_summary.txt'
s = open(pp).read()
mark = '## final code of the round (tools/final_measure.sh ' 
if mark in s:
    s = s[:s.index(mark)]
    open(pp, 'w').write(s.rstrip('\n') + '\n')
open(pp, 'a').write('\n' + mark + tag + ', tools/collect_final.py): kernels of the MU iteration, config 3, default dispatch\n' + '\n'.join(lines) + '\n')
