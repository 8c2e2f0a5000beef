#!/bin/bash
# per-dispatch FETCH_SIZE / WRITE_SIZE of the row / column transforms at the config-5 shard (largest dispatches first)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_o
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for cnt in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $cnt --output-format csv -d $out/$cnt -- python3 $R/bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline --no-fft-variant --no-parity > $out/$cnt.log 2>&1 || { echo "pmc failed"; tail -3 $out/$cnt.log; exit 1; }
done
cd $R
python3 - <<'PY'
import csv,glob,os,collections
out=os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/r04_o'
for cnt in ('FETCH_SIZE','WRITE_SIZE'):
    rows=[]
    for f in glob.glob(out+'/'+cnt+'/**/*counter_collection.csv',recursive=True):
        for r in csv.DictReader(open(f)):
            if 'k_fft_rows_fwd' in r['Kernel_Name'] or 'k_fft_cols_fwd' in r['Kernel_Name']:
                rows.append((r['Kernel_Name'][27:50], float(r['Counter_Value'])*1024/1e9, r.get('Grid_Size','')))
    for name in ('k_fft_rows_fwd','k_fft_cols_fwd'):
        vals=sorted([v for n,v,g in rows if name in n], reverse=True)
        print(cnt, name, 'dispatches', len(vals), 'largest GB (raw counter x 1024):', [round(v,3) for v in vals[:4]], 'smallest', [round(v,3) for v in vals[-3:]])
PY
rm -rf $out/FETCH_SIZE $out/WRITE_SIZE
