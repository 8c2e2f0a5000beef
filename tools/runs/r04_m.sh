#!/bin/bash
# round 4, batch m: bench.py --gpus 2 at full size, self-launched, two ranks sharing the one GPU with the collective through
# gloo -- a functional rehearsal of the SCALE command (not a measurement)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_m
mkdir -p $out
cd $R
export TNMF_BENCH_DIST_BACKEND=gloo
SECONDS=0; timeout -k 10 900 python3 bench.py --gpus 2 --steps 10 --warmup 2 > $out/rehearsal.json 2> $out/rehearsal.err || { echo failed; tail -30 $out/rehearsal.err; exit 1; }
echo "wall seconds: $SECONDS"
python3 - <<'PY'
import json,os
d=json.load(open(os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/r04_m/rehearsal.json'))
print('value',round(d['value'],1),'n_gpus',d['n_gpus'],'ranks seen',d['rccl_ranks_seen'],d['distributed']['backend'])
for k in ('strong_scaling','config4_cyclic','config5_cyclic'):
    print(k, round(d[k]['value'],2), d[k].get('speedup_over_one_gpu'), d[k]['global_samples'])
PY
echo batch done
