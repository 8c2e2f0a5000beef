#!/bin/bash
# round 4, batch j: new split shapes; bench.py's process-group code path on RCCL with ONE rank (the legs of an N > 1 run)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_j
mkdir -p $out
cd $R
timeout -k 10 600 python3 -m pytest tests/test_hip_parity.py -q -x -k "split_h_gradient" > $out/pytest.log 2>&1 || { echo "tests failed"; tail -40 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
TNMF_BENCH_FORCE_DIST=1 timeout -k 10 600 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-fft-variant > $out/bench_rccl_one_rank.json 2> $out/bench_rccl_one_rank.err || { echo "bench failed"; tail -20 $out/bench_rccl_one_rank.err; exit 1; }
python3 - <<'PY'
import json,os
d=json.load(open(os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/r04_j/bench_rccl_one_rank.json'))
print('value',d['value'],'ranks seen',d.get('rccl_ranks_seen'),d['distributed']['backend'],'exchange us',d['distributed']['exchange_us_avg_of_50'])
for k in ('strong_scaling','config4_cyclic','config5_cyclic'):
    print(k, d[k]['value'], d[k]['unit'], d[k].get('speedup_over_one_gpu'))
PY
echo batch done
