#!/bin/bash
# round 4, batch b: XCD-contiguous block order of the 8-column tile kernels + column length 540 outside path='fft':
# parity at the config-4/5 geometries, bench lines of configs 5, 4, rocprof stats + PMC traffic
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_b
mkdir -p $out
cd $R
timeout -k 10 1100 python3 -m pytest tests/test_hip_parity.py tests/test_hip_scale.py -q -x -k "baseline_sizes or shard or fft_family or persistent or minibatch_slices" > $out/pytest.log 2>&1 || { echo "tests failed"; tail -40 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
for c in 5 4; do
  timeout -k 10 400 python3 bench.py --config $c --no-cpu-baseline --steps 8 --warmup 2 > $out/bench_config$c.json 2> $out/bench_config$c.err || { echo "bench $c failed"; tail -5 $out/bench_config$c.err; exit 1; }
  python3 tools/benchsum.py $out/bench_config$c.json | head -12
done
bash tools/final_measure.sh r04_b "5 4" nobench
echo batch done
