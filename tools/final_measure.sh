#!/bin/bash
# Round-end measurement batch on the GPU box: bench lines, rocprof kernel stats, FETCH/WRITE PMC passes.
# usage (through gpurun): bash tools/final_measure.sh <tag>      -> files under gpurun_out/final_<tag>/
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r}
out=$R/gpurun_out/final_$tag
mkdir -p $out
cd $R
timeout -k 10 400 python bench.py > $out/bench_n1.json 2> $out/bench_n1.err || { echo "bench failed"; tail -3 $out/bench_n1.err; exit 1; }
echo "bench n1 done"
for c in 2 4 5; do
  timeout -k 10 300 python bench.py --config $c --no-cpu-baseline --steps 8 --warmup 2 > $out/bench_config$c.json 2> $out/bench_config$c.err || { echo "bench $c failed"; exit 1; }
  echo "bench config $c done"
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-fft-variant --no-parity > $out/stats.log 2>&1 || { echo "stats failed"; exit 1; }
echo "stats done"
for cnt in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $cnt --output-format csv -d $out/pmc_$cnt -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-fft-variant --no-parity > $out/pmc_$cnt.log 2>&1 || { echo "pmc $cnt failed"; exit 1; }
  echo "pmc $cnt done"
done
cd $R
python tools/pmcsum.py $out k_ > $out/pmc_summary.txt 2>&1
echo all done
