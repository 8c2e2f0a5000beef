#!/bin/bash
# Round-end measurement batch on the GPU box: bench lines, rocprof kernel stats and FETCH/WRITE PMC passes of every
# BASELINE configuration that fits one GPU (2, 3, 4-shard, 5-shard), Cyclic-MU lines of configs 4 and 5.
# usage (through gpurun): bash tools/final_measure.sh <tag> [configs, default "3 2 4 5"]   -> gpurun_out/final_<tag>/
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r}
configs=${2:-"3 2 4 5"}
out=$R/gpurun_out/final_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
PROF_ARGS="--steps 10 --warmup 2 --no-cpu-baseline --no-fft-variant --no-parity"
for c in $configs; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_config$c -o s -- python3 $R/bench.py --config $c $PROF_ARGS > $out/stats_config$c.log 2>&1 || { echo "stats $c failed"; tail -3 $out/stats_config$c.log; exit 1; }
  echo "stats config $c done"
  for cnt in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $cnt --output-format csv -d $out/pmc_config$c/$cnt -- python3 $R/bench.py --config $c $PROF_ARGS > $out/pmc_config${c}_$cnt.log 2>&1 || { echo "pmc $c $cnt failed"; exit 1; }
  done
  echo "pmc config $c done"
done
cd $R
for c in $configs; do
  python3 tools/traffic_from_pmc.py $out/pmc_config$c $c 12 $out/traffic_config$c.json || exit 1
  cp $(find $out/stats_config$c -name '*kernel_stats.csv' | head -1) $out/rocprof_kernel_stats_config$c.csv || exit 1
  rm -rf $out/pmc_config$c $out/stats_config$c     # (raw traces: tens of MB)
done
echo all done
