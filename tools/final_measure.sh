#!/bin/bash
# Round-end measurement batch on the GPU box: bench lines, rocprof kernel stats and FETCH/WRITE PMC passes of every
# BASELINE configuration that fits one GPU (2, 3, 4-shard, 5-shard), Cyclic-MU lines of configs 4 and 5, the small-batch
# schedules, the inhibition leg.
# usage (through gpurun): bash tools/final_measure.sh <tag> [configs, default "3 2 4 5"] [nobench|noprof]  -> gpurun_out/final_<tag>/
# (the whole batch is longer than one gpurun call allows: run it as `... <tag> "3 2 4 5" noprof` and `... <tag> "3 2 4 5" nobench`)
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r}
configs=${2:-"3 2 4 5"}
out=$R/gpurun_out/final_$tag
mkdir -p $out
cd $R
python3 -c "import bench; print(bench.csrc_digest())" > $out/sources.sha16 || exit 1   # what these profiles are measured on
Q="--no-cpu-baseline --no-fft-variant --no-parity"
if [ "$3" != "nobench" ]; then
  timeout -k 10 500 python3 bench.py > $out/bench_n1.json 2> $out/bench_n1.err || { echo "bench failed"; tail -3 $out/bench_n1.err; exit 1; }
  echo "bench n1 done"
  for c in 2 4 5; do
    timeout -k 10 400 python3 bench.py --config $c --no-cpu-baseline --steps 8 --warmup 2 > $out/bench_config$c.json 2> $out/bench_config$c.err || { echo "bench $c failed"; exit 1; }
    echo "bench config $c done"
  done
  timeout -k 10 300 python3 bench.py --config 4 $Q --steps 8 --warmup 2 --algorithm cyclic --batch-size 64 > $out/bench_config4_cyclic.json 2> $out/b.err || exit 1
  timeout -k 10 300 python3 bench.py --config 5 $Q --steps 8 --warmup 2 --algorithm cyclic --batch-size 32 > $out/bench_config5_cyclic.json 2> $out/b.err || exit 1
  timeout -k 10 300 python3 bench.py $Q --steps 10 --warmup 2 --inhibition 0.1 > $out/bench_config3_inhibition.json 2> $out/b.err || exit 1
  timeout -k 10 300 python3 bench.py $Q --steps 10 --warmup 2 --inhibition 0.1 --cross-inhibition 0.05 > $out/bench_config3_inhibition_cross.json 2> $out/b.err || exit 1
  timeout -k 10 300 python3 bench.py --config 1 $Q --steps 200 --warmup 20 > $out/bench_config1.json 2> $out/b.err || exit 1
  timeout -k 10 300 python3 bench.py --config 1 $Q --steps 200 --warmup 20 --eager > $out/bench_config1_eager.json 2> $out/b.err || exit 1
  timeout -k 10 300 python3 bench.py --config 7 $Q --steps 20 --warmup 3 > $out/bench_long_1d.json 2> $out/b.err || exit 1
  timeout -k 10 300 python3 bench.py --config 9 $Q --steps 10 --warmup 2 > $out/bench_volumes.json 2> $out/b.err || exit 1
  for a in asg asag cyclic gsg gsag; do
    timeout -k 10 300 python3 bench.py --config 8 --batch-size 3 $Q --steps 5 --warmup 2 --algorithm $a > $out/bench_minibatch_geometry_$a.json 2> $out/b.err || exit 1
    timeout -k 10 300 python3 bench.py --config 8 --batch-size 3 $Q --steps 5 --warmup 2 --algorithm $a --eager > $out/bench_minibatch_geometry_${a}_eager.json 2> $out/b.err || exit 1
  done
  echo "bench legs done"
fi
if [ "$3" == "noprof" ]; then echo "all done (no profiles)"; exit 0; fi
cd /tmp && export TMPDIR=/tmp
PROF_ARGS="--steps 10 --warmup 2 $Q"
for c in $configs; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_config$c -o s -- python3 $R/bench.py --config $c $PROF_ARGS > $out/stats_config$c.log 2>&1 || { echo "stats $c failed"; tail -3 $out/stats_config$c.log; exit 1; }
  echo "stats config $c done"
  for cnt in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $cnt --output-format csv -d $out/pmc_config$c/$cnt -- python3 $R/bench.py --config $c $PROF_ARGS > $out/pmc_config${c}_$cnt.log 2>&1 || { echo "pmc $c $cnt failed"; exit 1; }
    # the same with 4 timed steps: the per-iteration traffic is the difference of the two runs (set-up launches cancel)
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $cnt --output-format csv -d $out/pmc_short_config$c/$cnt -- python3 $R/bench.py --config $c --steps 4 --warmup 2 $Q > $out/pmc_short_config${c}_$cnt.log 2>&1 || { echo "pmc short $c $cnt failed"; exit 1; }
  done
  echo "pmc config $c done"
done
cd $R
for c in $configs; do
  python3 tools/traffic_from_pmc.py $out/pmc_config$c $c 10 $out/traffic_config$c.json $out/pmc_short_config$c 4 || exit 1
  cp $(find $out/stats_config$c -name '*kernel_stats.csv' | head -1) $out/rocprof_kernel_stats_config$c.csv || exit 1
  rm -rf $out/pmc_config$c $out/pmc_short_config$c $out/stats_config$c     # (raw traces: tens of MB)
done
echo all done
