#!/bin/bash
# round 4, first GPU batch: the new / changed tests, the bench line (N = 1 and the self-launched two-rank rehearsal), and
# the FETCH_SIZE / WRITE_SIZE calibration on the column kernels' half-line access shape
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_a
mkdir -p $out
cd $R
timeout -k 10 900 python3 -m pytest tests/test_hip_parity.py -q -x -k "adversarial or f32_loop_parity" > $out/pytest_parity.log 2>&1 || { echo "parity subset failed"; tail -30 $out/pytest_parity.log; exit 1; }
tail -2 $out/pytest_parity.log
timeout -k 10 900 python3 -m pytest tests/test_hip_scale.py -q -x -k "config2 or refit or bring_their_own or rehearsal or launches_its_own" > $out/pytest_scale.log 2>&1 || { echo "scale subset failed"; tail -40 $out/pytest_scale.log; exit 1; }
tail -2 $out/pytest_scale.log
timeout -k 10 500 python3 bench.py > $out/bench_n1.json 2> $out/bench_n1.err || { echo "bench failed"; tail -5 $out/bench_n1.err; exit 1; }
python3 tools/benchsum.py $out/bench_n1.json 2>/dev/null | head -20
cd /tmp && export TMPDIR=/tmp
for cnt in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $cnt --output-format csv -d $out/calib_$cnt -- $R/tools/probes/fetch_calib_cols > $out/calib_$cnt.log 2>&1 || { echo "calib $cnt failed"; tail -5 $out/calib_$cnt.log; exit 1; }
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/calib_stats -- $R/tools/probes/fetch_calib_cols > $out/calib_stats.log 2>&1
cd $R
(python3 tools/pmcsum.py $out/calib_FETCH_SIZE; python3 tools/pmcsum.py $out/calib_WRITE_SIZE; cat $out/calib_FETCH_SIZE.log | grep k_cols) > $out/calib_summary.txt 2>&1
cat $out/calib_summary.txt
find $out/calib_stats -name '*kernel_stats.csv' | head -1 | xargs cat | cut -c1-150 >> $out/calib_summary.txt
rm -rf $out/calib_FETCH_SIZE $out/calib_WRITE_SIZE $out/calib_stats
echo done
