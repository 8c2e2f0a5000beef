#!/bin/bash
# round 4, batch d: persistent prefetching column transform -- parity of the FFT family, then same-box A/B against the
# one-tile-per-workgroup kernel (libtnmf_hip_colsplain.so) at the config-5 and config-4 shards
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_d
mkdir -p $out
cd $R
timeout -k 10 900 python3 -m pytest tests/test_hip_parity.py tests/test_hip_scale.py -q -x -k "fft or baseline_sizes or shard or cache or long_run" > $out/pytest.log 2>&1 || { echo "tests failed"; tail -40 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
timeout -k 10 600 python3 tools/probes/lib_ab.py libtnmf_hip.so libtnmf_hip_colsplain.so -- --config 5 --steps 8 --warmup 2 > $out/ab_config5.txt 2>&1
cat $out/ab_config5.txt
timeout -k 10 600 python3 tools/probes/lib_ab.py libtnmf_hip.so libtnmf_hip_colsplain.so -- --config 4 --steps 8 --warmup 2 > $out/ab_config4.txt 2>&1
cat $out/ab_config4.txt
echo batch done
