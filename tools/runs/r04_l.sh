#!/bin/bash
# round 4, batch l: column transform of the long lengths on whole lines (16-column tiles as two halves of eight) -- parity, A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_l
mkdir -p $out
cd $R
timeout -k 10 900 python3 -m pytest tests/test_hip_parity.py tests/test_hip_scale.py -q -x -k "fft or baseline_sizes or shard or config5" > $out/pytest.log 2>&1 || { echo "tests failed"; tail -40 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
timeout -k 10 600 python3 tools/probes/lib_ab.py libtnmf_hip.so libtnmf_hip_colshalf.so -- --config 5 --steps 8 --warmup 2 > $out/ab_config5.txt 2>&1
cat $out/ab_config5.txt
echo batch done
