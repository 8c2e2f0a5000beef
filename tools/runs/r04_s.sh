#!/bin/bash
# round 4, batch s: the column transform of the long lengths as 16-column tiles with twice the threads (libtnmf_hip_colswide.so)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_s
mkdir -p $out
cd $R
timeout -k 10 600 python3 tools/probes/lib_ab.py libtnmf_hip.so libtnmf_hip_colswide.so -- --config 5 --steps 8 --warmup 2 > $out/ab_config5.txt 2>&1
cat $out/ab_config5.txt
echo batch done
