#!/bin/bash
# round 4, batch k: what the three LDS stages of the column transform cost (timing-only ablation: libtnmf_hip_nostages.so
# skips them -- wrong results by design) at the config-5 and config-4 shards
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_k
mkdir -p $out
cd $R
timeout -k 10 600 python3 tools/probes/lib_ab.py libtnmf_hip.so libtnmf_hip_nostages.so -- --config 5 --steps 8 --warmup 2 > $out/ab_config5.txt 2>&1
cat $out/ab_config5.txt
timeout -k 10 600 python3 tools/probes/lib_ab.py libtnmf_hip.so libtnmf_hip_nostages.so -- --config 4 --steps 8 --warmup 2 > $out/ab_config4.txt 2>&1
cat $out/ab_config4.txt
echo batch done
