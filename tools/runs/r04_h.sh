#!/bin/bash
# round 4, batch h: where the XCD-local schedule kernel's time goes -- members per CU 8 / 4 / 1, and the same walk with the
# block functions taken out (barriers + loop overhead only); ASG-MU on the reference's mini-batch geometry
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_h
mkdir -p $out
cd $R
timeout -k 10 900 python3 tools/probes/lib_ab.py libtnmf_hip.so libtnmf_hip_pc8.so libtnmf_hip_pc1.so libtnmf_hip_nowork.so -- --config 8 --batch-size 3 --steps 5 --warmup 2 --algorithm asg > $out/ab.txt 2>&1
cat $out/ab.txt
echo batch done
