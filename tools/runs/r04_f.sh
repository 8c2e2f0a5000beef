#!/bin/bash
# round 4, batch f: MFMA groups on zero padding skipped (16x16x32 forms) -- parity, then same-box A/B:
# libtnmf_hip.so (skip) / libtnmf_hip_noskip.so / libtnmf_hip_m16only.so (12 x 12 on the 32x32x16 form, no skip there)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_f
mkdir -p $out
cd $R
timeout -k 10 900 python3 -m pytest tests/test_hip_parity.py tests/test_hip_scale.py -q -x -k "split or adversarial or lateral or non_finite or config3 or config2 or shard or loop_parity or row_padded" > $out/pytest.log 2>&1 || { echo "tests failed"; tail -40 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
timeout -k 10 600 python3 tools/probes/lib_ab.py libtnmf_hip.so libtnmf_hip_noskip.so libtnmf_hip_m16only.so > $out/ab_config3.txt 2>&1
cat $out/ab_config3.txt
timeout -k 10 600 python3 tools/probes/lib_ab.py libtnmf_hip.so libtnmf_hip_noskip.so libtnmf_hip_m16only.so -- --config 4 --steps 8 --warmup 2 > $out/ab_config4.txt 2>&1
cat $out/ab_config4.txt
timeout -k 10 600 python3 tools/probes/lib_ab.py libtnmf_hip.so libtnmf_hip_noskip.so -- --config 5 --steps 8 --warmup 2 > $out/ab_config5.txt 2>&1
cat $out/ab_config5.txt
echo batch done
