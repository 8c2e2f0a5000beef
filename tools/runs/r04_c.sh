#!/bin/bash
# round 4, batch c: the 16 x 16 split instantiation on v_mfma_f32_16x16x32_bf16 -- parity, then a same-box A/B against
# the 32x32x16 form (libtnmf_hip_m32.so) at the config-5 shard
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_c
mkdir -p $out
cd $R
timeout -k 10 900 python3 -m pytest tests/test_hip_parity.py tests/test_hip_scale.py -q -x -k "split or adversarial or config5 or lateral or non_finite" > $out/pytest.log 2>&1 || { echo "tests failed"; tail -40 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
timeout -k 10 900 python3 tools/probes/lib_ab.py libtnmf_hip.so libtnmf_hip_m32.so -- --config 5 --steps 8 --warmup 2 > $out/ab_config5.txt 2>&1
cat $out/ab_config5.txt
echo batch done
