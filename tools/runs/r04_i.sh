#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_i
mkdir -p $out
cd $R
timeout -k 10 900 python3 -m pytest tests/test_hip_parity.py tests/test_hip_scale.py tests/test_hip_volumes.py -q -x -k "minibatch or schedule or stream or known_answer or persistent or refit or one_dimensional" > $out/pytest.log 2>&1 || { echo "tests failed"; tail -40 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
Q="--no-cpu-baseline --no-fft-variant --no-parity"
timeout -k 10 200 python3 bench.py --config 8 --batch-size 3 $Q --steps 5 --warmup 2 --algorithm asg > $out/bench_mb_asg.json 2> $out/b.err && python3 -c "import json;d=json.load(open('$out/bench_mb_asg.json'));print('asg', round(d['ms_per_step'],3),'ms/epoch')"
timeout -k 10 200 python3 bench.py --config 1 $Q --steps 200 --warmup 20 > $out/bench_config1.json 2> $out/b.err && python3 -c "import json;d=json.load(open('$out/bench_config1.json'));print('config1', round(d['ms_per_step'],4),'ms/iter')"
echo batch done
