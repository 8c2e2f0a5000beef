#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_r
mkdir -p $out
cd $R
timeout -k 10 900 python3 -m pytest tests/test_hip_parity.py tests/test_hip_scale.py -q -x -k "golden or oracle or known_answer or minibatch or schedule or stream or joined or persistent or two_ranks or own_blocks or ordered or one_dimensional or errors or lateral" > $out/pytest.log 2>&1 || { echo "tests failed"; tail -40 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
Q="--no-cpu-baseline --no-fft-variant --no-parity"
for a in asg cyclic asag; do
  timeout -k 10 200 python3 bench.py --config 8 --batch-size 3 $Q --steps 5 --warmup 2 --algorithm $a > $out/bench_minibatch_geometry_$a.json 2> $out/b.err || { echo "bench $a failed"; tail -5 $out/b.err; exit 1; }
  python3 -c "import json;d=json.load(open('$out/bench_minibatch_geometry_$a.json'));print('$a', round(d['ms_per_step'],3),'ms/epoch', d['config']['energy_after_run'])"
done
timeout -k 10 200 python3 bench.py --config 1 $Q --steps 200 --warmup 20 > $out/bench_config1.json 2> $out/b.err && python3 -c "import json;d=json.load(open('$out/bench_config1.json'));print('config1', round(d['ms_per_step'],4),'ms/iter')"
echo batch done
