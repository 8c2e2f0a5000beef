#!/bin/bash
# round 4, batch g: the XCD-local schedule kernel -- XCC ids of a plain launch, the mini-batch tests, the small-batch bench legs
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_g
mkdir -p $out
cd $R
timeout -k 5 60 tools/probes/xcc_probe > $out/xcc_probe.txt 2>&1; cat $out/xcc_probe.txt
timeout -k 10 600 python3 -m pytest tests/test_hip_parity.py tests/test_hip_scale.py tests/test_hip_volumes.py -q -x -k "minibatch or schedule or stream or known_answer" > $out/pytest.log 2>&1 || { echo "tests failed"; tail -40 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
Q="--no-cpu-baseline --no-fft-variant --no-parity"
for a in asg cyclic gsg asag gsag; do
  timeout -k 10 200 python3 bench.py --config 8 --batch-size 3 $Q --steps 5 --warmup 2 --algorithm $a > $out/bench_mb_$a.json 2> $out/b.err || { echo "bench $a failed"; tail -5 $out/b.err; exit 1; }
  python3 -c "import json;d=json.load(open('$out/bench_mb_$a.json'));print('$a', round(d['ms_per_step'],3),'ms/epoch', d['config']['energy_after_run'])"
done
timeout -k 10 200 python3 bench.py --config 8 --batch-size 3 $Q --steps 5 --warmup 2 --algorithm asg --eager > $out/bench_mb_asg_eager.json 2> $out/b.err
python3 -c "import json;d=json.load(open('$out/bench_mb_asg_eager.json'));print('asg eager', round(d['ms_per_step'],3),'ms/epoch', d['config']['energy_after_run'])"
echo batch done
