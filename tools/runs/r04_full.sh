#!/bin/bash
# the whole GPU suite + smoke, as the driver runs them at round end
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_full
mkdir -p $out
cd $R
timeout -k 10 1150 python3 -m pytest tests/ -x -q -m gpu > $out/pytest.log 2>&1; rc=$?
tail -5 $out/pytest.log
[ $rc -eq 0 ] || { grep -n "Error\|error\|FAILED" $out/pytest.log | head -20; exit 1; }
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1 || { echo "smoke failed"; tail -20 $out/smoke.log; exit 1; }
tail -3 $out/smoke.log
echo suite done
