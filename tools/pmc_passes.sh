#!/bin/bash
# usage: tools/pmc_passes.sh <outdir under gpurun_out> "<counters of pass 1>" "<counters of pass 2>" ... -- <python script and args>
# one rocprofv3 --pmc pass per counter group (kernel trace only beside it), each under its own timeout
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$1; shift
groups=()
while [ "$1" != "--" ]; do groups+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp
i=0
for g in "${groups[@]}"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $g --output-format csv -d $R/gpurun_out/$out/p$i -- python "$@" > $R/gpurun_out/$out.p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $R/gpurun_out/$out.p$i.log | cut -c1-300; exit 1; }
done
echo done
