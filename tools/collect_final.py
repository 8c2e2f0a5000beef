"""Copies the results of tools/final_measure.sh <tag> (gpurun_out/final_<tag>/) into profiles/ as r<NN>_*:
rocprof kernel stats and PMC traffic per configuration (what bench.py reads for roofline.traffic,
roofline.rocprof_avg_launch_ms and iteration.traffic_measured).
    python tools/collect_final.py <tag> <round>"""
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], int(sys.argv[2])
src = os.path.join(ROOT, 'gpurun_out', 'final_' + tag)
pre = os.path.join(ROOT, 'profiles', 'r%02d_' % rnd)
if os.path.exists(src + '/sources.sha16'):   # digest of the kernel sources the batch ran on (bench.py: *_is_current)
    shutil.copy(src + '/sources.sha16', pre + 'sources.sha16')
for f in sorted(glob.glob(src + '/traffic_config*.json') + glob.glob(src + '/rocprof_kernel_stats_config*.csv') +
                glob.glob(src + '/bench_*.json')):
    dst = pre + os.path.basename(f)
    shutil.copy(f, dst)
    print('->', os.path.relpath(dst, ROOT))
