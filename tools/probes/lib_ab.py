"""A/B of library builds on bench.py's main leg: python tools/probes/lib_ab.py <lib1.so> <lib2.so> ... [-- bench args]
(one process per library and round, interleaved rounds; prints ms of the kernel groups)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
args = sys.argv[1:]
extra = []
if '--' in args:
    extra = args[args.index('--') + 1:]
    args = args[:args.index('--')]
code = ("import sys; sys.path.insert(0, %r); from tnmf_amd import _lib; _lib.LIB_PATH = sys.argv[1]; import bench; "
        "sys.argv = ['bench.py', '--steps', '20', '--warmup', '3', '--no-cpu-baseline', '--no-fft-variant', '--no-parity'] "
        "+ sys.argv[2:]; bench.main()") % ROOT
for rnd in range(2):
    for lib in args:
        path = lib if os.path.isabs(lib) else os.path.join(ROOT, 'tnmf_amd', 'lib', lib)
        out = subprocess.run([sys.executable, '-c', code, path] + extra, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            k = {n: round(v['avg_ms'], 3) for n, v in d['kernels'].items()}
            print(f'{lib:28s} round {rnd}: {d["value"]:.2f} it/s  {k}', flush=True)
        except Exception as exc:  # noqa: BLE001
            print(lib, 'failed', exc, out.stderr[-300:])
