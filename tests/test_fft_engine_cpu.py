"""Host check of the tile FFT engine of the HIP FFT kernel family (tnmf_amd/csrc/fft_engine.h).

The stage functions that the HIP kernels run per thread are plain host+device C++: tests/native/fft_engine_check.cpp
executes them task by task on the CPU for every supported transform length (float and double) and compares with a
naive O(L^2) DFT: forward transform incl. the digit-reversed output order, inverse round trip, and the split/merge of
two real rows carried as one complex sequence.  No GPU, no oracle needed.
"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fft_engine_matches_naive_dft(tmp_path):
    exe = str(tmp_path / 'fft_engine_check')
    src = os.path.join(ROOT, 'tests', 'native', 'fft_engine_check.cpp')
    # AddressSanitizer + UndefinedBehaviorSanitizer on the host build (GPU sanitizers are not available
    # on this pool): the stage functions index the tile with compile-time strides and digit-reversal tables -- an index
    # one past the tile, a signed overflow in a twiddle index or a misaligned access aborts the run
    subprocess.run(['g++', '-O1', '-g', '-std=c++17', '-fsanitize=address,undefined', '-fno-sanitize-recover=all',
                    '-I', os.path.join(ROOT, 'tnmf_amd', 'csrc'), src, '-o', exe], check=True)
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1')
    out = subprocess.run([exe], check=False, capture_output=True, text=True, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith('OK')
    # every transform length the library instantiates is covered
    lens = {int(line.split()[0][2:]) for line in out.stdout.splitlines() if line.startswith('L=')}
    makefile = open(os.path.join(ROOT, 'tnmf_amd', 'csrc', 'Makefile')).read()
    built = {int(t) for t in makefile.split('FFTLENS :=')[1].splitlines()[0].split()}
    assert lens == built
