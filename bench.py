#!/usr/bin/env python3
"""
bench.py -- MU-iterations/sec of the shift-invariant multiplicative-update loop on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 3] [--path auto|generic|mfma|split|fft|hybrid]
                    [--algorithm full|cyclic --batch-size B] [--no-cpu-baseline] [--no-fft-variant] [--no-parity]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one full MU iteration (H half step + W half step, TransformInvariantNMF.py:334-340 of the reference) over
the rank's resident samples, driven through the product front-end and the C ABI, with V, W, H already in HBM.
Workload: BASELINE.json configs[2] = 256 samples x 1 channel x 256x256, 32 atoms 12x12, float32, synthetic planted
model.  With N > 1 ranks every rank holds one such shard -- and only that: HIP_Backend(sharded_input=True) -- (weak scaling:
256 samples per GPU, global N = 256 * ranks); the only exchange is one all-reduce (RCCL) of the 2 x 32 x 1 x 12 x 12 W
numerator/denominator per iteration.

`value` = (shard-iterations completed by all ranks) / (max-over-ranks wall time), i.e. at --gpus 1 exactly the
MU-iterations/sec of the 256-sample problem and at --gpus N the iterations/sec of N such problems run as one job.

`python bench.py --gpus N` WITHOUT WORLD_SIZE in the environment starts the N ranks itself (a child process running
torch.distributed.run; this process never touches the GPU) and relays rank 0's line.  At N > 1 the line also carries
  strong_scaling   config 3 as ONE problem (256 global samples, 256 / N per GPU), iterations/sec of the global problem, with the
                   same problem on one GPU in the same run (rank 0 alone) and the measured speed-up / efficiency beside it
  config4_cyclic, config5_cyclic   BASELINE configs[3] / configs[4] as worded: Cyclic-MU epochs, sample-sharded (the per-GPU
                   shards 256 x 3 x 256^2 / 128 x 3 x 512^2 on every rank), ONE collective per epoch
  rccl_ranks_seen, distributed   an all-reduce of one 1.0 per rank (the ranks the collective reached), backend, latency of the
                   gradient exchange, rank / device / pid of every rank
(--legs selects them; TNMF_BENCH_DIST_BACKEND=gloo lets several ranks share the GPUs of a smaller box: a rehearsal.)

--algorithm cyclic: one step = one Cyclic-MU epoch (reference TransformInvariantNMF.py:457-465) over the rank's samples in
mini-batches of --batch-size (global batch = the union of the ranks' local batches; one all-reduce per epoch) -- the
way BASELINE.json words configs 4 and 5.  Same kernels, same work per step as a full-batch iteration.

Extra objects on the JSON line:
  roofline      the dominant single KERNEL of the step (the fused H update: one kernel per launch, the longest single kernel
                in profiles/rNN_rocprof_kernel_stats.csv at every BASELINE config), named as in the rocprof CSV; average
                launch duration from HIP events recorded on the launch stream inside the timed region.  `achieved` =
                SURVEY 8d's ALGORITHMIC flops (matrix-core kernels) or bytes (FFT-family kernels) of that primitive per
                launch / that duration: alg_flops = 2 F (both correlations, F = 2 N C M prod(A) prod(D)), alg_bytes =
                s (2 |H| + |V| + |R|) for the H half step; F and s (|H| + |R|) for a reconstruct; 2 F and
                s (|H| + |V| + |R|) for the W gradient.  `peak`: 157.3 TFLOP/s for kernels on the exact f32 MFMA; for the
                split kernel 2500 / 6 TFLOP/s -- the dense bf16 peak divided by the six bf16 products one float32-grade
                product costs (so frac = executed-but-unpadded bf16 flops / 2500); 8000 GB/s for HBM-bound kernels.
                Beside it: executed_flops (every MFMA the launch issues, tile padding included), formulation_stream_bytes
                (what the FFT formulation must stream), frac_hbm_peak (alg_bytes side), and `traffic` = HBM bytes per
                launch of that kernel from the committed PMC passes (profiles/rNN_traffic_config<c>.json), null if none.
                roofline_by_kernel prices every kernel group the same way; iteration.traffic_measured sums the PMC
                traffic of all kernels of a step.
  parity        the other half of BASELINE.json's metric: the same code path (same --path) run for 5 iterations from
                the reference's seeded start (np.random.seed(42), H drawn before W) on the first `samples` samples of
                the workload, against the float64 C oracle on the same samples: max |dW| / max |W|, max |dH| / max |H|,
                relative energy gap.  Outside the timed region.
  exact_f32_variant, direct_variant, fft_variant   (--gpus 1 only) the same iterations from the same start with the H
                update on the exact f32 MFMA (split off), with every group on the direct f32 kernels (path='mfma'), on
                the FFT family (path='fft'): speed relative to the main leg and max |dW| / max |W| against it.
  cpu_baseline  the CPU oracle ("port" of the reference NumPy backend's algorithm: windows + tensordot contraction)
                timed on a bounded sample of the same workload on this box's host cores (rank 0, --gpus 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # BASELINE.json "configs" (1-based here)
    1: dict(N=10, C=3, D=(60,), M=8, A=(20,)),
    2: dict(N=64, C=1, D=(128, 128), M=16, A=(9, 9)),
    3: dict(N=256, C=1, D=(256, 256), M=32, A=(12, 12)),
    4: dict(N=256, C=3, D=(256, 256), M=32, A=(12, 12)),   # per-GPU shard of the 2048-sample problem
    5: dict(N=128, C=3, D=(512, 512), M=64, A=(16, 16)),   # per-GPU shard of the 1024-sample problem
    # not a BASELINE config: config 3 with 245 x 245 samples, whose activation rows (256 floats) are whole cache lines --
    # the alignment experiment of DESIGN.md 4c
    6: dict(N=256, C=1, D=(245, 245), M=32, A=(12, 12)),
    # not a BASELINE config: long 1-D signals (config 1's kind of data at a size that is not launch-bound)
    7: dict(N=2048, C=3, D=(500,), M=32, A=(64,)),
    # not a BASELINE config: the geometry of the reference's mini-batch tests (tnmf/tests/test_minibatch.py:35-73: 768
    # patches 1 x 32 x 32, 10 atoms 7 x 7, batch_size 3) -- small-batch stochastic schedules are launch-latency bound
    8: dict(N=768, C=1, D=(32, 32), M=10, A=(7, 7)),
    # not a BASELINE config: volumes (three shift axes; the reference's PyTorch backend takes them through conv3d,
    # tnmf/backends/PyTorch.py:13-17) -- the direct kernels of tnmf_amd/csrc/volume.hip
    9: dict(N=16, C=1, D=(64, 64, 64), M=8, A=(5, 5, 5)),
}
PEAK_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: f32 vector = f32 MFMA
PEAK_BF16_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (AMD's 5 PF figure includes 2:1 sparsity)
PEAK_HBM_GBS = 8000.0


def conv_flops(cfg, n):
    """F = 2 * N * C * M * prod(A) * prod(D): one direct convolution/correlation over n samples."""
    return 2.0 * n * cfg['C'] * cfg['M'] * float(np.prod(cfg['A'])) * float(np.prod(cfg['D']))


def alg_bytes(cfg, n, itemsize=4):
    """bytes_alg = s * (3 * N*M*prod(D') + 2 * N*C*prod(D))  (SURVEY 8d)."""
    Dp = [d + a - 1 for d, a in zip(cfg['D'], cfg['A'])]
    return itemsize * (3.0 * n * cfg['M'] * float(np.prod(Dp)) + 2.0 * n * cfg['C'] * float(np.prod(cfg['D'])))


def synth_V_on_device(cfg, n_local, seed, device):
    """Planted shift-invariant model, generated with the product's own reconstruct kernel:
    V = reconstruct(W*, H*) + 0.01 U,  W* ~ U normalised,  H* = U * Bernoulli(0.01)."""
    import torch
    from tnmf_amd.backends.HIP import HIP_Backend
    k = len(cfg['A'])
    gen_state = torch.cuda.get_rng_state(device)
    torch.cuda.manual_seed(seed)
    be = HIP_Backend(device=device, init='device')
    zeros = np.zeros((n_local, cfg['C']) + tuple(cfg['D']), dtype=np.float32)
    Wt, Ht = be.initialize(zeros, tuple(cfg['A']), cfg['M'], None, tuple(range(-k, 0)))
    chunk = max(1, n_local // 8)
    for lo in range(0, n_local, chunk):
        Ht[lo:lo + chunk].mul_((torch.rand_like(Ht[lo:lo + chunk]) < 0.01).to(Ht.dtype))
    V = be.reconstruct(Wt, Ht)
    V.add_(0.01 * torch.rand_like(V))
    out = V.cpu().numpy()
    del be, Wt, Ht, V
    torch.cuda.empty_cache()
    torch.cuda.set_rng_state(gen_state, device)
    return out


# kernel group of the timeline -> kernel family -> (name of its main kernel as rocprof prints it, further kernels)
GROUP_KERNELS = {
    'update_H': {'split': ('k_split_corr_W', ()), 'mfma': ('k_mfma_corr_W', ()), 'generic': ('k_corr_W', ()),
                 'volume': ('k_vol_corr_W', ()),
                 'fft': ('k_fft_rows_mu', ('k_fft_grad_H', 'k_fft_rows_fwd', 'k_fft_cols_fwd'))},
    'reconstruct': {'fft': ('k_mix_reconstruct', ('k_fft_rows_fwd', 'k_fft_rows_inv', 'k_fft_contract_R',
                                                   'k_spectral_contract_R', 'k_fft_cols_fwd', 'k_fft_cols_inv')),
                    'mfma': ('k_mfma_reconstruct', ()), 'generic': ('k_reconstruct', ()),
                    'volume': ('k_vol_reconstruct', ())},
    'grad_W': {'fft': ('k_mix_grad_W', ('k_fft_grad_W', 'k_spectral_grad_W', 'k_fft_sum_groups', 'k_fft_rows_fwd')),
               'mfma': ('k_mfma_corr_H', ('k_corr_H_finalize',)), 'generic': ('k_corr_H', ('k_corr_H_finalize',)),
               'volume': ('k_vol_corr_H', ())},
}


def csrc_digest():
    """sha256 (first 16 hex digits) over the kernel sources the library is built from: what a committed profile was
    measured on (tools/final_measure.sh writes it beside the profiles) against what this run executes."""
    import glob
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, 'tnmf_amd', 'csrc')
    for f in sorted(glob.glob(os.path.join(src, '*.hip')) + glob.glob(os.path.join(src, '*.h')) +
                    [os.path.join(src, 'Makefile'), os.path.join(ROOT, 'include', 'tnmf_hip.h')]):
        if os.path.basename(f) == 'fft_len_twiddles.h':   # generated at build time
            continue
        h.update(os.path.basename(f).encode() + b'\0' + open(f, 'rb').read())
    return h.hexdigest()[:16]


def profile_is_current(path):
    """True / False: the sources digest recorded with the profile's round (profiles/rNN_sources.sha16) equals / differs
    from the sources of this run; None when that round recorded none (rounds 1-3)."""
    if not path:
        return None
    stamp = os.path.join(ROOT, 'profiles', os.path.basename(path)[:3] + '_sources.sha16')
    try:
        return open(stamp).read().split()[0] == csrc_digest()
    except (OSError, IndexError):
        return None


def committed_profile(cfg_id, what):
    """Newest committed profiles/rNN_<what>_config<c>.<ext> (config 3 also without the suffix)."""
    import glob
    pats = [f'r*_{what}_config{cfg_id}.*'] + ([f'r*_{what}.*'] if cfg_id == 3 else [])
    found = sorted((f for p in pats for f in glob.glob(os.path.join(ROOT, 'profiles', p))),
                   key=lambda f: os.path.basename(f)[:3], reverse=True)
    return found[0] if found else None


def measured_traffic(cfg_id):
    """{'file', 'iteration_bytes', 'per_kernel': {rocprof kernel name: {...}}} of the committed PMC passes, or None."""
    f = committed_profile(cfg_id, 'traffic')
    if not f or not f.endswith('.json'):
        return None
    try:
        d = json.load(open(f))
        if 'per_kernel' not in d:
            return None
        d['file'] = os.path.relpath(f, ROOT)
        return d
    except (ValueError, OSError):
        return None


def kernel_traffic(traffic, substr):
    """(rocprof name, bytes per launch) of the most expensive kernel whose name contains `substr`."""
    if not traffic:
        return None, None
    hits = [(k, v) for k, v in traffic['per_kernel'].items() if '::' + substr in k]
    if not hits:
        return None, None
    k, v = max(hits, key=lambda kv: kv[1]['bytes_per_iteration'])
    return k, v['bytes_per_launch']


def rocprof_average_ms(cfg_id, substr):
    """Average duration of the kernel in the committed rocprofv3 --kernel-trace --stats summary (ms), with the file."""
    import csv
    f = committed_profile(cfg_id, 'rocprof_kernel_stats')
    if not f:
        return None, None
    try:
        rows = [r for r in csv.DictReader(open(f)) if '::' + substr in r['Name']]
    except (OSError, KeyError):
        return None, None
    if not rows:
        return None, None
    r = max(rows, key=lambda r: float(r['TotalDurationNs']))
    return float(r['AverageNs']) * 1e-6, os.path.relpath(f, ROOT)


def cpu_baseline(cfg, budget_s=20.0):
    """Time the oracle's contraction form (the reference NumPy backend's algorithm) on a few samples of the workload."""
    from oracle import tnmf_oracle as orc
    rng = np.random.default_rng(5)
    k = len(cfg['A'])
    Dp = tuple(d + a - 1 for d, a in zip(cfg['D'], cfg['A']))

    def run(n, chunk):
        V = rng.random((n, cfg['C']) + tuple(cfg['D']), dtype=np.float32)
        W = rng.random((cfg['M'], cfg['C']) + tuple(cfg['A']), dtype=np.float32)
        W /= W.sum(axis=tuple(range(-k, 0)), keepdims=True)
        H = rng.random((n, cfg['M']) + Dp, dtype=np.float32)
        t0 = time.perf_counter()
        orc.mu_iteration_chunked(V, W, H, chunk=chunk)
        return time.perf_counter() - t0

    # sample chunks sized so that the im2col temporary of the contraction stays <= 8 GB (lets BLAS see a
    # large GEMM); chunk = 1 beside it; the faster of the two is the baseline
    im2col = 4.0 * float(np.prod(cfg['D'])) * cfg['M'] * float(np.prod(cfg['A'])) * max(1, cfg['C'])
    chunk8 = int(max(1, min(cfg['N'], (8 << 30) // im2col)))
    t1 = run(1, 1)
    n = int(max(1, min(32, (budget_s / 2) // max(t1, 1e-3))))   # ~budget_s / 2 seconds of host work per variant
    per_sample_1 = (run(n, 1) if n > 1 else t1) / n
    tried = {'chunk=1': per_sample_1}
    if chunk8 > 1:
        n8 = max(chunk8, (n // chunk8) * chunk8)
        tried[f'chunk={chunk8} (im2col <= 8 GB)'] = run(n8, chunk8) / n8
    how, per_sample = min(tried.items(), key=lambda kv: kv[1])
    its = 1.0 / (per_sample * cfg['N'])

    # second, stronger comparator: the reference's default backend 'numpy_fft' (FFT form restated in the oracle)
    def run_fft(nf):
        V = rng.random((nf, cfg['C']) + tuple(cfg['D']), dtype=np.float32)
        W = rng.random((cfg['M'], cfg['C']) + tuple(cfg['A']), dtype=np.float32)
        W /= W.sum(axis=tuple(range(-k, 0)), keepdims=True)
        H = rng.random((nf, cfg['M']) + Dp, dtype=np.float32)
        t0 = time.perf_counter()
        orc.mu_iteration_fft(V, W, H)
        return time.perf_counter() - t0

    tf1 = run_fft(2)
    nf = int(max(2, min(32, 2 * (budget_s / 2) // max(tf1, 1e-3))))
    tf = run_fft(nf) if nf > 2 else tf1
    its_fft = 1.0 / (tf / nf * cfg['N'])
    return {
        'value': its, 'unit': 'MU-iterations/sec', 'cores': os.cpu_count(), 'kind': 'port',
        'sample': f'{n} of {cfg["N"]} samples of the same workload, 1 MU iteration, float32, sample-chunked '
                  f'({how}; {per_sample:.2f} s/sample), scaled x{cfg["N"] / n:g}',
        'chunkings_s_per_sample': tried,
        'fft_variant': {'value': its_fft, 'unit': 'MU-iterations/sec',
                        'what': "FFT form of the same iteration (the reference's default 'numpy_fft' algorithm, "
                                'scipy.fft with workers=-1)',
                        'sample': f'{nf} of {cfg["N"]} samples ({tf / nf:.2f} s/sample), scaled x{cfg["N"] / nf:g}'},
    }


def launch_ranks(n):
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node n ... bench.py <same arguments>` as a child process;
    its stdout (rank 0's JSON line) goes to ours, its exit code is returned."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC: RCCL across processes needs it on this pool
    env.setdefault('OMP_NUM_THREADS', '8')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print('bench.py: starting', ' '.join(cmd), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', type=int, default=3, choices=sorted(CONFIGS))
    ap.add_argument('--samples', type=int, default=None, help='override samples per GPU (debug)')
    ap.add_argument('--path', default='auto', choices=['auto', 'generic', 'mfma', 'split', 'fft', 'hybrid'])
    ap.add_argument('--algorithm', default='full', choices=['full', 'cyclic', 'asg', 'gsg', 'asag', 'gsag'],
                    help='full: full-batch MU iterations; cyclic: Cyclic-MU epochs over mini-batches (configs 4, 5); '
                         'asg / gsg / asag / gsag: one epoch of the stochastic schedules (reference '
                         'TransformInvariantNMF.py:467-504) per step')
    ap.add_argument('--eager', action='store_true', help='stochastic schedules: drive every batch from Python '
                    '(one C-ABI call per half step) instead of one tnmf_hip_run_schedule call per epoch')
    ap.add_argument('--batch-size', type=int, default=None, help='global mini-batch size of --algorithm cyclic '
                    '(default: a quarter of the global sample count)')
    ap.add_argument('--inhibition', type=float, default=0., help='lateral inhibition strength of the H half step '
                    '(reference inhibition_strength; default range = atom size - 1)')
    ap.add_argument('--cross-inhibition', type=float, default=0., help='cross-atom inhibition strength')
    ap.add_argument('--reduce', default='all_reduce', choices=['all_reduce', 'ordered'],
                    help="cross-rank sum of the W gradient: one RCCL all-reduce, or all-gather + fixed rank-order sum")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-fft-variant', action='store_true', help='skip the extra timed legs on the other kernel families')
    ap.add_argument('--no-parity', action='store_true', help='skip the parity leg against the float64 oracle')
    ap.add_argument('--parity-samples', type=int, default=None)
    ap.add_argument('--cpu-budget', type=float, default=20.0)
    ap.add_argument('--legs', default='strong,config4,config5',
                    help='--gpus N > 1: further timed legs beside the weak-scaling headline, comma separated: strong '
                         '(config 3, 256 GLOBAL samples, 256 / N per GPU), config4 / config5 (Cyclic-MU epochs on the '
                         'per-GPU shards of BASELINE configs[3] / configs[4], one collective per epoch); "none" skips them')
    ap.add_argument('--leg-steps', type=int, default=None, help='timed steps of the further legs (default: min(steps, 20); '
                    'config5: min(steps, 8))')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # Started as plain `python bench.py --gpus N`: this process has not touched the GPU (torch is not even imported
        # yet) and stays that way -- the N ranks are CHILD processes under torch.distributed.run, one per GPU, and rank
        # 0's JSON line is relayed.  (Never os.exec* after a GPU call on this pool.)
        sys.exit(launch_ranks(args.gpus))

    # Only the result line may reach stdout: libraries (the RCCL version banner, ...) write to fd 1 from C, so fd 1 is
    # pointed at stderr for the whole run and the JSON line is written to the saved descriptor at the end.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from tnmf_amd.TransformInvariantNMF import TransformInvariantNMF

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)'
    assert torch.cuda.is_available(), 'bench.py needs a GPU; the hip backend has no CPU path'
    # one rank per GPU.  Rehearsal on a box with fewer GPUs than ranks (TNMF_BENCH_DIST_BACKEND=gloo): the ranks share the
    # visible devices and the collective goes through gloo -- RCCL refuses two ranks on one device
    dist_backend = os.environ.get('TNMF_BENCH_DIST_BACKEND', 'nccl')
    dev_index = local_rank if dist_backend == 'nccl' else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    group = None
    force_dist = os.environ.get('TNMF_BENCH_FORCE_DIST') == '1'   # exercise the RCCL path with a single rank
    if world > 1 or force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29531')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        if dist_backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)   # nccl == RCCL on ROCm
        else:
            dist.init_process_group(dist_backend)
        group = dist.group.WORLD

    cfg = dict(CONFIGS[args.config])
    if args.samples:
        cfg['N'] = args.samples
    n_local = cfg['N']
    n_global = n_local * world
    k = len(cfg['A'])

    # this rank's shard of the synthetic data sits at [rank * n_local, (rank + 1) * n_local) of the global sample axis
    # (every rank holds ONLY its own samples: HIP_Backend(sharded_input=True); the global array exists nowhere)
    V = synth_V_on_device(cfg, n_local, 1234 + rank, device)

    F = conv_flops(cfg, n_local)
    Hs = tuple(d + a - 1 for d, a in zip(cfg['D'], cfg['A']))
    h_bytes = 4.0 * n_local * cfg['M'] * float(np.prod(Hs))
    # row spectra of the FFT family: N*M*Hy*(Lx/2+1) complex64; 1-D signals: one row per (sample, atom)
    Lx = next((L for L in (32, 48, 64, 96, 144, 192, 270, 288, 384, 540, 576) if L >= Hs[-1]), 0)
    t_bytes = 8.0 * n_local * cfg['M'] * (Hs[0] if k == 2 else 1) * (Lx // 2 + 1)

    # samples one kernel launch processes, as a fraction of the rank's samples (Cyclic-MU launches work on one batch)
    batch_size_g = args.batch_size if args.batch_size else max(1, n_global // 4)
    launch_scale = 1.0 if args.algorithm == 'full' else min(1.0, -(-batch_size_g // world) / n_local)

    v_bytes = 4.0 * n_local * cfg['C'] * float(np.prod(cfg['D']))   # |V| = |R|
    traffic = measured_traffic(args.config) if not args.samples and args.algorithm == 'full' else None

    def split_executed_flops():
        """Every bf16 MFMA flop one launch of the split kernel issues (tile padding included): per 8 x 32-pixel tile,
        atom tile and channel, 4 waves x KB k-blocks x 2 rows x (V, R) x 6 products of 32 x 32 x 16."""
        if k == 1:   # 1-D instantiation: rows of a tile = eight samples, k blocks of 16 taps (atoms padded to 16 / 32 / 64)
            ax = cfg['A'][0]
            kb = (4 if ax <= 16 else 8 if ax <= 32 else 16) // 4
            tiles = -(-n_local // 8) * -(-Hs[0] // 32) * -(-cfg['M'] // 32) * cfg['C']
            return launch_scale * tiles * 4 * kb * 4 * 6 * (2.0 * 32 * 32 * 16)
        ay, ax = cfg['A']
        # the instantiation that runs the shape (split.hip: the covering (atom rows, runs of four taps) pair with the
        # fewest k blocks); 16 x 16 atoms, and 12 x 12 atoms with several channels, run on v_mfma_f32_16x16x32_bf16 (k blocks
        # of 32 = eight (row, run) slots)
        cover = [(a, r) for a, r in ((12, 3), (9, 3), (16, 4), (7, 2), (8, 2), (5, 2)) if a >= ay and 4 * r >= ax]
        if not cover:
            return None
        ai, ri = min(cover, key=lambda ar: (((ar[0] + 1) // 2) * ar[1] + 1) // 2)
        tiles = n_local * -(-Hs[0] // 8) * -(-Hs[1] // 32) * -(-cfg['M'] // 32) * cfg['C']
        if (ai, ri) == (16, 4) or ((ai, ri) == (12, 3) and cfg['C'] > 1):   # (12 x 12: with several channels only)
            kb32 = -(-ai * ri // 8)
            # per wave (two rows of the tile) and k block: 8 groups (row, V | R, pixel half) x 2 atom halves x 6 products
            return launch_scale * tiles * 4 * kb32 * 8 * 12 * (2.0 * 16 * 16 * 32)
        kb = (((ai + 1) // 2) * ri + 1) // 2
        return launch_scale * tiles * 4 * kb * 4 * 6 * (2.0 * 32 * 32 * 16)

    def group_roofline(name, avg_ms, paths):
        """Roofline entry of one kernel group, priced with SURVEY 8d's ALGORITHMIC work of the primitive per launch:
        matrix-core kernels by flops (f32 MFMA: 157.3 TFLOP/s; split kernel: the dense bf16 peak / 6 products per
        float32-grade product), FFT-family kernels by bytes against HBM."""
        fam = paths.get(name)
        alg_f = {'reconstruct': F, 'update_H': 2 * F, 'grad_W': 2 * F}.get(name)
        if alg_f is None or not avg_ms:
            return None
        alg_f *= launch_scale
        alg_b = launch_scale * {'reconstruct': h_bytes + v_bytes, 'update_H': 2 * h_bytes + 2 * v_bytes,
                                'grad_W': h_bytes + 2 * v_bytes}[name]
        sec = avg_ms * 1e-3
        main, _others = GROUP_KERNELS.get(name, {}).get(fam, (name, ()))
        kname, kbytes = kernel_traffic(traffic, main)
        r = {'kernel': kname or main, 'group': name, 'family': fam, 'avg_launch_ms': avg_ms,
             'alg_flops': alg_f, 'alg_bytes': alg_b,
             'alg_tflops': alg_f / sec / 1e12, 'alg_gbs': alg_b / sec / 1e9, 'frac_hbm_peak': alg_b / sec / 1e9 / PEAK_HBM_GBS,
             'traffic': kbytes}
        if fam == 'split':
            peak = PEAK_BF16_TFLOPS / 6.0
            ex = split_executed_flops() or 0.0
            r.update({'bound': 'mfma', 'achieved': r['alg_tflops'], 'peak': peak, 'unit': 'TFLOP/s',
                      'frac': r['alg_tflops'] / peak,
                      'peak_basis': 'dense bf16 MFMA peak (2500 TFLOP/s) / 6 bf16 products per float32-grade product '
                                    '(3 x bf16 operand splits)',
                      'executed_flops': ex, 'executed_tflops': ex / sec / 1e12,
                      'executed_frac_of_bf16_peak': ex / sec / 1e12 / PEAK_BF16_TFLOPS,
                      'alg_frac_of_f32_peak': r['alg_tflops'] / PEAK_F32_TFLOPS})
        elif fam == 'fft':
            hybrid = paths.get('update_H') != 'fft'     # H changes outside the family: its row spectra are redone
            streams = launch_scale * {'reconstruct': t_bytes + (0.5 * (h_bytes + t_bytes) if hybrid else 0.0),
                                      'update_H': 5 * t_bytes + 2 * h_bytes, 'grad_W': t_bytes}[name]
            r.update({'bound': 'hbm', 'achieved': r['alg_gbs'], 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                      'frac': r['frac_hbm_peak'], 'formulation_stream_bytes': streams,
                      'formulation_stream_gbs': streams / sec / 1e9,
                      'note': 'a group of several kernels (row transforms, contraction, inverse rows): the group time '
                              'prices the primitive; `traffic` is the main kernel alone'})
        else:
            r.update({'bound': 'mfma', 'achieved': r['alg_tflops'], 'peak': PEAK_F32_TFLOPS, 'unit': 'TFLOP/s',
                      'frac': r['alg_tflops'] / PEAK_F32_TFLOPS, 'executed_flops': alg_f})
        return r

    batch_size = batch_size_g

    def run_leg(path, pg, split=True, leg_cfg=None, leg_V=None, algorithm=None, batch=None, steps=None, warmup=None):
        """`warmup` untimed + `steps` timed steps from the fixed start on kernel family `path`.  A step is one
        full-batch MU iteration, or (algorithm cyclic) one Cyclic-MU epoch driven by the front end's epoch function.
        Defaults: the headline workload (cfg, V, --algorithm, --steps, --warmup); the further legs of an N > 1 run pass
        their own.  `leg_V` is THIS rank's block of samples."""
        leg_cfg = cfg if leg_cfg is None else leg_cfg
        leg_V = V if leg_V is None else leg_V
        algorithm = args.algorithm if algorithm is None else algorithm
        batch = batch_size if batch is None else batch
        steps = args.steps if steps is None else steps
        warmup = args.warmup if warmup is None else warmup
        np.random.seed(42)             # same W on every rank
        torch.cuda.manual_seed(4242 + rank)
        model = TransformInvariantNMF(n_atoms=leg_cfg['M'], atom_shape=tuple(leg_cfg['A']), backend='hip', device=device,
                                      path=path, init='device', process_group=pg, split=split, reduce=args.reduce,
                                      sharded_input=pg is not None)
        model._initialize_matrices(leg_V, keep_W=False)
        b = model._backend
        if algorithm != 'full':
            batches = b.minibatch_slices(batch)
            h_args = dict(sparsity=0., inhibition=args.inhibition, cross_inhibition=args.cross_inhibition)
            epoch_fn = getattr(model, '_epoch_' + algorithm)
            if args.eager:
                model._use_schedules = False
            state = [None]

            def step():
                state[0] = epoch_fn(state[0], batches, h_args, 0.8 if algorithm in ('asag', 'gsag') else 1.)
        else:
            it_args = dict(sparsity=0., inhibition=args.inhibition, cross_inhibition=args.cross_inhibition)
            if args.eager:
                model._use_schedules = False

            def step():
                model._iteration(it_args)

        def fence():
            if pg is not None:
                dist.barrier()
            torch.cuda.synchronize(device)

        for _ in range(warmup):
            step()
        fence()
        b.start_timeline()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        el = time.perf_counter() - t0
        paths = b.timeline_paths
        spans = b.stop_timeline()
        t = torch.tensor([el], dtype=torch.float64, device=device)
        if pg is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return model, float(t.item()), spans, paths

    nmf, elapsed, spans, paths = run_leg(args.path, group)
    be = nmf._backend
    energy = nmf._energy_function()     # collective when sharded; outside the timed region
    last_family = be.last_path

    # N > 1: evidence that the collective saw every rank, the latency of the one exchange step, and the further legs the
    # north star words (strong scaling of config 3; configs 4 and 5 as sample-sharded Cyclic-MU) -- all measured here,
    # none estimated.  Outside the headline's timed region.
    dist_info, scaling_legs = None, {}
    if group is not None:
        ones = torch.ones(1, dtype=torch.float32, device=device)
        dist.all_reduce(ones)           # every rank contributes 1: the sum is the number of ranks the collective reached
        buf = torch.zeros(2 * cfg['M'] * cfg['C'] * int(np.prod(cfg['A'])), dtype=torch.float32, device=device)
        for _ in range(5):
            be._all_reduce(buf)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(50):
            be._all_reduce(buf)
        torch.cuda.synchronize(device)
        exch_us = (time.perf_counter() - t0) / 50 * 1e6
        who = [None] * world
        dist.all_gather_object(who, {'rank': rank, 'device': dev_index, 'name': torch.cuda.get_device_name(dev_index),
                                     'pid': os.getpid()})
        dist_info = {'backend': dist.get_backend(), 'is_rccl': dist.get_backend() == 'nccl',
                     'rccl_ranks_seen': int(round(float(ones.item()))), 'world_size': world, 'reduce': args.reduce,
                     'what': 'all-reduce (sum) of one 1.0 per rank over the process group the W gradient uses',
                     'exchange_bytes': buf.numel() * 4, 'exchange_us_avg_of_50': exch_us, 'ranks': who}
        del buf, ones
        del nmf, be
        nmf = be = None
        torch.cuda.empty_cache()
        from tnmf_amd import sharding
        want = [] if args.legs in ('', 'none') or args.algorithm != 'full' else [x.strip() for x in args.legs.split(',')]
        ls = args.leg_steps or min(args.steps, 20)

        def leg_kernels(sp, pth):
            return {nm: {'launches': len(ms), 'avg_ms': float(np.mean(ms)), 'family': pth.get(nm)} for nm, ms in sp.items()}

        def guarded(name, fn):
            """One further leg; a failure on any rank (a shape the device cannot hold, ...) becomes an `error` entry on the
            line instead of taking the headline down with it.  (A rank that dies INSIDE a collective cannot be caught.)"""
            err, res = None, None
            try:
                res = fn()
            except Exception as exc:   # noqa: BLE001
                err = repr(exc)[:300]
            bad = torch.tensor([1.0 if err else 0.0], dtype=torch.float32, device=device)
            dist.all_reduce(bad)
            if float(bad.item()) > 0:
                torch.cuda.empty_cache()
                scaling_legs[name] = {'error': err or 'failed on another rank', 'ranks_failed': int(round(float(bad.item())))}
            else:
                scaling_legs[name] = res

        def leg_strong():
            # STRONG scaling of the headline problem: cfg['N'] (256) GLOBAL samples, ceil(256 / N) per GPU (this rank takes
            # them from the front of its own synthetic block), one W collective per iteration; beside it the same problem
            # on ONE GPU without a process group (rank 0, its whole block), in the same run.
            lo, hi = sharding.shard_bounds(cfg['N'], rank, world)
            m_s, el_s, sp_s, pth_s = run_leg(args.path, group, leg_V=V[:hi - lo], algorithm='full', steps=ls)
            n_glob_s = m_s._backend.n_samples
            e_s = m_s._energy_function()
            del m_s
            torch.cuda.empty_cache()
            el_1 = None
            if rank == 0:
                m_1, el_1, _sp1, _p1 = run_leg(args.path, None, leg_V=V, algorithm='full', steps=ls)
                del m_1
                torch.cuda.empty_cache()
            dist.barrier()
            return {
                'what': f'config 3 as ONE problem: {n_glob_s} global samples, {hi - lo} on rank 0, full-batch MU, one '
                        f'collective of the W numerator / denominator per iteration',
                'value': ls / el_s, 'unit': 'MU-iterations/sec (of the global problem)', 'scaling': 'strong',
                'global_samples': n_glob_s, 'samples_on_rank0': hi - lo, 'steps': ls, 'ms_per_step': el_s / ls * 1e3,
                'same_problem_on_one_gpu_same_run': {'value': ls / el_1, 'ms_per_step': el_1 / ls * 1e3,
                                                     'where': 'rank 0 alone, no process group'} if el_1 else None,
                'speedup_over_one_gpu': (el_1 / el_s) if el_1 else None,
                'efficiency': (el_1 / el_s / world) if el_1 else None,
                'kernels': leg_kernels(sp_s, pth_s), 'energy_after_run': e_s,
            }

        def leg_cyclic(cid, local_batch):
            # BASELINE configs[3] / configs[4]: mini-batch (Cyclic) MU, sample-sharded, ONE W collective per epoch; every
            # GPU holds the per-GPU shard of the 8-GPU problem (at N = 8 this IS the configuration; at N < 8 the same
            # shards, i.e. a problem of N / 8 of its size)
            c = dict(CONFIGS[cid])
            if args.samples:
                c['N'] = args.samples
                local_batch = max(1, args.samples // 2)
            steps_c = args.leg_steps or (min(args.steps, 20) if cid == 4 else min(args.steps, 8))
            Vc = synth_V_on_device(c, c['N'], 1234 + rank, device)
            m_c, el_c, sp_c, pth_c = run_leg(args.path, group, leg_cfg=c, leg_V=Vc, algorithm='cyclic',
                                             batch=local_batch * world, steps=steps_c, warmup=min(args.warmup, 2))
            e_c = m_c._energy_function()
            n_glob_c = m_c._backend.n_samples
            del m_c, Vc
            torch.cuda.empty_cache()
            return {
                'what': f'BASELINE.json configs[{cid - 1}] as worded: Cyclic-MU epochs (reference TransformInvariantNMF.py:'
                        f'457-465), sample-sharded, {c["N"]} samples x {c["C"]} ch x {"x".join(map(str, c["D"]))} per GPU, '
                        f'{c["M"]} atoms {"x".join(map(str, c["A"]))}, global batch {local_batch * world} = {local_batch} '
                        f'per GPU, ONE collective of the W numerator / denominator per epoch',
                'value': steps_c / el_c, 'unit': 'Cyclic-MU epochs/sec (of the global problem)',
                'scaling': 'weak (per-GPU shard fixed)', 'global_samples': n_glob_c, 'samples_per_gpu': c['N'],
                'global_batch': local_batch * world, 'steps': steps_c, 'ms_per_step': el_c / steps_c * 1e3,
                'sample_epochs_per_sec': n_glob_c * steps_c / el_c,
                'kernels': leg_kernels(sp_c, pth_c), 'energy_after_run': e_c,
            }

        if 'strong' in want:
            guarded('strong_scaling', leg_strong)
        for leg, cid, local_batch in (('config4', 4, 64), ('config5', 5, 32)):
            if leg in want:
                guarded(leg + '_cyclic', lambda cid=cid, local_batch=local_batch: leg_cyclic(cid, local_batch))

    # Further legs (single GPU only): the same iterations from the same start on the other kernel families -- the
    # direct-vs-FFT crossover of BASELINE.json configs[4].
    variants = {}
    fams = set(paths.values())
    main_family = ('hybrid' if 'fft' in fams and (fams & {'mfma', 'split', 'generic'}) else last_family)
    if 'split' in fams:
        main_family += '+split' if main_family != 'split' else ''
    if world == 1 and group is None and not args.no_fft_variant and k == 2:   # (with a process group the main model is gone)
        W_main = nmf.W
        legs = [('mfma', 'direct_variant', True), ('fft', 'fft_variant', True)]
        if 'split' in fams:
            legs.insert(0, (args.path, 'exact_f32_variant', False))   # same dispatch, H update on the exact f32 MFMA
        for vpath, label, vsplit in legs:
            if vsplit and (vpath == main_family or args.path == vpath):
                continue
            try:
                m2, el2, spans2, paths2 = run_leg(vpath, None, split=vsplit)
            except Exception as exc:  # noqa: BLE001   (family does not cover the shape)
                variants[label] = {'error': repr(exc)[:200]}
                continue
            ms2 = {name: float(np.mean(ms)) for name, ms in spans2.items()}
            variants[label] = {
                'value': args.steps / el2, 'unit': 'MU-iterations/sec', 'ms_per_step': el2 / args.steps * 1e3,
                'path': vpath, 'split': vsplit, 'kernel_families': paths2, 'kernels_ms': ms2,
                'what': 'same data, same start, same iteration count, every kernel group forced onto this family',
                'parity_scope': ('W and energy only (float32 transform noise in H: not a parity-grade H update)'
                                 if vpath == 'fft' else 'W, H and energy'),
                'W_max_rel_diff_vs_main': float(np.abs(m2.W - W_main).max() / np.abs(W_main).max()),
                'energy_after_run': m2._energy_function(),
                'speed_relative_to_main': (args.steps / el2) / (world * args.steps / elapsed),
                'roofline': [r for r in (group_roofline(nm, ms2.get(nm), paths2) for nm in ('reconstruct', 'update_H', 'grad_W')) if r],
            }
            del m2
            torch.cuda.empty_cache()

    # parity leg: the other half of BASELINE.json's metric (outside the timed region, rank 0, single GPU)
    parity = None
    if world == 1 and not args.no_parity:
        from oracle import tnmf_oracle as orc
        # ~3.2 thread-seconds per config-3 sample and iteration for the float64 C oracle: keep the leg near 10-20 s
        rel_cost = conv_flops(cfg, 1) / conv_flops(CONFIGS[3], 1)
        n_par = args.parity_samples or int(max(2, min(8, 8 // max(1.0, rel_cost / 3))))
        n_par = min(n_par, n_local)
        its_par = 5
        orc.set_threads(orc.default_threads(cap=64))
        Vp = np.ascontiguousarray(V[:n_par])
        np.random.seed(42)
        mp_ = TransformInvariantNMF(n_atoms=cfg['M'], atom_shape=tuple(cfg['A']), backend='hip', device=device,
                                    path=args.path)
        mp_.fit(Vp, n_iterations=its_par, progress_callback=lambda *_: True)
        np.random.seed(42)
        ref = orc.OracleNMF(n_atoms=cfg['M'], atom_shape=tuple(cfg['A']), impl='c').fit(Vp.astype(np.float64),
                                                                                       n_iterations=its_par)
        E_ref = ref.energy()
        parity = {
            'samples': n_par, 'iterations': its_par, 'path': args.path,
            'reference': 'float64 C oracle (oracle/tnmf_oracle_c.c, pinned to the reference), same seeds '
                         '(np.random.seed(42), H drawn before W), same first samples of the workload',
            'W_rel_diff_vs_oracle': float(np.abs(mp_.W - ref.W).max() / np.abs(ref.W).max()),
            'H_rel_diff_vs_oracle': float(np.abs(mp_.H - ref.H).max() / np.abs(ref.H).max()),
            'energy_gap_vs_oracle': float(abs(mp_._energy_function() - E_ref) / E_ref),
            'energy_oracle': float(E_ref),
        }
        del mp_, ref
        torch.cuda.empty_cache()

    if rank == 0:
        kernels = {}
        for name, ms in spans.items():
            kernels[name] = {'launches': len(ms), 'avg_ms': float(np.mean(ms)), 'total_ms': float(np.sum(ms)),
                             'family': paths.get(name)}
        rl = {name: group_roofline(name, kernels[name]['avg_ms'], paths) for name in kernels}
        rl = {n: r for n, r in rl.items() if r}
        for n, r in rl.items():
            kernels[n]['direct_equivalent_tflops'] = r['alg_tflops']
        # the dominant single kernel: the fused H update is ONE kernel per launch (plus a 5 us operand preparation) and
        # the longest single kernel of the step at every BASELINE config (profiles/rNN_rocprof_kernel_stats*.csv); the
        # FFT-family groups are several kernels each, none of them longer.  Under --path fft the H update is a group
        # too: the group with the largest total time is reported then.
        single = [n for n in rl if paths.get(n) in ('split', 'mfma', 'generic') and n == 'update_H']
        ms_per_step = elapsed / args.steps * 1e3
        if rl:
            dom = single[0] if single else max(rl, key=lambda n: kernels[n]['total_ms'])
            roof = dict(rl[dom])
            roof['dominance'] = ('longest single kernel of the step' if single else
                                 'kernel group with the largest total time (every group is several kernels on this path)')
            main_kernel = GROUP_KERNELS.get(dom, {}).get(paths.get(dom), (dom, ()))[0]
            prof_ms, prof_file = (rocprof_average_ms(args.config, main_kernel)
                                  if args.algorithm == 'full' and not args.samples else (None, None))
            roof['rocprof_avg_launch_ms'] = prof_ms
            roof['rocprof_file'] = prof_file
            # the two figures above that are READ from committed profiles (rocprof average, PMC traffic) were measured on
            # the kernel sources of that round: False = the sources have changed since (figures stale), None = unknown
            roof['rocprof_file_is_current'] = profile_is_current(prof_file)
            roof['traffic_file'] = traffic['file'] if traffic else None
            roof['traffic_file_is_current'] = profile_is_current(traffic['file']) if traffic else None
        else:
            # a whole epoch issued by one library call (tnmf_hip_run_schedule): no per-kernel events; the step is a chain
            # of tiny dependent launches, bound by launch latency, not by a roofline
            roof = {'kernel': 'tnmf_hip_run_schedule (whole epoch)', 'bound': 'latency', 'achieved': None, 'peak': None,
                    'unit': None, 'frac': None, 'traffic': None,
                    'alg_tflops': 6 * F / (ms_per_step * 1e-3) / 1e12}
        line = {
            'metric': 'MU-iterations/sec', 'value': world * args.steps / elapsed, 'unit': 'MU-iterations/sec',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {
                'workload': (f'{k}-D shift-invariant MU, full batch: ' if args.algorithm == 'full' else
                             f'{k}-D shift-invariant MU, {args.algorithm.upper()}-MU epochs (global batch {batch_size}'
                             f'{", driven batch by batch from Python" if args.eager else ""}): ') +
                            f'{n_local} samples x {cfg["C"]} ch x '
                            f'{"x".join(map(str, cfg["D"]))} per GPU, {cfg["M"]} atoms '
                            f'{"x".join(map(str, cfg["A"]))} (BASELINE.json configs[{args.config - 1}])',
                'samples_per_gpu': n_local, 'global_samples': n_global, 'path': args.path, 'kernel_path': main_family,
                'kernel_families': paths, 'algorithm': args.algorithm, 'inhibition': args.inhibition,
                'cross_inhibition': args.cross_inhibition,
                'batch_size': batch_size if args.algorithm != 'full' else None,
                'h_update_arithmetic': ('3 x bf16 operand splits on the bf16 matrix cores (float32-grade; parity '
                                        'object and exact_f32_variant beside it)' if 'split' in fams else
                                        'exact f32' if paths.get('update_H') in ('mfma', 'generic') else 'fft'),
                'parallelism': (f'sample-sharded x{world}, ' + ('all-reduce' if args.reduce == 'all_reduce' else
                                'all-gather + rank-order sum') + ' of W num/den per step') if world > 1 else 'single GPU',
                'value_definition': 'shard-iterations completed by all ranks / max-over-ranks wall time',
                'energy_after_run': energy,
            },
            'iteration': {
                # the direct formulation's flop count divided by this run's time: a cross-family yardstick, NOT a
                # roofline fraction (the FFT family executes a fraction of these flops, the split kernel 6x bf16 ones)
                'direct_equivalent_flops': 6 * F, 'bytes_alg': alg_bytes(cfg, n_local),
                'direct_equivalent_tflops': 6 * F / (ms_per_step * 1e-3) / 1e12,
                'hbm_gbs_alg': alg_bytes(cfg, n_local) / (ms_per_step * 1e-3) / 1e9,
                'frac_hbm_peak': alg_bytes(cfg, n_local) / (ms_per_step * 1e-3) / 1e9 / PEAK_HBM_GBS,
                # HBM bytes of one step summed over every kernel, from the committed PMC passes of this configuration
                'traffic_measured': traffic['iteration_bytes'] if traffic else None,
                'traffic_over_bytes_alg': traffic['iteration_bytes'] / alg_bytes(cfg, n_local) if traffic else None,
                'traffic_file': traffic['file'] if traffic else None,
                'traffic_file_is_current': profile_is_current(traffic['file']) if traffic else None,
                'sources_sha16': csrc_digest(),
            },
            'kernels': kernels,
            'roofline': roof,
            'roofline_by_kernel': rl,
        }
        line.update(variants)
        line.update(scaling_legs)
        if dist_info is not None:
            line['distributed'] = dist_info
            line['rccl_ranks_seen'] = dist_info['rccl_ranks_seen']
        if parity is not None:
            line['parity'] = parity
            line['energy_gap_vs_oracle'] = parity['energy_gap_vs_oracle']
            line['W_rel_diff_vs_oracle'] = parity['W_rel_diff_vs_oracle']
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(cfg, args.cpu_budget)
            line['cpu_baseline']['gpu_over_cpu'] = line['value'] / line['cpu_baseline']['value']
            line['cpu_baseline']['fft_variant']['gpu_over_cpu'] = line['value'] / line['cpu_baseline']['fft_variant']['value']
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(line) + '\n').encode())

    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
