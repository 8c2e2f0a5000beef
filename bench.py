#!/usr/bin/env python3
"""
bench.py -- MU-iterations/sec of the shift-invariant multiplicative-update loop on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 3] [--path auto|generic|mfma] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one full MU iteration (H half step + W half step, TransformInvariantNMF.py:334-340 of the reference) over
the rank's resident samples, driven through the product front-end and the C ABI, with V, W, H already in HBM.
Workload: BASELINE.json configs[2] = 256 samples x 1 channel x 256x256, 32 atoms 12x12, float32, synthetic planted
model.  With N > 1 ranks every rank holds one such shard (weak scaling: 256 samples per GPU, global N = 256 * ranks);
the only exchange is one all-reduce (RCCL) of the 2 x 32 x 1 x 12 x 12 W numerator/denominator per iteration.

`value` = (shard-iterations completed by all ranks) / (max-over-ranks wall time), i.e. at --gpus 1 exactly the
MU-iterations/sec of the 256-sample problem and at --gpus N the iterations/sec of N such problems run as one job.

Extra objects on the JSON line:
  roofline      dominant kernel (by time): algorithmic FLOP per launch / average launch duration from HIP events
                recorded on the launch stream inside the timed region; peak = 157.3 TFLOP/s (f32 MFMA = f32 vector).
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
}
PEAK_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: f32 vector = f32 MFMA
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


def measured_traffic(kernel_name, cfg_id, path):
    """HBM bytes per launch from the committed PMC passes (profiles/r01_traffic.json), config 3 / MFMA path only."""
    f = os.path.join(ROOT, 'profiles', 'r01_traffic.json')
    if cfg_id != 3 or path != 'mfma' or not os.path.exists(f):
        return None
    try:
        return json.load(open(f))['kernels'][kernel_name]['traffic_bytes']
    except (KeyError, ValueError):
        return None


def cpu_baseline(cfg, budget_s=20.0):
    """Time the oracle's contraction form (the reference NumPy backend's algorithm) on a few samples of the workload."""
    from oracle import tnmf_oracle as orc
    rng = np.random.default_rng(5)
    k = len(cfg['A'])
    Dp = tuple(d + a - 1 for d, a in zip(cfg['D'], cfg['A']))

    def run(n):
        V = rng.random((n, cfg['C']) + tuple(cfg['D']), dtype=np.float32)
        W = rng.random((cfg['M'], cfg['C']) + tuple(cfg['A']), dtype=np.float32)
        W /= W.sum(axis=tuple(range(-k, 0)), keepdims=True)
        H = rng.random((n, cfg['M']) + Dp, dtype=np.float32)
        t0 = time.perf_counter()
        orc.mu_iteration_chunked(V, W, H, chunk=1)
        return time.perf_counter() - t0

    t1 = run(1)
    n = int(max(1, min(32, budget_s // max(t1, 1e-3))))   # ~budget_s seconds of host work
    t = run(n) if n > 1 else t1
    per_sample = t / n
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
                  f'(chunk=1; {per_sample:.2f} s/sample), scaled x{cfg["N"] / n:g}',
        'fft_variant': {'value': its_fft, 'unit': 'MU-iterations/sec',
                        'what': "FFT form of the same iteration (the reference's default 'numpy_fft' algorithm, "
                                'scipy.fft with workers=-1)',
                        'sample': f'{nf} of {cfg["N"]} samples ({tf / nf:.2f} s/sample), scaled x{cfg["N"] / nf:g}'},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', type=int, default=3, choices=sorted(CONFIGS))
    ap.add_argument('--samples', type=int, default=None, help='override samples per GPU (debug)')
    ap.add_argument('--path', default='auto', choices=['auto', 'generic', 'mfma', 'fft', 'hybrid'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-fft-variant', action='store_true', help='skip the second timed leg on the FFT kernel family')
    ap.add_argument('--cpu-budget', type=float, default=20.0)
    args = ap.parse_args()

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
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    group = None
    force_dist = os.environ.get('TNMF_BENCH_FORCE_DIST') == '1'   # exercise the RCCL path with a single rank
    if world > 1 or force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29531')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        dist.init_process_group('nccl', device_id=device)   # nccl == RCCL on ROCm
        group = dist.group.WORLD

    cfg = dict(CONFIGS[args.config])
    if args.samples:
        cfg['N'] = args.samples
    n_local = cfg['N']
    n_global = n_local * world
    k = len(cfg['A'])

    # this rank's shard of the synthetic data sits at [rank * n_local, (rank + 1) * n_local) of the global sample axis
    V_local = synth_V_on_device(cfg, n_local, 1234 + rank, device)
    if world > 1:
        V = np.zeros((n_global, cfg['C']) + tuple(cfg['D']), dtype=np.float32)
        V[rank * n_local:(rank + 1) * n_local] = V_local
    else:
        V = V_local

    np.random.seed(42)             # same W on every rank
    torch.cuda.manual_seed(4242 + rank)
    nmf = TransformInvariantNMF(n_atoms=cfg['M'], atom_shape=tuple(cfg['A']), backend='hip', device=device,
                                path=args.path, init='device', process_group=group)
    nmf._initialize_matrices(V, keep_W=False)
    be = nmf._backend

    def step():
        nmf._update_H()
        nmf._update_W()

    def fence():
        if group is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step()
    fence()
    be.start_timeline()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    spans = be.stop_timeline()

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if group is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    energy = nmf._energy_function()     # collective when sharded; outside the timed region

    # Second leg (single GPU only): the same iterations from the same start on the FFT kernel family -- the
    # frequency-domain formulation of the same update (BASELINE.json configs[4]: "FFT-vs-direct crossover").
    fft_variant = None
    if world == 1 and args.path != 'fft' and not args.no_fft_variant and k == 2:
        np.random.seed(42)
        torch.cuda.manual_seed(4242 + rank)
        nmf2 = TransformInvariantNMF(n_atoms=cfg['M'], atom_shape=tuple(cfg['A']), backend='hip', device=device,
                                     path='fft', init='device')
        nmf2._initialize_matrices(V, keep_W=False)
        for _ in range(args.warmup):
            nmf2._update_H()
            nmf2._update_W()
        torch.cuda.synchronize(device)
        nmf2._backend.start_timeline()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            nmf2._update_H()
            nmf2._update_W()
        torch.cuda.synchronize(device)
        el2 = time.perf_counter() - t1
        spans2 = nmf2._backend.stop_timeline()
        Wd, Wf = nmf.W, nmf2.W
        # the formulation's own HBM streams per iteration: row spectra T = N*M*Hy*(Lx/2+1) complex64
        # read 5x / written 3x, H read and written once, everything else is N*C- or M*C-sized
        Hy, Hx = (d + a - 1 for d, a in zip(cfg['D'], cfg['A']))
        Lx = next(L for L in (32, 48, 64, 96, 144, 192, 288, 384, 576) if L >= Hx)
        t_bytes = n_local * cfg['M'] * Hy * (Lx // 2 + 1) * 8
        h_bytes = n_local * cfg['M'] * Hy * Hx * 4
        fft_stream_bytes = 8 * t_bytes + 2 * h_bytes
        fft_variant = {
            'value': args.steps / el2, 'unit': 'MU-iterations/sec', 'ms_per_step': el2 / args.steps * 1e3,
            'kernel_path': nmf2._backend.last_path,
            'what': "same data, same start, same iteration count on path='fft' (own LDS transforms, fused contractions)",
            'kernels_ms': {name: float(np.mean(ms)) for name, ms in spans2.items()},
            'W_max_rel_diff_vs_direct': float(np.abs(Wf - Wd).max() / np.abs(Wd).max()),
            'energy_after_run': nmf2._energy_function(),
            'speedup_over_direct': (args.steps / el2) / (world * args.steps / elapsed),
            'roofline': {'bound': 'hbm', 'achieved': fft_stream_bytes / (el2 / args.steps) / 1e9, 'peak': PEAK_HBM_GBS,
                         'unit': 'GB/s', 'frac': fft_stream_bytes / (el2 / args.steps) / 1e9 / PEAK_HBM_GBS,
                         'stream_bytes_per_iteration': fft_stream_bytes,
                         'what': 'whole iteration: bytes the FFT formulation must stream (8 passes over the row '
                                 'spectra + H read/write) / iteration time'},
        }
        del nmf2

    if rank == 0:
        F = conv_flops(cfg, n_local)
        flops_per_launch = {'reconstruct': F, 'update_H': 2 * F, 'grad_W': 2 * F}
        kernels = {}
        for name, ms in spans.items():
            kernels[name] = {'launches': len(ms), 'avg_ms': float(np.mean(ms)), 'total_ms': float(np.sum(ms))}
            if name in flops_per_launch:
                kernels[name]['tflops'] = flops_per_launch[name] / (np.mean(ms) * 1e-3) / 1e12
        dom = max((n for n in kernels if n in flops_per_launch), key=lambda n: kernels[n]['total_ms'])
        achieved = kernels[dom]['tflops']
        ms_per_step = elapsed / args.steps * 1e3
        line = {
            'metric': 'MU-iterations/sec', 'value': world * args.steps / elapsed, 'unit': 'MU-iterations/sec',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {
                'workload': f'2-D shift-invariant MU, full batch: {n_local} samples x {cfg["C"]} ch x '
                            f'{"x".join(map(str, cfg["D"]))} per GPU, {cfg["M"]} atoms '
                            f'{"x".join(map(str, cfg["A"]))} (BASELINE.json configs[{args.config - 1}])',
                'samples_per_gpu': n_local, 'global_samples': n_global, 'kernel_path': be.last_path,
                'parallelism': f'sample-sharded x{world}, all-reduce of W num/den per iteration' if world > 1 else 'single GPU',
                'value_definition': 'shard-iterations completed by all ranks / max-over-ranks wall time',
                'energy_after_run': energy,
            },
            'iteration': {
                'flops_alg': 6 * F, 'bytes_alg': alg_bytes(cfg, n_local),
                'tflops': 6 * F / (ms_per_step * 1e-3) / 1e12, 'frac_f32_peak': 6 * F / (ms_per_step * 1e-3) / 1e12 / PEAK_F32_TFLOPS,
                'hbm_gbs_alg': alg_bytes(cfg, n_local) / (ms_per_step * 1e-3) / 1e9,
                'frac_hbm_peak': alg_bytes(cfg, n_local) / (ms_per_step * 1e-3) / 1e9 / PEAK_HBM_GBS,
            },
            'kernels': kernels,
            'roofline': {
                'kernel': dom, 'bound': 'mfma', 'achieved': achieved, 'peak': PEAK_F32_TFLOPS, 'unit': 'TFLOP/s',
                'frac': achieved / PEAK_F32_TFLOPS,
                'traffic': measured_traffic(dom, args.config, be.last_path) if not args.samples else None,
                'flops_per_launch': flops_per_launch[dom], 'avg_launch_ms': kernels[dom]['avg_ms'],
            },
        }
        if fft_variant is not None:
            line['fft_variant'] = fft_variant
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
