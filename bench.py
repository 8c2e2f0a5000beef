#!/usr/bin/env python3
"""
bench.py -- MU-iterations/sec of the shift-invariant multiplicative-update loop on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 3] [--path auto|generic|mfma|fft|hybrid]
                    [--no-cpu-baseline] [--no-fft-variant]
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
  roofline      dominant kernel group (by time) of the main leg; average launch duration from HIP events recorded on the
                launch stream inside the timed region.  A group on the matrix-core kernels is priced as algorithmic FLOP
                of the direct formulation per launch against 157.3 TFLOP/s (f32 MFMA = f32 vector); a group on the FFT
                family as the bytes that formulation must stream per launch against 8 TB/s HBM.
                `traffic` = HBM bytes per launch from the committed PMC passes (profiles/r01_traffic.json).
                roofline_by_kernel holds the same entry for every group.
  direct_variant, fft_variant   (--gpus 1 only) the same iterations from the same start with every group forced onto
                one kernel family (path='mfma' / path='fft'): speed relative to the main leg and max |dW| / max |W|.
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


def measured_traffic(kernel_name, cfg_id, family):
    """HBM bytes per launch of a kernel group from the committed PMC passes (profiles/r01_traffic.json, config 3)."""
    f = os.path.join(ROOT, 'profiles', 'r01_traffic.json')
    if cfg_id != 3 or not os.path.exists(f):
        return None
    try:
        entry = json.load(open(f))['kernels'][kernel_name]
        return entry['traffic_bytes'] if entry.get('family', 'mfma') == family else None
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
    ap.add_argument('--no-fft-variant', action='store_true', help='skip the extra timed legs on the other kernel families')
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

    F = conv_flops(cfg, n_local)
    Hs = tuple(d + a - 1 for d, a in zip(cfg['D'], cfg['A']))
    h_bytes = 4.0 * n_local * cfg['M'] * float(np.prod(Hs))
    t_bytes = 0.0
    if k == 2:   # row spectra of the FFT family: N*M*Hy*(Lx/2+1) complex64
        Lx = next((L for L in (32, 48, 64, 96, 144, 192, 288, 384, 576) if L >= Hs[1]), 0)
        t_bytes = 8.0 * n_local * cfg['M'] * Hs[0] * (Lx // 2 + 1)

    def group_roofline(name, avg_ms, paths):
        """Roofline entry of one kernel group: matrix-core groups against the f32 MFMA rate with the algorithmic flops of
        the direct formulation, FFT-family groups against HBM with the streams that formulation must move per launch."""
        fam = paths.get(name)
        flops = {'reconstruct': F, 'update_H': 2 * F, 'grad_W': 2 * F}.get(name)
        if flops is None or not avg_ms:
            return None
        if fam == 'fft':
            hybrid = paths.get('update_H') != 'fft'     # H changes outside the family: its row spectra are redone
            streams = {'reconstruct': t_bytes + (0.5 * (h_bytes + t_bytes) if hybrid else 0.0),
                       'update_H': 5 * t_bytes + 2 * h_bytes, 'grad_W': t_bytes}[name]
            gbs = streams / (avg_ms * 1e-3) / 1e9
            return {'kernel': name, 'family': fam, 'bound': 'hbm', 'achieved': gbs, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                    'frac': gbs / PEAK_HBM_GBS, 'bytes_per_launch': streams, 'avg_launch_ms': avg_ms,
                    'direct_equivalent_tflops': flops / (avg_ms * 1e-3) / 1e12}
        tf = flops / (avg_ms * 1e-3) / 1e12
        return {'kernel': name, 'family': fam, 'bound': 'mfma', 'achieved': tf, 'peak': PEAK_F32_TFLOPS, 'unit': 'TFLOP/s',
                'frac': tf / PEAK_F32_TFLOPS, 'flops_per_launch': flops, 'avg_launch_ms': avg_ms}

    def run_leg(path, pg):
        """args.warmup untimed + args.steps timed MU iterations from the fixed start on kernel family `path`."""
        np.random.seed(42)             # same W on every rank
        torch.cuda.manual_seed(4242 + rank)
        model = TransformInvariantNMF(n_atoms=cfg['M'], atom_shape=tuple(cfg['A']), backend='hip', device=device,
                                      path=path, init='device', process_group=pg)
        model._initialize_matrices(V, keep_W=False)
        b = model._backend

        def fence():
            if pg is not None:
                dist.barrier()
            torch.cuda.synchronize(device)

        for _ in range(args.warmup):
            model._update_H()
            model._update_W()
        fence()
        b.start_timeline()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            model._update_H()
            model._update_W()
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

    # Further legs (single GPU only): the same iterations from the same start on the other kernel families -- the
    # direct-vs-FFT crossover of BASELINE.json configs[4].
    variants = {}
    main_family = 'hybrid' if set(paths.values()) >= {'fft', 'mfma'} else be.last_path
    if world == 1 and not args.no_fft_variant and k == 2:
        W_main = nmf.W
        for vpath, label in (('mfma', 'direct_variant'), ('fft', 'fft_variant')):
            if vpath == main_family or args.path == vpath:
                continue
            try:
                m2, el2, spans2, paths2 = run_leg(vpath, None)
            except Exception as exc:  # noqa: BLE001   (family does not cover the shape)
                variants[label] = {'error': repr(exc)[:200]}
                continue
            ms2 = {name: float(np.mean(ms)) for name, ms in spans2.items()}
            variants[label] = {
                'value': args.steps / el2, 'unit': 'MU-iterations/sec', 'ms_per_step': el2 / args.steps * 1e3,
                'path': vpath, 'kernel_families': paths2, 'kernels_ms': ms2,
                'what': 'same data, same start, same iteration count, every kernel group forced onto this family',
                'W_max_rel_diff_vs_main': float(np.abs(m2.W - W_main).max() / np.abs(W_main).max()),
                'energy_after_run': m2._energy_function(),
                'speed_relative_to_main': (args.steps / el2) / (world * args.steps / elapsed),
                'roofline': [r for r in (group_roofline(nm, ms2.get(nm), paths2) for nm in ('reconstruct', 'update_H', 'grad_W')) if r],
            }
            del m2
            torch.cuda.empty_cache()

    if rank == 0:
        kernels = {}
        for name, ms in spans.items():
            kernels[name] = {'launches': len(ms), 'avg_ms': float(np.mean(ms)), 'total_ms': float(np.sum(ms)),
                             'family': paths.get(name)}
        rl = {name: group_roofline(name, kernels[name]['avg_ms'], paths) for name in kernels}
        rl = {n: r for n, r in rl.items() if r}
        for n, r in rl.items():
            kernels[n]['tflops_direct_equivalent'] = {'reconstruct': F, 'update_H': 2 * F, 'grad_W': 2 * F}[n] / (kernels[n]['avg_ms'] * 1e-3) / 1e12
        dom = max(rl, key=lambda n: kernels[n]['total_ms'])
        ms_per_step = elapsed / args.steps * 1e3
        roof = dict(rl[dom])
        roof['traffic'] = measured_traffic(dom, args.config, paths.get(dom)) if not args.samples else None
        line = {
            'metric': 'MU-iterations/sec', 'value': world * args.steps / elapsed, 'unit': 'MU-iterations/sec',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {
                'workload': f'2-D shift-invariant MU, full batch: {n_local} samples x {cfg["C"]} ch x '
                            f'{"x".join(map(str, cfg["D"]))} per GPU, {cfg["M"]} atoms '
                            f'{"x".join(map(str, cfg["A"]))} (BASELINE.json configs[{args.config - 1}])',
                'samples_per_gpu': n_local, 'global_samples': n_global, 'path': args.path, 'kernel_path': main_family,
                'kernel_families': paths,
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
            'roofline': roof,
            'roofline_by_kernel': rl,
        }
        line.update(variants)
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
