"""Probe: capture one MU iteration (fused H and W half steps through the C ABI) in a HIP graph and replay it.
Run on the GPU box:  python tools/probes/graph_probe.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tnmf_amd.TransformInvariantNMF import TransformInvariantNMF  # noqa: E402


def main():
    for (N, C, D, M, A) in ((64, 1, (128, 128), 16, (9, 9)), (256, 1, (256, 256), 32, (12, 12))):
        rng = np.random.default_rng(0)
        V = rng.random((N, C) + D).astype(np.float32)
        res = {}
        for mode in ('eager', 'graph'):
            np.random.seed(42)
            torch.cuda.manual_seed(1)
            nmf = TransformInvariantNMF(n_atoms=M, atom_shape=A, backend='hip', init='device')
            nmf._initialize_matrices(V, keep_W=False)

            def step():
                nmf._update_H()
                nmf._update_W()

            for _ in range(3):
                step()
            torch.cuda.synchronize()
            if mode == 'graph':
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    step()
                run = g.replay
            else:
                run = step
            run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                run()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 50
            res[mode] = (dt, nmf.W.copy(), nmf._energy_function())
            print('%s %s: %.3f ms/iteration  energy %.6g' % ((N, C, D, M, A), mode, dt * 1e3, res[mode][2]), flush=True)
        print('  W difference eager vs graph after the same number of iterations: %.2e' % (
            np.abs(res['eager'][1] - res['graph'][1]).max() / np.abs(res['eager'][1]).max()))


if __name__ == '__main__':
    main()
