"""Probe: do the matrix-core H update of one half of the samples and the HBM-bound W-half-step kernels (row transform,
mixed reconstruct, mixed W gradient) of the other half overlap when issued on two streams with two contexts?
Run on the GPU box:  python tools/probes/overlap_probe.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tnmf_amd.backends.HIP import HIP_Backend  # noqa: E402


def main():
    N, C, D, M, A = 128, 1, (256, 256), 32, (12, 12)
    rng = np.random.default_rng(0)
    halves = []
    for i in range(2):
        V = rng.random((N, C) + D).astype(np.float32)
        be = HIP_Backend(init='device')
        np.random.seed(1)
        W, H = be.initialize(V, A, M, None, (-2, -1))
        halves.append((be, V, W, H))
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()

    def h_step(i):
        be, V, W, H = halves[i]
        be.fused_update_H(V, W, H, slice(None), sparsity=0., eps=1e-9)

    def w_step(i):
        be, V, W, H = halves[i]
        return be.local_gradient_W(V, W, H, slice(None))

    for _ in range(2):   # warm-up: workspaces, caches
        h_step(0); w_step(0); h_step(1); w_step(1)
    torch.cuda.synchronize()

    def timed(fn, reps=10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    t_h = timed(lambda: h_step(0))
    t_w = timed(lambda: w_step(1))

    def seq():
        h_step(0)
        w_step(1)

    def conc():
        with torch.cuda.stream(sA):
            h_step(0)
        with torch.cuda.stream(sB):
            w_step(1)

    t_seq = timed(seq)
    t_conc = timed(conc)
    print('H update of half A alone %.3f ms; W-half kernels of half B alone %.3f ms' % (t_h, t_w))
    print('one after the other %.3f ms; on two streams %.3f ms  (ideal overlap %.3f ms)' % (t_seq, t_conc, max(t_h, t_w)))


if __name__ == '__main__':
    main()
