"""Probe: does the mixed W gradient (VALU-issue bound) of one half of the samples overlap with the mixed reconstruct
(HBM bound) of the other half when the two are issued on two streams (two contexts, row spectra cached)?
Run on the GPU box:  python tools/probes/overlap_w_half_probe.py"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tnmf_amd.backends.HIP import HIP_Backend, _ptr  # noqa: E402


def main():
    N, C, D, M, A = 128, 1, (256, 256), 32, (12, 12)
    rng = np.random.default_rng(0)
    halves = []
    for i in range(2):
        V = rng.random((N, C) + D).astype(np.float32)
        be = HIP_Backend(init='device')
        np.random.seed(1)
        W, H = be.initialize(V, A, M, None, (-2, -1))
        be.fused_update_W(V, W.clone(), H, slice(None))     # row spectra of H and V are cached from here on
        halves.append((be, V, W, H, torch.empty_like(be._R_scratch), torch.empty_like(be._negpos)))
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()

    def recon(i, stream=None):
        be, V, W, H, R, np_ = halves[i]
        st = ctypes.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        g = be._geom(H.shape[0], W.shape[0], be._row_stride(H))
        assert be._lib.tnmf_hip_reconstruct(be._ctx, ctypes.byref(g), _ptr(W), _ptr(H), _ptr(R), st) == 0

    def gradw(i, stream=None):
        be, V, W, H, R, np_ = halves[i]
        st = ctypes.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        g = be._geom(H.shape[0], W.shape[0], be._row_stride(H))
        assert be._lib.tnmf_hip_grad_W_fused(be._ctx, ctypes.byref(g), _ptr(be._V_dev), _ptr(W), _ptr(H), _ptr(R), 1,
                                             _ptr(np_), st) == 0

    for _ in range(3):
        recon(0); gradw(0); recon(1); gradw(1)
    torch.cuda.synchronize()
    for i in range(2):
        print('cache counters', halves[i][0].cache_counters)

    def timed(fn, reps=20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    t_r = timed(lambda: recon(0))
    t_g = timed(lambda: gradw(1))

    def seq():
        recon(0)
        gradw(1)

    def conc():
        recon(0, sA)
        gradw(1, sB)

    t_seq = timed(seq)
    t_conc = timed(conc)
    print('reconstruct of half A alone %.3f ms; W gradient of half B alone %.3f ms' % (t_r, t_g))
    print('one after the other %.3f ms; on two streams %.3f ms  (ideal overlap %.3f ms)' % (t_seq, t_conc, max(t_r, t_g)))


if __name__ == '__main__':
    main()
