"""Long-run agreement of the kernel-family choices (float32, 300 MU iterations on a sparse planted model):
energies along the way, sign of H, final W difference.  Run on the GPU box:  python tools/probes/long_run.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import tnmf_oracle as orc  # noqa: E402
from tnmf_amd.TransformInvariantNMF import TransformInvariantNMF  # noqa: E402


def main():
    N, C, D, M, A = 32, 1, (128, 128), 16, (9, 9)
    rng = np.random.default_rng(3)
    Hs = tuple(d + a - 1 for d, a in zip(D, A))
    Wt = rng.random((M, C) + A)
    Wt /= Wt.sum(axis=(-1, -2), keepdims=True)
    Ht = rng.random((N, M) + Hs) * (rng.random((N, M) + Hs) < 0.002)
    V = orc.reconstruct(Wt, Ht, 'c')
    V[:, :, :40, :] = 0            # a blank band: V == 0 exactly, so the model must drive H to zero there
    V = V.astype(np.float32)
    runs = {}
    for path in ('mfma', 'auto', 'fft'):
        np.random.seed(42)
        nmf = TransformInvariantNMF(n_atoms=M, atom_shape=A, backend='hip', path=path)
        nmf._prepare_fit(V) if hasattr(nmf, '_prepare_fit') else None
        E = []
        np.random.seed(42)
        nmf._initialize_matrices(V, keep_W=False)
        for it in range(300):
            nmf._update_H()
            nmf._update_W()
            if it in (0, 9, 49, 99, 199, 299):
                E.append(nmf._energy_function())
        H = nmf.H
        runs[path] = (E, nmf.W, H)
        print('%-5s energies %s  min(H) %.3e  negative entries %d  H==0 in blank band: %.4f' % (
            path, ' '.join('%.6g' % e for e in E), H.min(), int((H < 0).sum()),
            float((H[:, :, :30, :] <= 1e-30).mean())), flush=True)
        del nmf
        torch.cuda.empty_cache()
    for path in ('auto', 'fft'):
        dW = np.abs(runs[path][1] - runs['mfma'][1]).max() / np.abs(runs['mfma'][1]).max()
        dE = abs(runs[path][0][-1] - runs['mfma'][0][-1]) / runs['mfma'][0][-1]
        print('%-5s vs direct after 300 iterations: dW %.2e  dE %.2e' % (path, dW, dE))


if __name__ == '__main__':
    main()
