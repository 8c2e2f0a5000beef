"""Accuracy of the kernel-family choices against the float64 oracle after 5 MU iterations (float32 runs).
Run on the GPU box:  python tools/probes/path_accuracy.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import tnmf_oracle as orc  # noqa: E402
from tnmf_amd.TransformInvariantNMF import TransformInvariantNMF  # noqa: E402


def relmax(got, want):
    want = np.asarray(want, dtype=np.float64)
    return np.abs(np.asarray(got, dtype=np.float64) - want).max() / np.abs(want).max()


def main():
    shapes = [(8, 1, (64, 64), 8, (9, 9)), (4, 1, (96, 80), 32, (12, 12)), (3, 3, (48, 48), 32, (12, 12)),
              (2, 1, (256, 256), 32, (12, 12)), (2, 3, (128, 128), 16, (9, 9))]
    for N, C, D, M, A in shapes:
        rng = np.random.default_rng(11)
        Hs = tuple(d + a - 1 for d, a in zip(D, A))
        Wt = rng.random((M, C) + A)
        Ht = rng.random((N, M) + Hs) * (rng.random((N, M) + Hs) < 0.01)
        V = (orc.reconstruct(Wt, Ht, 'c') + 0.01 * rng.random((N, C) + D)).astype(np.float32)
        np.random.seed(42)
        ref = orc.OracleNMF(n_atoms=M, atom_shape=A, impl='c').fit(V.astype(np.float64), n_iterations=5)
        for path in ('auto', 'hybrid', 'fft'):
            np.random.seed(42)
            nmf = TransformInvariantNMF(n_atoms=M, atom_shape=A, backend='hip', path=path)
            nmf.fit(V, n_iterations=5, progress_callback=lambda *_: True)
            print('%-28s %-7s dW %.2e  dH %.2e  dE %.2e' % (f'{N}x{C}x{D} M{M} A{A}', path, relmax(nmf.W, ref.W),
                  relmax(nmf.H, ref.H), abs(nmf._energy_function() - ref.energy()) / ref.energy()), flush=True)


if __name__ == '__main__':
    main()
