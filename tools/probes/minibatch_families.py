"""Mini-batch schedules on batches large enough for the hybrid dispatch: path=auto and path=fft against path=mfma.
Run on the GPU box:  python tools/probes/minibatch_families.py"""
import numpy as np, sys
sys.path.insert(0, '/root/repo')
from tnmf_amd.TransformInvariantNMF import TransformInvariantNMF, MiniBatchAlgorithm
rng = np.random.default_rng(0)
V = rng.random((48, 1, 128, 128)).astype(np.float32)
res = {}
for path in ('mfma', 'auto', 'fft'):
    for alg in (MiniBatchAlgorithm.Cyclic_MU, MiniBatchAlgorithm.ASG_MU, MiniBatchAlgorithm.GSAG_MU):
        np.random.seed(42)
        nmf = TransformInvariantNMF(n_atoms=16, atom_shape=(9, 9), backend='hip', path=path)
        nmf.fit_minibatches(V, algorithm=alg, batch_size=16, n_epochs=3, sag_lambda=0.8, sparsity_H=0.05)
        res[(path, alg.name)] = (nmf._energy_function(), nmf.W.copy(), nmf._backend.last_path)
for alg in ('Cyclic_MU', 'ASG_MU', 'GSAG_MU'):
    e0, w0, _ = res[('mfma', alg)]
    for path in ('auto', 'fft'):
        e, w, lp = res[(path, alg)]
        print(alg, path, 'last', lp, 'dE %.2e dW %.2e' % (abs(e - e0) / e0, np.abs(w - w0).max() / np.abs(w0).max()))
