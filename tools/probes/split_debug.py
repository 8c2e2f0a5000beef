"""Where does the split H gradient differ from the oracle? (debug helper)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import tnmf_oracle as orc
from tnmf_amd.backends.HIP import HIP_Backend
N, C, D, M, A = 2, 1, (int(sys.argv[1]), int(sys.argv[2])), 32, (12, 12)
rng = np.random.default_rng(N * 1000 + M)
V = rng.random((N, C) + D); Wn = rng.random((M, C) + A); Wn /= Wn.sum(axis=(-2, -1), keepdims=True)
Hn = rng.random((N, M) + tuple(d + a - 1 for d, a in zip(D, A)))
on, op = orc.gradient_H(V, Wn, Hn, slice(None), 'c')
be = HIP_Backend(path='split'); np.random.seed(1); be.initialize(V.astype(np.float32), A, M, None, (-2, -1))
W = torch.from_numpy(Wn.astype(np.float32)).cuda(); H = torch.from_numpy(Hn.astype(np.float32)).cuda()
neg, pos = be.reconstruction_gradient_H(V, W, H)
neg = neg.cpu().numpy()
err = np.abs(neg - on) / np.abs(on).max()
print('path', be.last_path, 'max err', err.max())
bad = err > 1e-4
print('bad fraction', bad.mean())
print('bad by sample', bad.mean(axis=(1, 2, 3)))
print('bad by atom', np.round(bad.mean(axis=(0, 2, 3)), 2))

bc = bad.mean(axis=(0, 1, 2)); print('bad cols', np.nonzero(bc > 0)[0][:60], 'bad rows', np.nonzero(bad.mean(axis=(0, 1, 3)) > 0)[0][:60])
i = np.argwhere(bad)[:5]
for idx in i: print(idx, neg[tuple(idx)], on[tuple(idx)])
