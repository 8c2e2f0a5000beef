"""Randomised agreement check of the kernel families (float32): many small random geometries, every primitive and the
fused half steps of path='hybrid' and path='fft' against path='generic'.  Run on the GPU box:
    python tools/probes/fuzz_families.py [n_cases] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tnmf_amd.backends.HIP import HIP_Backend  # noqa: E402


def relmax(a, b):
    s = np.abs(b).max()
    return np.abs(a - b).max() / (s if s > 0 else 1.0)


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    worst = {}
    for case in range(n_cases):
        N, C, M = int(rng.integers(1, 6)), int(rng.integers(1, 4)), int(rng.integers(1, 41))
        A = (int(rng.integers(1, 17)), int(rng.integers(1, 17)))
        D = (int(rng.integers(max(2, A[0]), 200)), int(rng.integers(max(4, A[1]), 200)))
        Hs = tuple(d + a - 1 for d, a in zip(D, A))
        V = rng.random((N, C) + D).astype(np.float32)
        Wn = rng.random((M, C) + A).astype(np.float32)
        Wn /= Wn.sum(axis=(-2, -1), keepdims=True)
        Hn = rng.random((N, M) + Hs).astype(np.float32)
        out = {}
        for path in ('generic', 'hybrid', 'fft'):
            be = HIP_Backend(path=path)
            np.random.seed(1)
            be.initialize(V, A, M, None, (-2, -1))
            W, H = torch.from_numpy(Wn).cuda(), torch.from_numpy(Hn).cuda()
            R = be.reconstruct(W, H)
            nH, pH = be.reconstruction_gradient_H(V, W, H)
            nW, pW = be.reconstruction_gradient_W(V, W, H)
            Hf, Wf = H.clone(), W.clone()
            for _ in range(2):
                be.fused_update_H(V, Wf, Hf, slice(None), sparsity=0.01, eps=1e-9)
                be.fused_update_W(V, Wf, Hf, slice(None), eps=1e-9)
            out[path] = [x.cpu().numpy() for x in (R, nH, pH, nW, pW, Hf, Wf)]
            if path == 'hybrid':
                # the same calls on row-padded activations (rows of whole 128-byte lines): identical bits, whichever
                # kernel family the geometry lands on (families that want contiguous H get a copy through TNMF_E_STRIDE)
                ld = -(-Hs[1] // 32) * 32
                store = torch.zeros((N, M, Hs[0], ld), dtype=torch.float32, device='cuda')
                Hp = store[..., :Hs[1]]
                Hp.copy_(H)
                Wp = W.clone()
                same = torch.equal(be.reconstruct(W, Hp), be.reconstruct(W, H))
                gp = be.reconstruction_gradient_W(V, W, Hp)
                same = same and torch.equal(gp[0], nW) and torch.equal(gp[1], pW)
                for _ in range(2):
                    be.fused_update_H(V, Wp, Hp, slice(None), sparsity=0.01, eps=1e-9)
                    be.fused_update_W(V, Wp, Hp, slice(None), eps=1e-9)
                same = same and torch.equal(Hp, Hf) and torch.equal(Wp, Wf)
                if not same:
                    print('CASE', case, (N, C, D, M, A), 'row-padded activations differ from contiguous ones', flush=True)
            del be
        for path in ('hybrid', 'fft'):
            errs = [relmax(a, b) for a, b in zip(out[path], out['generic'])]
            names = ('R', 'negH', 'posH', 'negW', 'posW', 'H2', 'W2')
            for nm, e in zip(names, errs):
                key = (path, nm)
                if e > worst.get(key, (0, None))[0]:
                    worst[key] = (e, (N, C, D, M, A))
            bad = [(nm, e) for nm, e in zip(names, errs) if not np.isfinite(e) or e > (2e-2 if (path == 'fft' and nm == 'H2') else 1e-4)]
            if bad:
                print('CASE', case, (N, C, D, M, A), path, bad, flush=True)
    print('row-padded activations: compared bitwise with contiguous ones in every case (path hybrid)')
    for key in sorted(worst):
        print('%-6s %-5s worst %.2e at %s' % (key[0], key[1], worst[key][0], worst[key][1]))


if __name__ == '__main__':
    main()
