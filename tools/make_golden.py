#!/usr/bin/env python3
"""
Generate the golden fixtures under tests/golden/ .  Runs ONLY in the build container (needs /root/reference and
the scipy sample image of the conda env); nothing under tests/, bench.py or the package reads /root/reference.

What it produces
----------------
primitives_<case>.npz   inputs (V, W, H in float64) and outputs of the *genuine* reference backend
                        ``tnmf.backends.PyTorch.PyTorch_Backend`` (imported from /root/reference, no stand-ins):
                        reconstruct, reconstruction_gradient_H/W (optionally on a slice), partial_reconstruct,
                        reconstruction_energy, convolve_multi_1d, and `initialize` under np.random.seed(42)
                        (pins the H-then-W draw order of backends/_Backend.py:92-95).
                        The reference's tests use this backend as their expected-factorisation fixture
                        (tnmf/tests/test_backends.py:53-56, test_minibatch.py:85-88).
modes_<mode>_<case>.npz the same primitives for reconstruction modes 'full', 'circular', 'reflect'
racoon_rgb_76x102.npz   the uint8 image behind tnmf/tests/test_backends.py:32-33 and test_sparsity_inhibition.py:55-56
                        (scipy's sample image `face`, PIL-resized to 0.1 scale exactly as utils/data_loading.py:8-12 does)
racoon_gray_patches.npz the 768 uint8 32x32 patches behind tnmf/tests/test_minibatch.py:35-45 / test_stream.py:29-39

The reference's NumPy backend needs opt_einsum (absent here, stays absent) and is not imported.
"""
import bz2
import os
import sys

import numpy as np

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
FACE_DAT = '/opt/conda/lib/python3.9/site-packages/scipy/misc/face.dat'
ONLY = ''


def reference_backend():
    sys.path.insert(0, REF)
    from tnmf.backends.PyTorch import PyTorch_Backend  # genuine reference code
    return PyTorch_Backend


def inhibition_kernels(rng):
    return tuple(1 - (np.arange(-i, i + 1) / (i + 1)) ** 2 for i in rng)


def make_primitives():
    import torch
    Backend = reference_backend()
    cases = {
        # name: (N, C, D, M, A, slice)
        '1d_c1': (3, 1, (15,), 3, (5,), None),
        '1d_c3': (4, 3, (33,), 5, (8,), None),
        '2d_c1': (3, 1, (17, 16), 6, (4, 4), None),
        '2d_c3': (2, 3, (20, 27), 4, (5, 7), None),
        '2d_slice': (5, 2, (13, 19), 3, (3, 6), (1, 4)),
        '2d_wide': (2, 1, (6, 37), 33, (2, 12), None),
        # three shift axes (volumes): the reference backend dispatches conv3d (backends/PyTorch.py:13-17)
        '3d_c1': (2, 1, (7, 9, 11), 3, (2, 3, 4), None),
        '3d_c2_slice': (3, 2, (6, 8, 10), 4, (3, 2, 3), (1, 3)),
    }
    for name, (N, C, D, M, A, sl) in cases.items():
        if ONLY and ONLY not in name:
            continue
        k = len(A)
        gen = np.random.default_rng(sum(map(ord, name)))
        V = gen.random((N, C) + D)
        be = Backend(reconstruction_mode='valid')
        np.random.seed(42)
        W0, H0 = be.initialize(V, A, M, None, tuple(range(-k, 0)))
        # independent random operands for the primitive checks (not the init values)
        Dp = tuple(d + a - 1 for d, a in zip(D, A))
        W = gen.random((M, C) + A)
        W /= W.sum(axis=tuple(range(-k, 0)), keepdims=True)
        H = gen.random((N, M) + Dp)
        Wt, Ht = torch.from_numpy(W), torch.from_numpy(H)
        s = slice(None) if sl is None else slice(*sl)
        R = be.reconstruct(Wt, Ht)
        nH, pH = be.reconstruction_gradient_H(V, Wt, Ht, s)
        nW, pW = be.reconstruction_gradient_W(V, Wt, Ht, s)
        E = be.reconstruction_energy(V, Wt, Ht)
        Rp = be.partial_reconstruct(Wt, Ht, M - 1)
        kern = inhibition_kernels(tuple(a - 1 for a in A))
        conv = be.convolve_multi_1d(Ht, kern, tuple(range(-k, 0)))
        np.savez_compressed(
            os.path.join(OUT, f'primitives_{name}.npz'),
            V=V, W=W, H=H, slice=np.array([-1, -1] if sl is None else sl),
            R=R.numpy(), neg_H=nH.numpy(), pos_H=pH.numpy(), neg_W=nW.numpy(), pos_W=pW.numpy(),
            energy=np.float64(E), R_partial_last=Rp.numpy(), inhibition_conv=conv.numpy(),
            init_W_seed42=W0.numpy(), init_H_seed42=H0.numpy())
        print(name, 'E =', E)


def make_mode_primitives():
    """Same outputs for the non-'valid' reconstruction modes (padding table: backends/_PyTorchBackend.py:42-52)."""
    import torch
    Backend = reference_backend()
    cases = {'1d': (3, 2, (17,), 3, (5,), (1, 3)), '2d': (2, 2, (13, 16), 3, (4, 5), None),
             '3d': (2, 2, (7, 8, 9), 3, (3, 2, 4), None)}
    for mode in ('full', 'circular', 'reflect'):
        for name, (N, C, D, M, A, sl) in cases.items():
            if ONLY and ONLY not in name:
                continue
            k = len(A)
            gen = np.random.default_rng(sum(map(ord, mode + name)))
            V = gen.random((N, C) + D)
            be = Backend(reconstruction_mode=mode)
            np.random.seed(42)
            W0, H0 = be.initialize(V, A, M, None, tuple(range(-k, 0)))
            W = gen.random((M, C) + A)
            W /= W.sum(axis=tuple(range(-k, 0)), keepdims=True)
            H = gen.random(tuple(H0.shape))
            Wt, Ht = torch.from_numpy(W), torch.from_numpy(H)
            s = slice(None) if sl is None else slice(*sl)
            R = be.reconstruct(Wt, Ht)
            nH, pH = be.reconstruction_gradient_H(V, Wt, Ht, s)
            nW, pW = be.reconstruction_gradient_W(V, Wt, Ht, s)
            np.savez_compressed(
                os.path.join(OUT, f'modes_{mode}_{name}.npz'),
                V=V, W=W, H=H, slice=np.array([-1, -1] if sl is None else sl),
                R=R.numpy(), neg_H=nH.numpy(), pos_H=pH.numpy(), neg_W=nW.numpy(), pos_W=pW.numpy(),
                energy=np.float64(be.reconstruction_energy(V, Wt, Ht)), init_H_shape=np.array(H0.shape))
            print(mode, name, 'H', tuple(H0.shape))


def load_face():
    with open(FACE_DAT, 'rb') as f:
        raw = bz2.decompress(f.read())
    return np.frombuffer(raw, dtype='uint8').reshape((768, 1024, 3))


def make_racoon():
    from PIL import Image
    face = load_face()
    # utils/data_loading.py:8-12 with gray=False, scale=0.1
    img = Image.fromarray(face)
    img = img.resize([int(0.1 * s) for s in img.size])
    rgb = np.array(img)
    assert rgb.shape == (76, 102, 3) and rgb.dtype == np.uint8
    np.savez_compressed(os.path.join(OUT, 'racoon_rgb_76x102.npz'), img=rgb)
    # gray=True (scipy 1.7.1 misc.face: 0.21 R + 0.71 G + 0.07 B -> uint8), scale=1 (PIL resize to same size = copy)
    gray = (0.21 * face[:, :, 0] + 0.71 * face[:, :, 1] + 0.07 * face[:, :, 2]).astype('uint8')
    # the reference's patch extraction (tests/test_minibatch.py:35-45) indexes the flat buffer with element strides
    # (768*32, 32, 768, 1) and shape (24, 32, 32, 32); restated with explicit flat indices:
    flat = gray.reshape(-1)
    i, j, y, x = np.ogrid[:24, :32, :32, :32]
    patches = flat[i * (768 * 32) + j * 32 + y * 768 + x].reshape(-1, 32, 32)
    assert patches.shape == (768, 32, 32)
    np.savez_compressed(os.path.join(OUT, 'racoon_gray_patches.npz'), patches=patches)


if __name__ == '__main__':
    # optional argument: only the cases whose name contains it (e.g. `3d`; the image fixtures are skipped then)
    ONLY = sys.argv[1] if len(sys.argv) > 1 else ''
    os.makedirs(OUT, exist_ok=True)
    make_primitives()
    make_mode_primitives()
    if not ONLY:
        make_racoon()
    print('written to', OUT)
