"""
Pins the CPU oracle (oracle/tnmf_oracle.py) against
  (1) outputs of the genuine reference PyTorch backend (tests/golden/primitives_*.npz, made by tools/make_golden.py),
  (2) the reference's own hard-coded known answers (file:line given per test).
CPU only.
"""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import tnmf_oracle as orc

CASES = sorted(glob.glob(os.path.join(GOLDEN, 'primitives_*.npz')))


def _slice(g):
    lo, hi = g['slice']
    return slice(None) if lo < 0 else slice(int(lo), int(hi))


@pytest.mark.parametrize('path', CASES, ids=[os.path.basename(p)[11:-4] for p in CASES])
def test_primitives_match_reference_backend(path):
    g = np.load(path)
    V, W, H, s = g['V'], g['W'], g['H'], _slice(g)
    A = W.shape[2:]
    k = len(A)
    tol = dict(rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(orc.reconstruct(W, H), g['R'], **tol)
    neg, pos = orc.gradient_H(V, W, H, s)
    np.testing.assert_allclose(neg, g['neg_H'], **tol)
    np.testing.assert_allclose(pos, g['pos_H'], **tol)
    neg, pos = orc.gradient_W(V, W, H, s)
    np.testing.assert_allclose(neg, g['neg_W'], **tol)
    np.testing.assert_allclose(pos, g['pos_W'], **tol)
    assert np.isclose(orc.energy(V, W, H), float(g['energy']), rtol=1e-13)
    np.testing.assert_allclose(orc.partial_reconstruct(W, H, W.shape[0] - 1), g['R_partial_last'], **tol)
    kern = orc.inhibition_kernels(tuple(a - 1 for a in A))
    np.testing.assert_allclose(orc.convolve_multi_1d(H, kern, range(-k, 0)), g['inhibition_conv'], **tol)
    # second, independent implementation agrees too
    np.testing.assert_allclose(orc.reconstruct_shiftsum(W, H), g['R'], **tol)
    np.testing.assert_allclose(orc.correlate_with_W_shiftsum(W, V[s]), g['neg_H'], **tol)
    np.testing.assert_allclose(orc.correlate_H_with_shiftsum(V[s], H[s], A), g['neg_W'], **tol)
    # the FFT form (the reference's default backend numpy_fft), bench.py's second CPU comparator
    if s == slice(None):
        np.testing.assert_allclose(orc.reconstruct_fft(W, H), g['R'], rtol=1e-9, atol=1e-10)
        neg, pos = orc.gradient_H_fft(V, W, H)
        np.testing.assert_allclose(neg, g['neg_H'], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(pos, g['pos_H'], rtol=1e-9, atol=1e-10)
        neg, pos = orc.gradient_W_fft(V, W, H)
        np.testing.assert_allclose(neg, g['neg_W'], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(pos, g['pos_W'], rtol=1e-9, atol=1e-9)
    # third: the C flavour (oracle/tnmf_oracle_c.c; one and two shift axes), in float64 and in float32 (double accumulation)
    if k == 3:
        return
    np.testing.assert_allclose(orc.reconstruct(W, H, 'c'), g['R'], **tol)
    neg, pos = orc.gradient_H(V, W, H, s, 'c')
    np.testing.assert_allclose(neg, g['neg_H'], **tol)
    np.testing.assert_allclose(pos, g['pos_H'], **tol)
    neg, pos = orc.gradient_W(V, W, H, s, 'c')
    np.testing.assert_allclose(neg, g['neg_W'], **tol)
    np.testing.assert_allclose(pos, g['pos_W'], **tol)
    f = np.float32
    np.testing.assert_allclose(orc.reconstruct(W.astype(f), H.astype(f), 'c'), g['R'], rtol=2e-6)
    neg, pos = orc.gradient_W(V.astype(f), W.astype(f), H.astype(f), s, 'c')
    np.testing.assert_allclose(neg, g['neg_W'], rtol=2e-6)
    np.testing.assert_allclose(pos, g['pos_W'], rtol=5e-6)


@pytest.mark.parametrize('path', CASES, ids=[os.path.basename(p)[11:-4] for p in CASES])
def test_seeded_init_draw_order(path):
    """H is drawn before W from the global legacy RNG (backends/_Backend.py:92-95)."""
    g = np.load(path)
    V, A, M = g['V'], g['W'].shape[2:], g['W'].shape[0]
    np.random.seed(42)
    W0, H0 = orc.init_matrices(V, A, M)
    assert np.array_equal(H0, g['init_H_seed42'])
    np.testing.assert_allclose(W0, g['init_W_seed42'], rtol=1e-15)


# ---------------------------------------------------------------------------------------------------------
# known answers held by the reference's tests
# ---------------------------------------------------------------------------------------------------------
# literal input of tnmf/tests/test_1d.py:32-36 (three periodic curves, singleton channel axis)
V_1D = np.array([[1., 2., 3., 2., 1., 1., 2., 3., 2., 1., 1., 2., 3., 2., 1.],
                 [1., 2., 2., 2., 1., 1., 2., 2., 2., 1., 1., 2., 2., 2., 1.],
                 [0., 1., 2., 3., 4., 0., 1., 2., 3., 4., 0., 1., 2., 3., 4.]])[:, np.newaxis, :]


def racoon_rgb_V():
    img = np.load(os.path.join(GOLDEN, 'racoon_rgb_76x102.npz'))['img'].astype(float) / 255
    return np.repeat(img.transpose((2, 0, 1))[np.newaxis, ...], 2, axis=0)     # tests/test_backends.py:32-33


def racoon_patches_V():
    p = np.load(os.path.join(GOLDEN, 'racoon_gray_patches.npz'))['patches'].astype(float) / 255
    return p[:, np.newaxis]                                                     # tests/test_minibatch.py:45


def test_known_answer_1d_with_inhibition():
    """tnmf/tests/test_1d.py:17-18,41-51: seed 42, 3 atoms of length 5, inhibition 0.1, 10 iterations -> 2.34946."""
    np.random.seed(42)
    nmf = orc.OracleNMF(n_atoms=3, atom_shape=(5,)).fit(V_1D, inhibition_strength=0.1, n_iterations=10)
    assert np.isclose(nmf.energy(), 2.34946)
    assert np.allclose(nmf.W.sum(axis=-1), 1.)


def test_known_answer_2d_rgb_sparsity():
    """tnmf/tests/test_backends.py:17-18,36-48: 2x3x76x102, 10 atoms 7x7, sparsity 0.1, 10 iterations -> 268.14423."""
    np.random.seed(42)
    nmf = orc.OracleNMF(n_atoms=10, atom_shape=(7, 7)).fit(racoon_rgb_V(), sparsity_H=0.1, n_iterations=10)
    assert np.isclose(nmf.energy(), 268.14423)


SPARSITY_INHIBITION_ROWS = [
    # (fit kwargs, ctor kwargs, energy, |H|_1, |H|_0)  -- tnmf/tests/test_sparsity_inhibition.py:20-52 (a subset)
    (dict(sparsity_H=0.0), dict(), 186.666013, 7704.38977, 176346),
    (dict(sparsity_H=1.0), dict(), 2429.69334, 2114.50047, 136396),
    (dict(inhibition_strength=0.5), dict(), 1831.92669, 3031.4130, 168931),
    (dict(inhibition_strength=1.0), dict(inhibition_range=(3, 3)), 1119.00855, 4657.19574, 168777),
    (dict(cross_atom_inhibition_strength=0.5), dict(inhibition_range=(3, 3)), 724.238350, 4953.89250, 175219),
]


@pytest.mark.parametrize('fit_kw,ctor_kw,E,l1,l0', SPARSITY_INHIBITION_ROWS)
def test_known_answer_sparsity_inhibition(fit_kw, ctor_kw, E, l1, l0):
    """tnmf/tests/test_sparsity_inhibition.py:58-84: 25 iterations, energy / L1 / L0 of H."""
    np.random.seed(42)
    nmf = orc.OracleNMF(n_atoms=10, atom_shape=(7, 7), impl='c', **ctor_kw).fit(racoon_rgb_V(), n_iterations=25, **fit_kw)
    H = nmf.H
    assert np.isclose(nmf.energy(), E)
    assert np.isclose(np.sum(np.abs(H)), l1)
    assert np.isclose(np.sum(H / H.max() > 1e-7), l0)


MINIBATCH_ROWS = [
    # tnmf/tests/test_minibatch.py:18-25
    ('full_batch', 14434.02658),
    (orc.MiniBatchAlgorithm.Cyclic_MU, 14434.02658),
    (orc.MiniBatchAlgorithm.ASG_MU, 4558.86695),
    (orc.MiniBatchAlgorithm.GSG_MU, 14223.14454),
    (orc.MiniBatchAlgorithm.ASAG_MU, 4560.03432),
    (orc.MiniBatchAlgorithm.GSAG_MU, 14310.92041),
]


@pytest.mark.parametrize('algorithm,E', MINIBATCH_ROWS, ids=[str(getattr(a, 'name', a)) for a, _ in MINIBATCH_ROWS])
def test_known_answer_minibatch(algorithm, E):
    """tnmf/tests/test_minibatch.py:48-76: 768 patches 1x32x32, 10 atoms 7x7, batch 3, 5 epochs, lambda 0.8."""
    V = racoon_patches_V()
    np.random.seed(42)
    nmf = orc.OracleNMF(n_atoms=10, atom_shape=(7, 7), impl='c')
    if algorithm == 'full_batch':
        nmf.fit_batch(V, sparsity_H=0.1, n_iterations=5)
    else:
        nmf.fit_minibatches(V, sparsity_H=0.1, algorithm=algorithm, batch_size=3, n_epochs=5, sag_lambda=0.8)
    assert np.isclose(nmf.energy(), E)


def test_known_answer_stream():
    """tnmf/tests/test_stream.py:19-25: fit_stream, subsample 50, ASAG-MU -> energy of the LAST subsample 96.7375921."""
    V = racoon_patches_V()
    np.random.seed(42)
    nmf = orc.OracleNMF(n_atoms=10, atom_shape=(7, 7), impl='c')
    nmf.fit(V, sparsity_H=0.1, algorithm=orc.MiniBatchAlgorithm.ASAG_MU, subsample_size=50, batch_size=3,
            n_epochs=5, sag_lambda=0.8)
    assert np.isclose(nmf.energy(), 96.7375921)


def test_known_answer_stream_limited():
    """tnmf/tests/test_stream.py:85-108: Cyclic-MU, max_subsamples=5 -> 629.109136."""
    V = racoon_patches_V()
    np.random.seed(42)
    nmf = orc.OracleNMF(n_atoms=10, atom_shape=(7, 7), impl='c')
    nmf.fit(V, sparsity_H=0.1, algorithm=orc.MiniBatchAlgorithm.Cyclic_MU, subsample_size=50, max_subsamples=5,
            batch_size=3, n_epochs=5, sag_lambda=0.8)
    assert np.isclose(nmf.energy(), 629.109136)


# ---------------------------------------------------------------------------------------------------------
# reconstruction modes 'full', 'circular', 'reflect' 
# ---------------------------------------------------------------------------------------------------------
MODE_CASES = sorted(glob.glob(os.path.join(GOLDEN, 'modes_*.npz')))


@pytest.mark.parametrize('path', MODE_CASES, ids=[os.path.basename(p)[6:-4] for p in MODE_CASES])
def test_mode_primitives_match_reference_backend(path):
    """Golden vectors of the genuine reference PyTorch backend in the non-'valid' modes (tools/make_golden.py)."""
    g = np.load(path)
    mode = os.path.basename(path).split('_')[1]
    V, W, H, s = g['V'], g['W'], g['H'], _slice(g)
    assert H.shape[2:] == orc.transform_shape(V.shape[2:], W.shape[2:], mode)
    tol = dict(rtol=1e-12, atol=1e-12)
    for impl in ('contract', 'c') if W.ndim - 2 < 3 else ('contract',):   # (the C flavour: one and two shift axes)
        np.testing.assert_allclose(orc.reconstruct(W, H, impl, mode), g['R'], **tol)
        neg, pos = orc.gradient_H(V, W, H, s, impl, mode)
        np.testing.assert_allclose(neg, g['neg_H'], **tol)
        np.testing.assert_allclose(pos, g['pos_H'], **tol)
        neg, pos = orc.gradient_W(V, W, H, s, impl, mode)
        np.testing.assert_allclose(neg, g['neg_W'], **tol)
        np.testing.assert_allclose(pos, g['pos_W'], **tol)
        assert np.isclose(orc.energy(V, W, H, impl, mode), float(g['energy']), rtol=1e-13)


@pytest.mark.parametrize('mode,E', [('full', 1.87180), ('circular', 3.13228), ('reflect', 3.16430)])
def test_known_answer_1d_modes(mode, E):
    """tnmf/tests/test_1d.py:17-22 (the reference keeps the 'reflect' value but does not run it, :27)."""
    np.random.seed(42)
    nmf = orc.OracleNMF(n_atoms=3, atom_shape=(5,), reconstruction_mode=mode)
    nmf.fit(V_1D, inhibition_strength=0.1, n_iterations=10)
    assert np.isclose(nmf.energy(), E)


@pytest.mark.parametrize('mode,E', [('full', 345.82498), ('circular', 265.35091)])
def test_known_answer_2d_rgb_modes(mode, E):
    """tnmf/tests/test_backends.py:17-22."""
    np.random.seed(42)
    nmf = orc.OracleNMF(n_atoms=10, atom_shape=(7, 7), impl='c', reconstruction_mode=mode)
    nmf.fit(racoon_rgb_V(), sparsity_H=0.1, n_iterations=10)
    assert np.isclose(nmf.energy(), E)
