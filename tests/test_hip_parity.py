"""
GPU parity tests proper: the 'hip' backend (hand-written gfx950 kernels behind the C ABI) against
  * the golden vectors of the genuine reference PyTorch backend (tests/golden/primitives_*.npz),
  * the pinned CPU oracle on seeded inputs (edge shapes, ragged tiles, empty slices),
  * the reference's own known-answer energies, run end to end through backend='hip' in float64.
Tolerances: float64 1e-10 relative; float32 per-primitive 2e-5 of the output's max (K up to ~5e3 f32 FMAs),
W after a fixed 5 iterations 1e-5 (north star), energy gap 1e-5.
"""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import tnmf_oracle as orc
from test_oracle_pinning import V_1D, racoon_patches_V, racoon_rgb_V

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from tnmf_amd.backends.HIP import HIP_Backend
    from tnmf_amd.TransformInvariantNMF import MiniBatchAlgorithm, TransformInvariantNMF

CASES = sorted(glob.glob(os.path.join(GOLDEN, 'primitives_*.npz')))
PATHS = ['generic', 'auto']   # 'auto' includes the split (3 x bf16) H update where it covers the shape


def dev(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).cuda()


def relmax(got, want):
    want = np.asarray(want, dtype=np.float64)
    scale = np.abs(want).max()
    return np.abs(np.asarray(got, dtype=np.float64) - want).max() / (scale if scale > 0 else 1.0)


def make_backend(V, A, M, path):
    be = HIP_Backend(path=path)
    k = len(A)
    np.random.seed(1)
    be.initialize(V, tuple(A), M, None, tuple(range(-k, 0)))
    return be


def _slice(g):
    lo, hi = g['slice']
    return slice(None) if lo < 0 else slice(int(lo), int(hi))


@pytest.mark.parametrize('path', PATHS)
@pytest.mark.parametrize('dtype,tol', [(np.float64, 1e-10), (np.float32, 2e-5)], ids=['f64', 'f32'])
@pytest.mark.parametrize('case', CASES, ids=[os.path.basename(p)[11:-4] for p in CASES])
def test_primitives_against_reference_golden(case, dtype, tol, path):
    g = np.load(case)
    V, s = g['V'].astype(dtype), _slice(g)
    A, M = g['W'].shape[2:], g['W'].shape[0]
    k = len(A)
    be = make_backend(V, A, M, path)
    W, H = dev(g['W'], dtype), dev(g['H'], dtype)
    assert relmax(be.to_ndarray(be.reconstruct(W, H)), g['R']) < tol
    neg, pos = be.reconstruction_gradient_H(V, W, H, s)
    assert neg.shape == H[s].shape
    assert relmax(be.to_ndarray(neg), g['neg_H']) < tol
    assert relmax(be.to_ndarray(pos), g['pos_H']) < tol
    neg, pos = be.reconstruction_gradient_W(V, W, H, s)
    assert neg.shape == W.shape
    assert relmax(be.to_ndarray(neg), g['neg_W']) < tol
    assert relmax(be.to_ndarray(pos), g['pos_W']) < tol
    assert abs(be.reconstruction_energy(V, W, H) - float(g['energy'])) / float(g['energy']) < tol
    assert relmax(be.to_ndarray(be.partial_reconstruct(W, H, M - 1)), g['R_partial_last']) < tol
    kern = orc.inhibition_kernels(tuple(a - 1 for a in A))
    assert relmax(be.to_ndarray(be.convolve_multi_1d(H, kern, tuple(range(-k, 0)))), g['inhibition_conv']) < tol
    assert be.last_path in ('generic', 'mfma', 'split', 'volume')


@pytest.mark.parametrize('case', CASES[:3], ids=[os.path.basename(p)[11:-4] for p in CASES[:3]])
def test_seeded_init_matches_reference(case):
    g = np.load(case)
    V, A, M = g['V'], g['W'].shape[2:], g['W'].shape[0]
    be = HIP_Backend()
    np.random.seed(42)
    W, H = be.initialize(V, tuple(A), M, None, tuple(range(-len(A), 0)))
    assert np.array_equal(be.to_ndarray(H), g['init_H_seed42'])
    np.testing.assert_allclose(be.to_ndarray(W), g['init_W_seed42'], rtol=1e-14)


SHAPES = [
    # N, C, D, M, A          (ragged tiles, M not a multiple of 32, wide atoms, long 1-D signals, single sample)
    (3, 1, (37, 45), 16, (9, 9)),
    (2, 3, (33, 31), 7, (5, 8)),
    (1, 1, (64, 64), 32, (12, 12)),
    (2, 2, (20, 70), 33, (16, 16)),
    (2, 1, (1000,), 8, (20,)),
    (5, 3, (257,), 3, (1,)),
    (2, 1, (5, 6), 2, (5, 6)),
]


@pytest.mark.parametrize('path', PATHS)
@pytest.mark.parametrize('dtype,tol', [(np.float64, 1e-10), (np.float32, 2e-5)], ids=['f64', 'f32'])
@pytest.mark.parametrize('shape', SHAPES, ids=[f'{s[0]}x{s[1]}x{"x".join(map(str, s[2]))}_m{s[3]}_a{"x".join(map(str, s[4]))}' for s in SHAPES])
def test_primitives_against_oracle(shape, dtype, tol, path):
    N, C, D, M, A = shape
    k = len(A)
    rng = np.random.default_rng(N * 1000 + M)
    V = rng.random((N, C) + D)
    Wn = rng.random((M, C) + A)
    Wn /= Wn.sum(axis=tuple(range(-k, 0)), keepdims=True)
    Hn = rng.random((N, M) + tuple(d + a - 1 for d, a in zip(D, A)))
    be = make_backend(V.astype(dtype), A, M, path)
    W, H = dev(Wn, dtype), dev(Hn, dtype)
    assert relmax(be.to_ndarray(be.reconstruct(W, H)), orc.reconstruct(Wn, Hn, 'c')) < tol
    for s in (slice(None), slice(0, 0), slice(N - 1, N)):
        on, op = orc.gradient_H(V, Wn, Hn, s, 'c')
        neg, pos = be.reconstruction_gradient_H(V, W, H, s)
        assert tuple(neg.shape) == on.shape
        if on.size:
            assert relmax(be.to_ndarray(neg), on) < tol and relmax(be.to_ndarray(pos), op) < tol
        on, op = orc.gradient_W(V, Wn, Hn, s, 'c')
        neg, pos = be.reconstruction_gradient_W(V, W, H, s)
        assert relmax(be.to_ndarray(neg), on) < tol and relmax(be.to_ndarray(pos), op) < tol
    # elementwise MU incl. the reference's in-place `pos += reg` side effect
    arr, neg, pos = dev(Hn, dtype), dev(rng.random(Hn.shape), dtype), dev(rng.random(Hn.shape), dtype)
    want_pos = be.to_ndarray(pos).copy() + dtype(0.25)
    want = (be.to_ndarray(arr) * be.to_ndarray(neg)) / want_pos
    be.multiplicative_update(arr, neg, pos, 0.25)
    assert np.array_equal(be.to_ndarray(pos), want_pos)
    np.testing.assert_allclose(be.to_ndarray(arr), want, rtol=1e-6 if dtype == np.float32 else 1e-14)
    # fused half steps == primitives + MU
    Hf = dev(Hn, dtype)
    be.fused_update_H(V, W, Hf, slice(None), sparsity=0.1, eps=1e-9)
    on, op = orc.gradient_H(V, Wn, Hn, slice(None), 'c')
    assert relmax(be.to_ndarray(Hf), Hn * on / (op + 1e-9 + 0.1)) < 2 * tol
    Wf = dev(Wn, dtype)
    be.fused_update_W(V, Wf, H, slice(None), eps=1e-9)
    on, op = orc.gradient_W(V, Wn, Hn, slice(None), 'c')
    Wo = Wn * on / (op + 1e-9)
    Wo /= Wo.sum(axis=tuple(range(-k, 0)), keepdims=True)
    assert relmax(be.to_ndarray(Wf), Wo) < 2 * tol


# ---------------------------------------------------------------------------------------------------------------
# end to end through backend='hip'
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('fused', [True, False], ids=['fused', 'unfused'])
def test_known_answer_1d_inhibition_f64(fused):
    np.random.seed(42)
    nmf = TransformInvariantNMF(n_atoms=3, atom_shape=(5,), backend='hip', use_fused_updates=fused)
    nmf.fit(V_1D, inhibition_strength=0.1, n_iterations=10)
    assert np.isclose(nmf._energy_function(), 2.34946)            # tnmf/tests/test_1d.py:18
    assert np.allclose(nmf.W.sum(axis=-1), 1.)


@pytest.mark.parametrize('fused', [True, False], ids=['fused', 'unfused'])
def test_known_answer_2d_rgb_f64(fused):
    V = racoon_rgb_V()
    np.random.seed(42)
    nmf = TransformInvariantNMF(n_atoms=10, atom_shape=(7, 7), backend='hip', use_fused_updates=fused)
    nmf.fit(V, sparsity_H=0.1, n_iterations=10)
    assert np.isclose(nmf._energy_function(), 268.14423)          # tnmf/tests/test_backends.py:18
    np.random.seed(42)
    ref = orc.OracleNMF(n_atoms=10, atom_shape=(7, 7), impl='c').fit(V, sparsity_H=0.1, n_iterations=10)
    assert np.allclose(nmf.W, ref.W) and np.allclose(nmf.H, ref.H)  # the reference's own acceptance criterion
    assert np.allclose(nmf.R, ref.R) and np.allclose(nmf.R_partial(0), ref.R_partial(0))
    assert np.allclose(nmf.W.sum(axis=(-1, -2)), 1.)


@pytest.mark.parametrize('fit_kw,ctor_kw,E,l1,l0', [
    # tnmf/tests/test_sparsity_inhibition.py:20-52 (subset)
    (dict(sparsity_H=1.0), dict(), 2429.69334, 2114.50047, 136396),
    (dict(inhibition_strength=1.0), dict(inhibition_range=(3, 3)), 1119.00855, 4657.19574, 168777),
    (dict(cross_atom_inhibition_strength=0.5), dict(inhibition_range=(3, 3)), 724.238350, 4953.89250, 175219),
])
def test_known_answer_sparsity_inhibition_f64(fit_kw, ctor_kw, E, l1, l0):
    np.random.seed(42)
    nmf = TransformInvariantNMF(n_atoms=10, atom_shape=(7, 7), backend='hip', **ctor_kw)
    nmf.fit(racoon_rgb_V(), n_iterations=25, **fit_kw)
    H = nmf.H
    assert np.isclose(nmf._energy_function(), E)
    assert np.isclose(np.sum(np.abs(H)), l1)
    assert np.isclose(np.sum(H / H.max() > 1e-7), l0)


@pytest.mark.parametrize('algorithm,E', [
    # tnmf/tests/test_minibatch.py:18-25
    ('full_batch', 14434.02658), ('Cyclic_MU', 14434.02658), ('ASG_MU', 4558.86695), ('GSG_MU', 14223.14454),
    ('ASAG_MU', 4560.03432), ('GSAG_MU', 14310.92041),
])
def test_known_answer_minibatch_f64(algorithm, E):
    V = racoon_patches_V()
    np.random.seed(42)
    nmf = TransformInvariantNMF(n_atoms=10, atom_shape=(7, 7), backend='hip')
    if algorithm == 'full_batch':
        nmf.fit_batch(V, sparsity_H=0.1, n_iterations=5)
    else:
        nmf.fit_minibatches(V, sparsity_H=0.1, algorithm=MiniBatchAlgorithm[algorithm], batch_size=3, n_epochs=5,
                            sag_lambda=0.8)
    assert np.isclose(nmf._energy_function(), E)


def test_known_answer_stream_f64():
    np.random.seed(42)
    nmf = TransformInvariantNMF(n_atoms=10, atom_shape=(7, 7), backend='hip')
    nmf.fit((v for v in racoon_patches_V()), sparsity_H=0.1, algorithm=MiniBatchAlgorithm.ASAG_MU, subsample_size=50,
            batch_size=3, n_epochs=5, sag_lambda=0.8)
    assert np.isclose(nmf._energy_function(), 96.7375921)         # tnmf/tests/test_stream.py:25


@pytest.mark.parametrize('path', PATHS + ['fft', 'hybrid', 'mfma', 'split'])
@pytest.mark.parametrize('N,C,D,M,A', [(8, 1, (64, 64), 8, (9, 9)), (4, 1, (96, 80), 32, (12, 12)), (3, 3, (48, 48), 32, (12, 12))])
def test_f32_loop_parity_with_f64_oracle(N, C, D, M, A, path):
    """North-star criterion: W within 1e-5 (max-relative) of the float64 reference after a fixed 5 iterations."""
    rng = np.random.default_rng(11)
    Wt = rng.random((M, C) + A)
    Ht = rng.random((N, M) + tuple(d + a - 1 for d, a in zip(D, A))) * (rng.random((N, M) + tuple(d + a - 1 for d, a in zip(D, A))) < 0.01)
    V = (orc.reconstruct(Wt, Ht, 'c') + 0.01 * rng.random((N, C) + D)).astype(np.float32)
    np.random.seed(42)
    nmf = TransformInvariantNMF(n_atoms=M, atom_shape=A, backend='hip', path=path)
    nmf.fit(V, n_iterations=5, progress_callback=lambda *_: True)
    np.random.seed(42)
    ref = orc.OracleNMF(n_atoms=M, atom_shape=A, impl='c').fit(V.astype(np.float64), n_iterations=5)
    assert relmax(nmf.W, ref.W) < 1e-5
    assert abs(nmf._energy_function() - ref.energy()) / ref.energy() < 1e-5
    # north star: H within 1e-5 as well -- on every path that claims parity on H.  path='fft' in float32 does NOT: it is
    # the opt-in for callers who need W and the energy only (HIP_Backend docstring, include/tnmf_hip.h): float32
    # transforms carry an absolute error of ~1e-7 of the largest gradient entry into every entry, so activations whose
    # gradients are tiny are relatively inexact (the reference's FFT backends share this in float32).  Its H is NOT held
    # to the parity bar -- but it is held to a regression guard: the measured error is up to 2e-3 of max|H| on these shapes,
    # and a broken fused FFT update or a stale spectrum cache (errors of tens of percent) must not pass as "a valid
    # factor".  5e-3 is that guard, not a parity claim; float64 transforms are held to 1e-10 elsewhere.
    if path == 'fft':
        assert np.isfinite(nmf.H).all() and (nmf.H >= 0).all()
        assert relmax(nmf.H, ref.H) < 5e-3
        # ... and the fused update, which runs on the spectra the W half step left in the cache, computes what the
        # family's own unfused gradient kernels compute from the same H on fresh storage (no cache): a stale spectrum
        # would show here at any size of error
        be = nmf._backend
        H0 = nmf.H
        nmf._update_H()
        assert be.last_path == 'fft'
        neg, pos = be.reconstruction_gradient_H(V, nmf._W, dev(H0, np.float32))
        own = H0.astype(np.float64) * be.to_ndarray(neg) / (be.to_ndarray(pos).astype(np.float64) + 1e-9)
        assert relmax(nmf.H, own) < 1e-5
    else:
        assert relmax(nmf.H, ref.H) < 1e-5


BASELINE_SHAPES = [
    # C, D, M, A   -- the geometries of BASELINE.json configs 2..5 (one sample each)
    (1, (128, 128), 16, (9, 9)),
    (1, (256, 256), 32, (12, 12)),
    (3, (256, 256), 32, (12, 12)),
    (3, (512, 512), 64, (16, 16)),
]


def _oracle_primitives(V, Wn, Hn, sparsity=0.05, eps=1e-9):
    """[R, neg_H, pos_H, neg_W, pos_W, H after the fused update] of the float64 C oracle (the referee at BASELINE sizes)."""
    V, Wn, Hn = (np.asarray(x, dtype=np.float64) for x in (V, Wn, Hn))
    orc.set_threads(orc.default_threads(cap=64))
    R = orc.reconstruct(Wn, Hn, 'c')
    nH, pH = orc.gradient_H(V, Wn, Hn, slice(None), 'c')
    nW, pW = orc.gradient_W(V, Wn, Hn, slice(None), 'c')
    return [R, nH, pH, nW, pW, Hn * nH / (pH + eps + sparsity)]


@pytest.mark.parametrize('C,D,M,A', BASELINE_SHAPES, ids=['config2', 'config3', 'config4', 'config5'])
def test_matrix_core_kernels_at_baseline_sizes(C, D, M, A):
    """At the BASELINE geometries every primitive must run on the MFMA kernels (path='mfma' has no fallback) and agree
    with the float64 C oracle (oracle/tnmf_oracle_c.c, pinned to the reference) -- not with another kernel family."""
    rng = np.random.default_rng(5)
    V = rng.random((1, C) + D).astype(np.float32)
    Wn = rng.random((M, C) + A).astype(np.float32)
    Hn = rng.random((1, M) + tuple(d + a - 1 for d, a in zip(D, A))).astype(np.float32)
    want = _oracle_primitives(V, Wn, Hn)
    be = make_backend(V, A, M, 'mfma')
    W, H = dev(Wn, np.float32), dev(Hn, np.float32)
    R = be.reconstruct(W, H)
    assert be.last_path == 'mfma'
    nH, pH = be.reconstruction_gradient_H(V, W, H)
    assert be.last_path == 'mfma'
    nW, pW = be.reconstruction_gradient_W(V, W, H)
    assert be.last_path == 'mfma'
    Hf = H.clone()
    be.fused_update_H(V, W, Hf, slice(None), sparsity=0.05, eps=1e-9)
    for got, ref in zip((R, nH, pH, nW, pW, Hf), want):
        assert relmax(be.to_ndarray(got), ref) < 2e-5


SPLIT_SHAPES = [
    # N, C, D, M, A -- every (atom rows, runs of four taps per atom row) pair split.hip instantiates, ragged tiles,
    # several channels (W image restaged per stage), M beyond one atom tile, odd atom heights (spare zero row)
    (2, 1, (70, 45), 32, (12, 12)),
    (1, 3, (37, 100), 40, (12, 10)),
    (3, 1, (33, 31), 16, (9, 9)),
    (2, 2, (20, 70), 33, (16, 16)),
    (1, 1, (64, 40), 5, (16, 13)),
    (2, 3, (50, 50), 10, (7, 7)),
    (1, 1, (9, 300), 3, (8, 5)),
    (2, 1, (40, 8), 64, (5, 8)),
    # atom shapes between the instantiations: they run on the smallest covering one (zero rows / taps in the operand image)
    (2, 1, (50, 60), 20, (10, 10)),
    (1, 2, (40, 44), 8, (6, 6)),
    (2, 1, (33, 70), 33, (13, 14)),
    (1, 1, (30, 30), 4, (11, 5)),
    (1, 1, (25, 40), 6, (3, 16)),
    (2, 1, (20, 20), 5, (3, 3)),
    # the 16x16x32 forms (round 4): 12-row instantiation with several channels on atoms it pads (odd height, taps not a
    # multiple of four), 16-row instantiation with one channel and a partial atom tile
    (2, 3, (30, 40), 12, (11, 9)),
    (1, 2, (26, 70), 35, (12, 12)),
    (2, 1, (40, 36), 20, (15, 13)),
]


@pytest.mark.parametrize('shape', SPLIT_SHAPES, ids=[f'{s[0]}x{s[1]}x{"x".join(map(str, s[2]))}_m{s[3]}_a{"x".join(map(str, s[4]))}' for s in SPLIT_SHAPES])
def test_split_h_gradient_against_oracle(shape):
    """path='split': the H gradient and the fused H update on the bf16 matrix cores (every f32 operand split exactly
    into three bf16 terms, six term products per product) against the float64 oracle -- held to the same bound as the
    exact f32 kernels, and not less accurate than them."""
    N, C, D, M, A = shape
    rng = np.random.default_rng(N * 1000 + M)
    V = rng.random((N, C) + D)
    Wn = rng.random((M, C) + A)
    Wn /= Wn.sum(axis=(-2, -1), keepdims=True)
    Hn = rng.random((N, M) + tuple(d + a - 1 for d, a in zip(D, A)))
    on, op = orc.gradient_H(V, Wn, Hn, slice(None), 'c')
    err = {}
    for path in ('split', 'mfma'):
        be = make_backend(V.astype(np.float32), A, M, path)
        W, H = dev(Wn, np.float32), dev(Hn, np.float32)
        neg, pos = be.reconstruction_gradient_H(V, W, H)
        assert be.last_path == path
        err[path] = max(relmax(be.to_ndarray(neg), on), relmax(be.to_ndarray(pos), op))
        assert err[path] < 2e-5
        for s in (slice(0, 0), slice(N - 1, N)):
            n2, p2 = be.reconstruction_gradient_H(V, W, H, s)
            assert tuple(n2.shape) == on[s].shape
            if on[s].size:
                assert relmax(be.to_ndarray(n2), on[s]) < 2e-5 and relmax(be.to_ndarray(p2), op[s]) < 2e-5
        Hf = dev(Hn, np.float32)
        be.fused_update_H(V, W, Hf, slice(None), sparsity=0.1, eps=1e-9)
        assert be.last_path == path
        assert relmax(be.to_ndarray(Hf), Hn * on / (op + 1e-9 + 0.1)) < 4e-5
    # pos carries the f32 error of R as well, so compare the families on the V correlation alone
    assert err['split'] < 2 * err['mfma'] + 1e-7


ADVERSARIAL_GEOMETRIES = {
    # id: (N, C, D, M, A, exact-f32 comparator family, lateral inhibition strength)
    'a12_c1': (2, 1, (48, 56), 32, (12, 12), 'mfma', 0.),          # four-wave <12,3> instantiation (round 2's case)
    'a16_c3': (2, 3, (40, 72), 40, (16, 16), 'mfma', 0.),          # eight-wave <16,4> workgroups, W image restaged per channel, partial atom tile
    'a64_1d': (12, 2, (300,), 20, (64,), 'generic', 0.),           # 1-D instantiation <1,16>: tile rows = samples, partial sample block
    'a12_extra': (2, 1, (48, 56), 32, (12, 12), 'mfma', 0.1),      # EXTRA-term epilogue: lateral inhibition in the fused kernel's denominator
}


def _adversarial_case(kind, geometry='a12_c1'):
    """Operands the uniform-random cases never produce (parity is thin on dynamic range; r3: ... and on the
    instantiations beyond 12 x 12 single-channel atoms)."""
    N, C, D, M, A = ADVERSARIAL_GEOMETRIES[geometry][:5]
    k = len(A)
    axes = tuple(range(-k, 0))
    rng = np.random.default_rng(97)
    Hs = tuple(d + a - 1 for d, a in zip(D, A))
    V = rng.random((N, C) + D)
    Wn = rng.random((M, C) + A)
    Hn = rng.random((N, M) + Hs)
    if kind == 'wide_V':
        # eight decades of dynamic range inside every window, exact zeros, a blank band
        V = 10.0 ** rng.uniform(-4, 4, size=V.shape) * (rng.random(V.shape) > 0.2)
        V[..., 20:26, :] = 0.0
        if k == 1:
            V[..., 100:130] = 0.0
    elif kind == 'tiny_W':
        # atoms with entries down to the smallest normal float32: whole rows, single taps, a whole atom; corners included
        Wn[0, ..., 3] = 1e-38
        Wn[1, :, ..., 5] = 3e-38
        if k == 2:
            Wn[0, :, 3] = 1e-38
            Wn[1, :, :, 5] = 3e-38
        Wn[2] = 10.0 ** rng.uniform(-38, -30, size=Wn[2].shape)
        Wn[3].reshape(C, -1)[:, -1] = 1.2e-38
        Wn[4].reshape(C, -1)[:, 0] = 0.0
        Wn[5] = 10.0 ** rng.uniform(-20, 0, size=Wn[5].shape)
    elif kind == 'sparse_H':
        # activations after 200 sparse MU iterations of the float64 oracle (most entries driven to ~0, a few large)
        orc.set_threads(orc.default_threads(cap=32))
        Vp = rng.random((N, C) + D) * (rng.random((N, C) + D) > 0.5)
        np.random.seed(5)
        ref = orc.OracleNMF(n_atoms=M, atom_shape=A, impl='c' if k == 2 else 'contract')
        ref.fit(Vp, n_iterations=200 if k == 2 and C == 1 else 60, sparsity_H=0.05)
        return N, C, D, M, A, Vp, ref.W.copy(), ref.H.copy()
    Wn /= Wn.sum(axis=axes, keepdims=True)
    return N, C, D, M, A, V, Wn, Hn


def _row_padded(Hc):
    """The values of a contiguous [N, M, Hy, Hx] tensor in storage whose rows are whole 128-byte lines (what initialize()
    allocates under the default dispatch; the EXTRA-term epilogue of the split kernel runs on this layout only)."""
    ld = -(-Hc.shape[3] // 32) * 32
    store = torch.zeros(tuple(Hc.shape[:3]) + (ld,), dtype=Hc.dtype, device=Hc.device)
    Hp = store[..., :Hc.shape[3]]
    Hp.copy_(Hc)
    return Hp


@pytest.mark.parametrize('kind', ['wide_V', 'tiny_W', 'sparse_H'])
@pytest.mark.parametrize('geometry', list(ADVERSARIAL_GEOMETRIES))
def test_split_h_update_on_adversarial_operands(kind, geometry):
    """The 3 x bf16 split H gradient / fused update against the float64 C oracle on operands with a wide dynamic range:
    never worse than twice the error of the exact f32 chain (f32-input MFMA; the generic f32 kernels for 1-D signals),
    measured against the output's maximum AND element by element (all terms are non-negative: an element's own value is
    the scale of its rounding error).  Every kind of instantiation of the split kernel: four-wave 12 x 12, eight-wave
    16 x 16 with three channels, 1-D, and the EXTRA-term epilogue (lateral inhibition in the fused denominator)."""
    import ctypes
    from tnmf_amd import _lib
    N, C, D, M, A, V, Wn, Hn = _adversarial_case(kind, geometry)
    exact, inhibition = ADVERSARIAL_GEOMETRIES[geometry][5:]
    k = len(A)
    impl = 'c' if k == 2 else 'contract'
    # float32 images of the operands are THE operands: the oracle sees what the kernels see
    V, Wn, Hn = (np.asarray(x, dtype=np.float32).astype(np.float64) for x in (V, Wn, Hn))
    on, op = orc.gradient_H(V, Wn, Hn, slice(None), impl)
    kernels = orc.inhibition_kernels(tuple(a - 1 for a in A))
    E = inhibition * (orc.convolve_multi_1d(Hn, kernels, range(-k, 0)) - Hn) if inhibition > 0 else 0.
    want_H = Hn * on / (op + E + 1e-9)

    def elementwise(got, want):
        want = np.asarray(want, dtype=np.float64)
        floor = 1e-12 * np.abs(want).max() + 1e-37        # (below ~1e-37 float32 itself has no bits left)
        return (np.abs(np.asarray(got, dtype=np.float64) - want) / (np.abs(want) + floor)).max()

    err = {}
    for path in ('split', exact):
        be = make_backend(V.astype(np.float32), A, M, path)
        W, H = dev(Wn, np.float32), dev(Hn, np.float32)
        neg = torch.empty_like(H)
        pos = torch.empty_like(H)
        # V correlation alone (pos also carries the f32 error of R): grad_H with the oracle's own R
        Rd = dev(orc.reconstruct(Wn, Hn, impl), np.float32)
        g = be._geom(N, M)
        _lib.check(be._lib.tnmf_hip_grad_H(be._ctx, ctypes.byref(g), ctypes.c_void_p(be._V_dev.data_ptr()),
                                           ctypes.c_void_p(Rd.data_ptr()), ctypes.c_void_p(W.data_ptr()),
                                           ctypes.c_void_p(H.data_ptr()), ctypes.c_void_p(neg.data_ptr()),
                                           ctypes.c_void_p(pos.data_ptr()), be._stream()), 'tnmf_hip_grad_H')
        assert be.last_path == path
        Hf = dev(Hn, np.float32)
        if inhibition > 0:
            Hf = _row_padded(Hf)
            be.fused_update_H(V, W, Hf, slice(None), sparsity=0., eps=1e-9, inhibition=inhibition,
                              inhibition_kernels=kernels)
        else:
            be.fused_update_H(V, W, Hf, slice(None), sparsity=0., eps=1e-9)
        assert be.last_path == path
        err[path] = dict(neg_max=relmax(be.to_ndarray(neg), on), neg_el=elementwise(be.to_ndarray(neg), on),
                         H_max=relmax(be.to_ndarray(Hf), want_H), H_el=elementwise(be.to_ndarray(Hf), want_H))
    print(kind, geometry, err)
    for key in ('neg_max', 'neg_el', 'H_max', 'H_el'):
        assert err['split'][key] <= 2 * err[exact][key] + 2.0 ** -22, (key, err)
    assert err['split']['neg_max'] < 2e-6 and err['split']['H_max'] < 2e-5


def test_non_finite_samples_propagate_on_both_h_update_kernels():
    """An infinite sample value (it passes the reference's `V >= 0` assert, TransformInvariantNMF.py:326) must come out
    as non-finite activations over its whole footprint -- never as finite garbage -- on the split and the f32 kernels."""
    N, C, D, M, A = 1, 1, (40, 40), 32, (12, 12)
    rng = np.random.default_rng(3)
    V = rng.random((N, C) + D).astype(np.float32)
    V[0, 0, 17, 23] = np.inf
    Wn = rng.random((M, C) + A)
    Wn /= Wn.sum(axis=(-2, -1), keepdims=True)
    Hn = rng.random((N, M) + tuple(d + a - 1 for d, a in zip(D, A)))
    foot = np.zeros(Hn.shape, dtype=bool)
    foot[:, :, 17:17 + A[0], 23:23 + A[1]] = True        # shifts u with u <= y <= u + A - 1 ... in padded coordinates
    for path in ('split', 'mfma'):
        be = make_backend(V, A, M, path)
        W, H = dev(Wn, np.float32), dev(Hn, np.float32)
        Rd = dev(orc.reconstruct(Wn, Hn, 'c'), np.float32)      # a finite R: only the V correlation sees the inf
        neg, pos = torch.empty_like(H), torch.empty_like(H)
        from tnmf_amd import _lib
        import ctypes
        g = be._geom(N, M)
        _lib.check(be._lib.tnmf_hip_grad_H(be._ctx, ctypes.byref(g), ctypes.c_void_p(be._V_dev.data_ptr()),
                                           ctypes.c_void_p(Rd.data_ptr()), ctypes.c_void_p(W.data_ptr()),
                                           ctypes.c_void_p(H.data_ptr()), ctypes.c_void_p(neg.data_ptr()),
                                           ctypes.c_void_p(pos.data_ptr()), be._stream()), 'tnmf_hip_grad_H')
        bad = ~np.isfinite(be.to_ndarray(neg))
        assert np.array_equal(bad, foot), (path, bad.sum(), foot.sum())
        assert np.isfinite(be.to_ndarray(pos)).all()


# ---------------------------------------------------------------------------------------------------------------
# FFT kernel family (path='fft'): the frequency-domain formulation must give the same numbers as the direct one
# ---------------------------------------------------------------------------------------------------------------
FFT_SHAPES = [
    # N, C, D, M, A, dtypes      transform lengths (y, x)
    (3, 1, (37, 45), 16, (9, 9), 'df'),        # 48, 64
    (2, 3, (33, 31), 7, (5, 8), 'df'),         # 48, 48
    (2, 2, (20, 70), 33, (16, 16), 'df'),      # 48, 96
    (2, 1, (5, 6), 2, (5, 6), 'df'),           # 32, 32 (atom as large as the sample)
    (1, 1, (64, 64), 32, (12, 12), 'df'),      # 96, 96
    (5, 5, (24, 40), 3, (3, 7), 'df'),         # 32, 48; more channels than one register group
    (2, 1, (40, 50), 4, (20, 6), 'df'),        # 64, 64; atoms taller than the mixed contractions take (16 rows)
    (2, 1, (128, 128), 16, (9, 9), 'df'),      # 144, 144
    (2, 3, (100, 170), 8, (12, 12), 'df'),     # 144, 192
    (1, 3, (256, 200), 8, (12, 12), 'df'),     # 288, 288
    (1, 1, (300, 500), 4, (16, 16), 'f'),      # 384, 576
]


@pytest.mark.parametrize('dtype,tol', [(np.float64, 1e-10), (np.float32, 2e-5)], ids=['f64', 'f32'])
@pytest.mark.parametrize('shape', FFT_SHAPES, ids=[f'{s[0]}x{s[1]}x{"x".join(map(str, s[2]))}_m{s[3]}_a{"x".join(map(str, s[4]))}' for s in FFT_SHAPES])
def test_fft_family_against_oracle(shape, dtype, tol):
    N, C, D, M, A, kinds = shape
    if ('d' if dtype == np.float64 else 'f') not in kinds:
        pytest.skip('float64 transforms are instantiated up to length 288')
    rng = np.random.default_rng(N * 1000 + M)
    V = rng.random((N, C) + D)
    Wn = rng.random((M, C) + A)
    Wn /= Wn.sum(axis=(-2, -1), keepdims=True)
    Hn = rng.random((N, M) + tuple(d + a - 1 for d, a in zip(D, A)))
    be = make_backend(V.astype(dtype), A, M, 'fft')
    W, H = dev(Wn, dtype), dev(Hn, dtype)
    assert relmax(be.to_ndarray(be.reconstruct(W, H)), orc.reconstruct(Wn, Hn, 'c')) < tol
    assert be.last_path == 'fft'
    for s in (slice(None), slice(N - 1, N)):
        on, op = orc.gradient_H(V, Wn, Hn, s, 'c')
        neg, pos = be.reconstruction_gradient_H(V, W, H, s)
        assert tuple(neg.shape) == on.shape
        assert relmax(be.to_ndarray(neg), on) < tol and relmax(be.to_ndarray(pos), op) < tol
        on, op = orc.gradient_W(V, Wn, Hn, s, 'c')
        neg, pos = be.reconstruction_gradient_W(V, W, H, s)
        assert relmax(be.to_ndarray(neg), on) < tol and relmax(be.to_ndarray(pos), op) < tol
        assert be.last_path == 'fft'
    # fused half steps (these keep the row spectra of H cached between the calls)
    Hf = dev(Hn, dtype)
    be.fused_update_H(V, W, Hf, slice(None), sparsity=0.1, eps=1e-9)
    on, op = orc.gradient_H(V, Wn, Hn, slice(None), 'c')
    Hnew = Hn * on / (op + 1e-9 + 0.1)
    assert relmax(be.to_ndarray(Hf), Hnew) < 2 * tol
    Wf = dev(Wn, dtype)
    be.fused_update_W(V, Wf, Hf, slice(None), eps=1e-9)       # uses the cached spectra of the updated H
    on, op = orc.gradient_W(V, Wn, Hnew, slice(None), 'c')
    Wo = Wn * on / (op + 1e-9)
    Wo /= Wo.sum(axis=(-2, -1), keepdims=True)
    assert relmax(be.to_ndarray(Wf), Wo) < 2 * tol
    be.fused_update_H(V, Wf, Hf, slice(None), sparsity=0., eps=1e-9)   # cached spectra again, new W
    on, op = orc.gradient_H(V, Wo, Hnew, slice(None), 'c')
    assert relmax(be.to_ndarray(Hf), Hnew * on / (op + 1e-9)) < 4 * tol


@pytest.mark.parametrize('scenario', ['rgb_full_batch', 'rgb_unfused', 'sparsity', 'inhibition', 'cross_inhibition',
                                      'Cyclic_MU', 'ASG_MU', 'GSAG_MU', 'stream'])
def test_fft_family_reproduces_reference_known_answers_f64(scenario):
    """The hard-coded energies of the reference's own tests (float64), this time through path='fft': full batch, the
    unfused front-end path (gradient primitives + elementwise MU), sparsity and inhibition terms, mini-batch schedules
    on slices of H (spectrum cache keyed by the slice) and the streaming fit."""
    np.random.seed(42)
    kw = dict(n_atoms=10, atom_shape=(7, 7), backend='hip', path='fft')
    if scenario in ('rgb_full_batch', 'rgb_unfused'):
        nmf = TransformInvariantNMF(use_fused_updates=scenario == 'rgb_full_batch', **kw)
        nmf.fit(racoon_rgb_V(), sparsity_H=0.1, n_iterations=10)
        E = 268.14423                                              # tnmf/tests/test_backends.py:18
    elif scenario in ('sparsity', 'inhibition', 'cross_inhibition'):
        fit_kw, ctor_kw, E = {                                     # tnmf/tests/test_sparsity_inhibition.py:20-52
            'sparsity': (dict(sparsity_H=1.0), dict(), 2429.69334),
            'inhibition': (dict(inhibition_strength=1.0), dict(inhibition_range=(3, 3)), 1119.00855),
            'cross_inhibition': (dict(cross_atom_inhibition_strength=0.5), dict(inhibition_range=(3, 3)), 724.238350),
        }[scenario]
        nmf = TransformInvariantNMF(**kw, **ctor_kw)
        nmf.fit(racoon_rgb_V(), n_iterations=25, **fit_kw)
    elif scenario == 'stream':
        nmf = TransformInvariantNMF(**kw)
        nmf.fit((v for v in racoon_patches_V()), sparsity_H=0.1, algorithm=MiniBatchAlgorithm.ASAG_MU,
                subsample_size=50, batch_size=3, n_epochs=5, sag_lambda=0.8)
        E = 96.7375921                                             # tnmf/tests/test_stream.py:25
    else:
        nmf = TransformInvariantNMF(**kw)
        nmf.fit_minibatches(racoon_patches_V(), sparsity_H=0.1, algorithm=MiniBatchAlgorithm[scenario], batch_size=3,
                            n_epochs=5, sag_lambda=0.8)
        E = {'Cyclic_MU': 14434.02658, 'ASG_MU': 4558.86695, 'GSAG_MU': 14310.92041}[scenario]   # test_minibatch.py:18-25
    assert nmf._backend.last_path == 'fft'
    assert np.isclose(nmf._energy_function(), E)


HYBRID_SHAPES = [
    # N, C, D, M, A -- ragged against every tile size of the row, mixed and column kernels
    (3, 1, (130, 301), 5, (7, 10)),       # transform lengths 144 x 384, one channel: mixed contractions
    (2, 2, (77, 45), 9, (16, 3)),         # 96 x 48, two channels: column-transform contractions
    (5, 1, (33, 500), 3, (1, 16)),        # atoms one row tall, long rows (576)
    (1, 3, (260, 40), 40, (13, 9)),       # 288 x 48, three channels, more atoms than one tile
    (7, 1, (20, 20), 33, (5, 5)),         # tiny planes, many of them
    (2, 1, (40, 50), 4, (20, 6)),         # atoms taller than 16 rows: column-transform contractions with one channel
]


@pytest.mark.parametrize('shape', HYBRID_SHAPES, ids=[f'{s[0]}x{s[1]}x{"x".join(map(str, s[2]))}_m{s[3]}_a{"x".join(map(str, s[4]))}' for s in HYBRID_SHAPES])
def test_hybrid_dispatch_on_ragged_shapes(shape):
    """path='hybrid' (forced, whatever the size) against the float64 oracle on shapes that are ragged against every
    tile size, with the fused half steps chained so that the cached spectra are exercised."""
    N, C, D, M, A = shape
    rng = np.random.default_rng(N * 100 + M)
    V = rng.random((N, C) + D)
    Wn = rng.random((M, C) + A)
    Wn /= Wn.sum(axis=(-2, -1), keepdims=True)
    Hn = rng.random((N, M) + tuple(d + a - 1 for d, a in zip(D, A)))
    be = make_backend(V.astype(np.float32), A, M, 'hybrid')
    W, H = dev(Wn, np.float32), dev(Hn, np.float32)
    tol = 2e-5
    assert relmax(be.to_ndarray(be.reconstruct(W, H)), orc.reconstruct(Wn, Hn, 'c')) < tol
    assert be.last_path == 'fft'
    on, op = orc.gradient_H(V, Wn, Hn, slice(None), 'c')
    neg, pos = be.reconstruction_gradient_H(V, W, H)
    assert be.last_path in ('mfma', 'generic', 'split')
    assert relmax(be.to_ndarray(neg), on) < tol and relmax(be.to_ndarray(pos), op) < tol
    on, op = orc.gradient_W(V, Wn, Hn, slice(None), 'c')
    neg, pos = be.reconstruction_gradient_W(V, W, H)
    assert be.last_path == 'fft'
    assert relmax(be.to_ndarray(neg), on) < tol and relmax(be.to_ndarray(pos), op) < tol
    # two chained iterations of fused half steps against the oracle's loop
    Hf, Wf = dev(Hn, np.float32), dev(Wn, np.float32)
    Ho, Wo = Hn.copy(), Wn.copy()
    for _ in range(2):
        be.fused_update_H(V, Wf, Hf, slice(None), sparsity=0., eps=1e-9)
        be.fused_update_W(V, Wf, Hf, slice(None), eps=1e-9)
        on, op = orc.gradient_H(V, Wo, Ho, slice(None), 'c')
        Ho = Ho * on / (op + 1e-9)
        on, op = orc.gradient_W(V, Wo, Ho, slice(None), 'c')
        Wo = Wo * on / (op + 1e-9)
        Wo /= Wo.sum(axis=(-2, -1), keepdims=True)
    assert relmax(be.to_ndarray(Hf), Ho) < 4 * tol and relmax(be.to_ndarray(Wf), Wo) < 4 * tol


ONE_D_SHAPES = [
    # N, C, D, M, A -- 1-D signals on the row-transform half of the FFT family (path='hybrid'): BASELINE config 1's
    # geometry (three channels, atoms of 20), one channel, two channels with a ragged transform length
    (10, 3, (60,), 8, (20,)),
    (2, 1, (500,), 4, (16,)),
    (3, 2, (301,), 5, (7,)),
    # several row blocks of eight samples with a ragged last one, 40 atoms (a partial second atom tile), 64-tap atoms
    (19, 3, (200,), 40, (64,)),
    (9, 1, (77,), 33, (33,)),
    (4, 2, (150,), 6, (70,)),      # atoms longer than the split kernel takes (64 taps): generic H update
]


@pytest.mark.parametrize('shape', ONE_D_SHAPES, ids=[f'{s[0]}x{s[1]}x{s[2][0]}_m{s[3]}_a{s[4][0]}' for s in ONE_D_SHAPES])
def test_one_dimensional_signals_on_the_fft_rows(shape):
    """1-D problems under path='hybrid': reconstruct and the W gradient are pointwise products of row spectra (own LDS
    FFT kernels, k_mix_reconstruct / k_mix_grad_W_1d), the H gradient / fused H update run on the bf16 matrix cores -- the
    split kernel's 1-D instantiation, eight samples per tile, atoms of up to 64 taps -- against the float64 oracle,
    with the fused half steps chained so that the cached spectra are exercised."""
    N, C, D, M, A = shape
    rng = np.random.default_rng(N * 100 + M)
    V = rng.random((N, C) + D)
    Wn = rng.random((M, C) + A)
    Wn /= Wn.sum(axis=-1, keepdims=True)
    Hn = rng.random((N, M, D[0] + A[0] - 1))
    be = make_backend(V.astype(np.float32), A, M, 'hybrid')
    W, H = dev(Wn, np.float32), dev(Hn, np.float32)
    tol = 2e-5
    assert relmax(be.to_ndarray(be.reconstruct(W, H)), orc.reconstruct(Wn, Hn, 'c')) < tol
    assert be.last_path == 'fft'
    on, op = orc.gradient_H(V, Wn, Hn, slice(None), 'c')
    neg, pos = be.reconstruction_gradient_H(V, W, H)
    assert be.last_path == ('split' if A[0] <= 64 else 'generic')
    assert relmax(be.to_ndarray(neg), on) < tol and relmax(be.to_ndarray(pos), op) < tol
    for s in (slice(0, 0), slice(N - 1, N), slice(1, N)):
        n2, p2 = be.reconstruction_gradient_H(V, W, H, s)
        assert tuple(n2.shape) == on[s].shape
        if on[s].size:
            assert relmax(be.to_ndarray(n2), on[s]) < tol and relmax(be.to_ndarray(p2), op[s]) < tol
    for s in (slice(None), slice(N - 1, N)):
        on, op = orc.gradient_W(V, Wn, Hn, s, 'c')
        neg, pos = be.reconstruction_gradient_W(V, W, H, s)
        assert be.last_path == 'fft'
        assert relmax(be.to_ndarray(neg), on) < tol and relmax(be.to_ndarray(pos), op) < tol
    Hf, Wf = dev(Hn, np.float32), dev(Wn, np.float32)
    Ho, Wo = Hn.copy(), Wn.copy()
    for _ in range(2):
        be.fused_update_H(V, Wf, Hf, slice(None), sparsity=0., eps=1e-9)
        assert be.last_path == ('split' if A[0] <= 64 else 'generic')
        be.fused_update_W(V, Wf, Hf, slice(None), eps=1e-9)
        on, op = orc.gradient_H(V, Wo, Ho, slice(None), 'c')
        Ho = Ho * on / (op + 1e-9)
        on, op = orc.gradient_W(V, Wo, Ho, slice(None), 'c')
        Wo = Wo * on / (op + 1e-9)
        Wo /= Wo.sum(axis=-1, keepdims=True)
    assert relmax(be.to_ndarray(Hf), Ho) < 4 * tol and relmax(be.to_ndarray(Wf), Wo) < 4 * tol


def test_kernel_families_agree_over_a_long_run_with_empty_regions():
    """120 float32 iterations on a sparse planted model with an exactly blank band (V == 0: both gradients vanish there,
    the hard case for a float32 frequency-domain update): the hybrid default and the pure FFT family must stay finite
    and non-negative, drive H to zero in the band and follow the energy trajectory of the direct kernels."""
    N, C, D, M, A = 16, 1, (128, 128), 16, (9, 9)
    rng = np.random.default_rng(3)
    Hs = tuple(d + a - 1 for d, a in zip(D, A))
    Wt = rng.random((M, C) + A)
    Wt /= Wt.sum(axis=(-1, -2), keepdims=True)
    Ht = rng.random((N, M) + Hs) * (rng.random((N, M) + Hs) < 0.002)
    V = orc.reconstruct(Wt, Ht, 'c')
    V[:, :, :40, :] = 0
    V = V.astype(np.float32)
    runs = {}
    for path in ('mfma', 'auto', 'fft'):
        np.random.seed(42)
        nmf = TransformInvariantNMF(n_atoms=M, atom_shape=A, backend='hip', path=path)
        nmf._initialize_matrices(V, keep_W=False)
        E = []
        for it in range(120):
            nmf._update_H()
            nmf._update_W()
            if it in (9, 59, 119):
                E.append(nmf._energy_function())
        runs[path] = (np.array(E), nmf.W, nmf.H)
    for path in ('auto', 'fft'):
        E, W, H = runs[path]
        assert np.all(np.isfinite(E)) and np.all(np.isfinite(W)) and np.all(np.isfinite(H))
        assert H.min() >= 0 and W.min() >= 0
        # exact zeros with the direct H update; the float32 transforms leave a little positive noise behind
        band = H[:, :, :30, :]
        assert np.mean(band == 0) > 0.999 if path == 'auto' else np.mean(band <= 1e-5 * H.max()) > 0.99
        np.testing.assert_allclose(E, runs['mfma'][0], rtol=1e-4)
        assert relmax(W, runs['mfma'][1]) < 1e-4


@pytest.mark.parametrize('C,D,M,A', BASELINE_SHAPES, ids=['config2', 'config3', 'config4', 'config5'])
def test_auto_dispatch_at_baseline_sizes(C, D, M, A):
    """path='auto' on float32 problems of this size is the hybrid dispatch: reconstruct and the W gradient on the FFT
    family, the H gradient and the fused H update on the matrix-core kernels -- and every result, the updated H and W
    included, agrees with the float64 C oracle at the tolerance of the direct path."""
    rng = np.random.default_rng(6)
    N = 16 if M == 16 else 4          # well above 2^19 activation entries, the threshold of the hybrid dispatch
    V = rng.random((N, C) + D).astype(np.float32)
    Wn = rng.random((M, C) + A).astype(np.float32)
    Hn = rng.random((N, M) + tuple(d + a - 1 for d, a in zip(D, A))).astype(np.float32)
    want = _oracle_primitives(V, Wn, Hn)
    nW, pW = orc.gradient_W(V.astype(np.float64), Wn.astype(np.float64), want[5], slice(None), 'c')
    Wo = Wn * nW / (pW + 1e-9)
    want.append(Wo / Wo.sum(axis=(-2, -1), keepdims=True))
    be = make_backend(V, A, M, 'auto')
    W, H = dev(Wn, np.float32), dev(Hn, np.float32)
    R = be.reconstruct(W, H)
    assert be.last_path == 'fft'
    nH, pH = be.reconstruction_gradient_H(V, W, H)
    assert be.last_path == 'split'          # H gradient: bf16 matrix cores, exact 3 x bf16 operand splits
    nW, pW = be.reconstruction_gradient_W(V, W, H)
    assert be.last_path == 'fft'
    Hf = H.clone()
    be.fused_update_H(V, W, Hf, slice(None), sparsity=0.05, eps=1e-9)
    assert be.last_path == 'split'
    Wf = W.clone()
    be.fused_update_W(V, Wf, Hf, slice(None), eps=1e-9)
    for name, got, ref in zip('R nH pH nW pW Hf Wf'.split(), (R, nH, pH, nW, pW, Hf, Wf), want):
        assert relmax(be.to_ndarray(got), ref) < 2e-5, name


@pytest.mark.parametrize('C,D,M,A', BASELINE_SHAPES, ids=['config2', 'config3', 'config4', 'config5'])
def test_fft_family_at_baseline_sizes(C, D, M, A):
    rng = np.random.default_rng(5)
    V = rng.random((2, C) + D).astype(np.float32)
    Wn = rng.random((M, C) + A).astype(np.float32)
    Hn = rng.random((2, M) + tuple(d + a - 1 for d, a in zip(D, A))).astype(np.float32)
    want = _oracle_primitives(V, Wn, Hn)
    be = make_backend(V, A, M, 'fft')
    W, H = dev(Wn, np.float32), dev(Hn, np.float32)
    R = be.reconstruct(W, H)
    assert be.last_path == 'fft'
    nH, pH = be.reconstruction_gradient_H(V, W, H)
    nW, pW = be.reconstruction_gradient_W(V, W, H)
    assert be.last_path == 'fft'
    Hf = H.clone()
    be.fused_update_H(V, W, Hf, slice(None), sparsity=0.05, eps=1e-9)
    for got, ref in zip((R, nH, pH, nW, pW), want[:5]):
        assert relmax(be.to_ndarray(got), ref) < 2e-5
    # The fused float32 update divides two gradients that each carry the transform's ABSOLUTE error (~1e-7 of the largest
    # entry): where both are tiny the quotient is not parity-grade, and path='fft' does not claim H parity in float32
    # (W-only opt-in).  What is asserted is that the fusion computes what the unfused kernels of the same family do.
    own = Hn.astype(np.float64) * be.to_ndarray(nH) / (be.to_ndarray(pH).astype(np.float64) + 1e-9 + 0.05)
    assert relmax(be.to_ndarray(Hf), own) < 1e-5


MODE_CASES = sorted(glob.glob(os.path.join(GOLDEN, 'modes_*.npz')))


@pytest.mark.parametrize('dtype,tol', [(np.float64, 1e-10), (np.float32, 2e-5)], ids=['f64', 'f32'])
@pytest.mark.parametrize('case', MODE_CASES, ids=[os.path.basename(p)[6:-4] for p in MODE_CASES])
def test_mode_primitives_against_reference_golden(case, dtype, tol):
    """'full' / 'circular' / 'reflect': pad + 'valid' kernels + fold against the reference PyTorch backend's outputs."""
    g = np.load(case)
    mode = os.path.basename(case).split('_')[1]
    V, s = g['V'].astype(dtype), _slice(g)
    A, M = g['W'].shape[2:], g['W'].shape[0]
    be = HIP_Backend(reconstruction_mode=mode)
    np.random.seed(1)
    _, H0 = be.initialize(V, tuple(A), M, None, tuple(range(-len(A), 0)))
    assert tuple(H0.shape) == tuple(g['init_H_shape'])
    W, H = dev(g['W'], dtype), dev(g['H'], dtype)
    assert relmax(be.to_ndarray(be.reconstruct(W, H)), g['R']) < tol
    neg, pos = be.reconstruction_gradient_H(V, W, H, s)
    assert neg.shape == H[s].shape
    assert relmax(be.to_ndarray(neg), g['neg_H']) < tol and relmax(be.to_ndarray(pos), g['pos_H']) < tol
    neg, pos = be.reconstruction_gradient_W(V, W, H, s)
    assert relmax(be.to_ndarray(neg), g['neg_W']) < tol and relmax(be.to_ndarray(pos), g['pos_W']) < tol
    assert abs(be.reconstruction_energy(V, W, H) - float(g['energy'])) / float(g['energy']) < tol


@pytest.mark.parametrize('mode,E', [('full', 1.87180), ('circular', 3.13228), ('reflect', 3.16430)])
def test_known_answer_1d_modes_f64(mode, E):
    np.random.seed(42)
    nmf = TransformInvariantNMF(n_atoms=3, atom_shape=(5,), backend='hip', reconstruction_mode=mode)
    nmf.fit(V_1D, inhibition_strength=0.1, n_iterations=10)
    assert np.isclose(nmf._energy_function(), E)                  # tnmf/tests/test_1d.py:17-22


@pytest.mark.parametrize('mode,E', [('full', 345.82498), ('circular', 265.35091)])
def test_known_answer_2d_rgb_modes_f64(mode, E):
    V = racoon_rgb_V()
    np.random.seed(42)
    nmf = TransformInvariantNMF(n_atoms=10, atom_shape=(7, 7), backend='hip', reconstruction_mode=mode)
    nmf.fit(V, sparsity_H=0.1, n_iterations=10)
    assert np.isclose(nmf._energy_function(), E)                  # tnmf/tests/test_backends.py:17-22
    np.random.seed(42)
    ref = orc.OracleNMF(n_atoms=10, atom_shape=(7, 7), impl='c', reconstruction_mode=mode)
    ref.fit(V, sparsity_H=0.1, n_iterations=10)
    assert np.allclose(nmf.W, ref.W) and np.allclose(nmf.H, ref.H) and np.allclose(nmf.R, ref.R)


def test_errors_like_the_reference():
    with pytest.raises(ValueError):
        HIP_Backend(reconstruction_mode='same')                   # unknown mode (_PyTorchBackend.py:50-52)
    be = HIP_Backend()
    with pytest.raises(TypeError):
        be.initialize(np.ones((1, 1, 8), dtype=np.int32), (3,), 2, None, (-1,))
    with pytest.raises(NotImplementedError):                         # four shift axes (the reference's PyTorch backend
        be.initialize(np.ones((1, 1, 4, 4, 4, 4)), (2, 2, 2, 2), 2, None, (-4, -3, -2, -1))   # asserts k <= 3: PyTorch.py:32)
