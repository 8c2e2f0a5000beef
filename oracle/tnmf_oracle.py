"""
CPU oracle for the shift-invariant multiplicative-update path of emdgroup/tnmf.

>>> TEST INFRASTRUCTURE ONLY. <<<
This module is the *checker*: a NumPy restatement of the reference's algorithm for the hot path
(SURVEY.md section 8a).  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  The product package ``tnmf_amd`` never imports it and has no CPU fallback.

Parity status: PINNED.  The functions below reproduce
  * the reference's own hard-coded known answers (tnmf/tests/test_1d.py:17-22, test_backends.py:17-22,
    test_minibatch.py:18-25, test_sparsity_inhibition.py:20-52) -- see tests/test_oracle_pinning.py, and
  * per-primitive outputs of the genuine reference ``tnmf.backends.PyTorch.PyTorch_Backend`` imported from
    /root/reference in the build container (tools/make_golden.py -> tests/golden/primitives_*.npz).
The reference's NumPy backend itself needs ``opt_einsum`` (requirements.txt:5, opt-einsum==3.3.0), which is
absent from this image and stays absent; for the two-operand contractions used on this path it reduces to
``numpy.tensordot`` -- the same products in BLAS summation order -- which is what ``contract`` below does.

Every function cites the reference lines it restates (paths relative to /root/reference/tnmf).
All shapes:  V[n, c, *D]   W[m, c, *A]   H[n, m, *D']   with 'valid' mode D' = D + A - 1.
"""
from __future__ import annotations

import ctypes
import enum
import os
import subprocess
from itertools import islice
from typing import Callable, Iterable, Optional, Sequence, Tuple

import numpy as np
from numpy.lib.stride_tricks import sliding_window_view

EPS = 1.0e-9  # TransformInvariantNMF.py:166


# ----------------------------------------------------------------------------------------------------------
# optional C flavour (oracle/tnmf_oracle_c.c): same index forms as plain loops, OpenMP, double accumulation.
# Pinned by the same golden vectors; used where the NumPy windows+contraction form is too slow for a test.
# ----------------------------------------------------------------------------------------------------------
_HERE = os.path.dirname(os.path.abspath(__file__))
_CLIB_PATH = os.path.join(_HERE, '_build', 'libtnmf_oracle.so')
_clib = None


_CLIB_AVX2_PATH = os.path.join(_HERE, '_build', 'libtnmf_oracle_avx2.so')


def build_c(force: bool = False) -> str:
    """Compile oracle/tnmf_oracle_c.c with gcc into oracle/_build/ (called by __graft_entry__.build()).

    Two flavours of the same source: a baseline x86-64 build and an AVX2+FMA build (same loops, wider vectors).  No
    -march=native: the libraries are built in the build container and run on the GPU box's host CPU; _c() picks the
    AVX2 flavour only when /proc/cpuinfo of the machine it runs on lists avx2 and fma."""
    src = os.path.join(_HERE, 'tnmf_oracle_c.c')
    for path, extra in ((_CLIB_PATH, []), (_CLIB_AVX2_PATH, ['-mavx2', '-mfma'])):
        if force or not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            os.makedirs(os.path.dirname(path), exist_ok=True)
            subprocess.check_call(['gcc', '-O3', '-fopenmp', '-shared', '-fPIC'] + extra + [src, '-o', path])
    return _CLIB_PATH


def _cpu_has_avx2_fma() -> bool:
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('flags'):
                    flags = set(line.split(':', 1)[1].split())
                    return 'avx2' in flags and 'fma' in flags
    except OSError:
        pass
    return False


class _Geom(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int) for k in ('N', 'M', 'C', 'Dy', 'Dx', 'Ay', 'Ax')]


def _c():
    global _clib
    if _clib is None:
        # NumPy's BLAS pool and libgomp's pool would oversubscribe the cores with spinning waiters
        os.environ.setdefault('OMP_WAIT_POLICY', 'passive')
        build_c()
        _clib = ctypes.CDLL(_CLIB_AVX2_PATH if _cpu_has_avx2_fma() else _CLIB_PATH)
        _clib.oracle_set_threads(int(os.environ.get('OMP_NUM_THREADS', default_threads())))
    return _clib


def default_threads(cap: int = 16) -> int:
    """Half of the cores this process may run on, at most `cap` (the NumPy BLAS pool wants the other half)."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 2
    return max(1, min(cap, avail // 2))


def set_threads(n: int) -> None:
    """OpenMP threads of the C flavour (the large-size parity tests and bench.py's parity leg raise the default)."""
    _c().oracle_set_threads(int(max(1, n)))


def _geom2d(N, M, C, D, A) -> _Geom:
    if len(A) == 1:
        return _Geom(N, M, C, 1, D[0], 1, A[0])
    if len(A) == 2:
        return _Geom(N, M, C, D[0], D[1], A[0], A[1])
    raise NotImplementedError('C oracle handles 1-D and 2-D shifts only')


def _c_call(name, dtype, g, *arrays):
    suf = {np.dtype('float64'): 'f64', np.dtype('float32'): 'f32'}[np.dtype(dtype)]
    fn = getattr(_c(), f'oracle_{name}_{suf}')
    fn.restype = None
    fn(ctypes.byref(g), *[a.ctypes.data_as(ctypes.c_void_p) for a in arrays])


def _c_reconstruct(W, H):
    W, H = np.ascontiguousarray(W), np.ascontiguousarray(H)
    A = W.shape[2:]
    D = tuple(h - a + 1 for h, a in zip(H.shape[2:], A))
    R = np.empty((H.shape[0], W.shape[1]) + D, dtype=H.dtype)
    _c_call('reconstruct', H.dtype, _geom2d(H.shape[0], W.shape[0], W.shape[1], D, A), W, H, R)
    return R


def _c_correlate_with_W(W, X):
    W, X = np.ascontiguousarray(W), np.ascontiguousarray(X)
    A, D = W.shape[2:], X.shape[2:]
    out = np.empty((X.shape[0], W.shape[0]) + tuple(d + a - 1 for d, a in zip(D, A)), dtype=X.dtype)
    _c_call('corr_with_W', X.dtype, _geom2d(X.shape[0], W.shape[0], W.shape[1], D, A), W, X, out)
    return out


def _c_correlate_H_with(X, H, A):
    X, H = np.ascontiguousarray(X), np.ascontiguousarray(H)
    out = np.empty((H.shape[1], X.shape[1]) + tuple(A), dtype=X.dtype)
    _c_call('corr_H_with', X.dtype, _geom2d(X.shape[0], H.shape[1], X.shape[1], X.shape[2:], A), H, X, out)
    return out


# ----------------------------------------------------------------------------------------------------------
# geometry and initialisation
# ----------------------------------------------------------------------------------------------------------
def transform_shape(sample_shape: Sequence[int], atom_shape: Sequence[int], mode: str = 'valid') -> Tuple[int, ...]:
    """Number of shifts per axis.  backends/_Backend.py:60-73."""
    d, a = np.asarray(sample_shape), np.asarray(atom_shape)
    if mode == 'valid':
        return tuple(int(x) for x in d + a - 1)
    if mode == 'full':
        return tuple(int(x) for x in d - a + 1)
    if mode in ('same', 'circular', 'reflect'):
        return tuple(int(x) for x in d)
    raise ValueError(mode)


def normalize(arr: np.ndarray, axes) -> None:
    """In-place division by the sum over ``axes``.  backends/_Backend.py:75-77."""
    arr /= arr.sum(axis=axes, keepdims=True)


def init_matrices(V: np.ndarray, atom_shape: Sequence[int], n_atoms: int,
                  W: Optional[np.ndarray] = None, mode: str = 'valid') -> Tuple[np.ndarray, np.ndarray]:
    """
    Random initialisation from the *global legacy* NumPy RNG: H is drawn first, then W (if not kept),
    both as ``1 - rand`` in float64 and cast to ``V.dtype``; W is normalised over the atom axes.
    backends/_Backend.py:83-98 (draw order :92-95), TransformInvariantNMF.py:165,273-280.
    """
    atom_shape = tuple(atom_shape)
    k = len(atom_shape)
    shifts = transform_shape(V.shape[2:], atom_shape, mode)
    H = np.asarray(1 - np.random.rand(V.shape[0], n_atoms, *shifts), dtype=V.dtype)
    if W is None:
        W = np.asarray(1 - np.random.rand(n_atoms, V.shape[1], *atom_shape), dtype=V.dtype)
        normalize(W, tuple(range(-k, 0)))
    return W, H


# ----------------------------------------------------------------------------------------------------------
# reconstruction modes other than 'valid': every mode is a 'valid' reconstruction of padded activations
# (backends/_PyTorchBackend.py:42-52 padding table, backends/PyTorch.py:36-41), so its gradients are the 'valid'
# gradients followed by the adjoint of the padding ("fold").
# ----------------------------------------------------------------------------------------------------------
MODES = ('valid', 'full', 'circular', 'reflect')


def pad_activations(H: np.ndarray, atom_shape: Sequence[int], mode: str) -> np.ndarray:
    """H[n,m,*D'] -> Hpad[n,m,*(D+A-1)]: zeros on both sides ('full'), wrap / reflect on the left ('circular'/'reflect')."""
    lead = ((0, 0), (0, 0))
    if mode == 'valid':
        return H
    if mode == 'full':
        return np.pad(H, lead + tuple((a - 1, a - 1) for a in atom_shape))
    if mode == 'circular':
        return np.pad(H, lead + tuple((a - 1, 0) for a in atom_shape), mode='wrap')
    if mode == 'reflect':
        return np.pad(H, lead + tuple((a - 1, 0) for a in atom_shape), mode='reflect')
    raise ValueError(mode)


def pad_source_index(n_shift: int, a: int, mode: str) -> np.ndarray:
    """For one shift axis: index into H of every padded position (-1 = a zero)."""
    if mode == 'valid':
        return np.arange(n_shift)
    if mode == 'full':
        return np.concatenate([np.full(a - 1, -1), np.arange(n_shift), np.full(a - 1, -1)])
    if mode == 'circular':
        return np.concatenate([np.arange(n_shift - (a - 1), n_shift), np.arange(n_shift)])
    if mode == 'reflect':
        return np.concatenate([np.arange(a - 1, 0, -1), np.arange(n_shift)])
    raise ValueError(mode)


def fold_gradient(Gpad: np.ndarray, shift_shape: Sequence[int], atom_shape: Sequence[int], mode: str) -> np.ndarray:
    """Adjoint of pad_activations: sums the gradient of every padded position into the activation it copies."""
    if mode == 'valid':
        return Gpad
    G = Gpad
    k = len(atom_shape)
    for ax in range(k):
        axis = G.ndim - k + ax
        src = pad_source_index(shift_shape[ax], atom_shape[ax], mode)
        out = np.zeros(G.shape[:axis] + (shift_shape[ax],) + G.shape[axis + 1:], dtype=G.dtype)
        Gm, om = np.moveaxis(G, axis, 0), np.moveaxis(out, axis, 0)
        for jpos, u in enumerate(src):
            if u >= 0:
                om[u] += Gm[jpos]
        G = out
    return G


# ----------------------------------------------------------------------------------------------------------
# the three primitives -- contraction form (same algorithm as backends/NumPy.py: windows + one contraction)
# ----------------------------------------------------------------------------------------------------------
def _shift_axes(k: int) -> Tuple[int, ...]:
    return tuple(range(-k, 0))


def contract(x, x_idx, y, y_idx, out_idx):
    """Two-operand contraction; stands where the reference calls opt_einsum.contract (NumPy.py:82,87,106,116,128)."""
    return np.einsum(x, x_idx, y, y_idx, out_idx, optimize=True)


def reconstruct(W: np.ndarray, H: np.ndarray, impl: str = 'contract', mode: str = 'valid') -> np.ndarray:
    """
    R[n,c,d] = sum_m sum_a H[n,m,d+a] * W[m,c,A-1-a]   ('valid' part of the full convolution H (*) W).
    backends/NumPy.py:122-132 (windows :124-127, flipped W :130).
    ``impl``: 'contract' (windows + one contraction, the reference's algorithm), 'shiftsum' or 'c'.
    ``mode``: reconstruction mode; non-'valid' modes pad H first (backends/PyTorch.py:36-43).
    """
    if mode != 'valid':
        return reconstruct(W, pad_activations(H, W.shape[2:], mode), impl)
    if impl == 'c':
        return _c_reconstruct(W, H)
    if impl == 'shiftsum':
        return reconstruct_shiftsum(W, H)
    k = W.ndim - 2
    A = W.shape[2:]
    Hw = sliding_window_view(H, A, axis=_shift_axes(k))          # [n, m, *D, *A]
    n_, m_, c_ = 0, 1, 2
    d_ = list(range(3, 3 + k))
    a_ = list(range(3 + k, 3 + 2 * k))
    return contract(Hw, [n_, m_] + d_ + a_, np.flip(W, _shift_axes(k)), [m_, c_] + a_, [n_, c_] + d_)


def _pad_atoms(X: np.ndarray, A: Sequence[int]) -> np.ndarray:
    """Zero-pad the shift axes by A-1 on both sides.  backends/NumPy.py:49,53,111."""
    return np.pad(X, ((0, 0), (0, 0)) + tuple((a - 1, a - 1) for a in A))


def _correlate_with_W(W: np.ndarray, X: np.ndarray, impl: str = 'contract') -> np.ndarray:
    """out[n,m,u] = sum_c sum_a W[m,c,a] * Xpad[n,c,u+a].  backends/NumPy.py:101-109 / :111-119."""
    if impl == 'c':
        return _c_correlate_with_W(W, X)
    if impl == 'shiftsum':
        return correlate_with_W_shiftsum(W, X)
    k = W.ndim - 2
    A = W.shape[2:]
    Xw = sliding_window_view(_pad_atoms(X, A), A, axis=_shift_axes(k))  # [n, c, *D', *A]
    n_, m_, c_ = 0, 1, 2
    d_ = list(range(3, 3 + k))
    a_ = list(range(3 + k, 3 + 2 * k))
    return contract(W, [m_, c_] + a_, Xw, [n_, c_] + d_ + a_, [n_, m_] + d_)


def gradient_H(V: np.ndarray, W: np.ndarray, H: np.ndarray, s: slice = slice(None), impl: str = 'contract',
               mode: str = 'valid'):
    """
    neg = correlation of V[s] with W, pos = correlation of R = reconstruct(W, H[s]) with W; both of H[s]'s shape.
    backends/NumPy.py:93-120.  Other modes: the same on the padded activations, folded back (what autograd does in
    backends/_PyTorchBackend.py:91-110).
    """
    A = W.shape[2:]
    neg = _correlate_with_W(W, V[s], impl)
    pos = _correlate_with_W(W, reconstruct(W, H[s], impl, mode), impl)
    return fold_gradient(neg, H.shape[2:], A, mode), fold_gradient(pos, H.shape[2:], A, mode)


def _correlate_H_with(X: np.ndarray, H: np.ndarray, A: Sequence[int], impl: str = 'contract') -> np.ndarray:
    """
    out[m,c,a] = sum_n sum_d H[n,m,d+A-1-a] * X[n,c,d]   (contract, then flip the shift axes).
    backends/NumPy.py:77-79,82-85.
    """
    if impl == 'c':
        return _c_correlate_H_with(X, H, A)
    if impl == 'shiftsum':
        return correlate_H_with_shiftsum(X, H, A)
    k = len(A)
    D = X.shape[2:]
    Hw = sliding_window_view(H, D, axis=_shift_axes(k))          # [n, m, *A, *D]
    n_, m_, c_ = 0, 1, 2
    a_ = list(range(3, 3 + k))
    d_ = list(range(3 + k, 3 + 2 * k))
    G = contract(Hw, [n_, m_] + a_ + d_, X, [n_, c_] + d_, [m_, c_] + a_)
    return np.flip(G, _shift_axes(k))


def gradient_W(V: np.ndarray, W: np.ndarray, H: np.ndarray, s: slice = slice(None), impl: str = 'contract',
               mode: str = 'valid'):
    """neg from V[s], pos from R = reconstruct(W, H[s]); both of W's shape.  backends/NumPy.py:69-91."""
    A = W.shape[2:]
    Hs = pad_activations(H[s], A, mode)
    neg = _correlate_H_with(V[s], Hs, A, impl)
    pos = _correlate_H_with(reconstruct(W, Hs, impl), Hs, A, impl)
    return neg, pos


def partial_reconstruct(W: np.ndarray, H: np.ndarray, i_atom: int, impl: str = 'contract',
                        mode: str = 'valid') -> np.ndarray:
    """backends/_Backend.py:124-125."""
    return reconstruct(W[i_atom:i_atom + 1], H[:, i_atom:i_atom + 1], impl, mode)


def energy(V: np.ndarray, W: np.ndarray, H: np.ndarray, impl: str = 'contract', mode: str = 'valid') -> float:
    """E = 1/2 sum (V - R)^2.  backends/_Backend.py:127-130."""
    R = reconstruct(W, H, impl, mode)
    assert R.shape == V.shape
    return float(0.5 * np.sum(np.square(V - R)))


# ----------------------------------------------------------------------------------------------------------
# independent second implementation (explicit sum over atom offsets) -- used to cross-check the windows form
# ----------------------------------------------------------------------------------------------------------
def _offsets(A: Sequence[int]):
    return np.ndindex(*A)


def reconstruct_shiftsum(W: np.ndarray, H: np.ndarray) -> np.ndarray:
    k = W.ndim - 2
    A = W.shape[2:]
    D = tuple(h - a + 1 for h, a in zip(H.shape[2:], A))
    R = np.zeros((H.shape[0], W.shape[1]) + D, dtype=np.result_type(W, H))
    for a in _offsets(A):
        win = (slice(None), slice(None)) + tuple(slice(ai, ai + di) for ai, di in zip(a, D))
        wf = W[(slice(None), slice(None)) + tuple(Ai - 1 - ai for Ai, ai in zip(A, a))]    # [m, c]
        R += np.tensordot(H[win], wf, axes=([1], [0])).transpose((0, k + 1) + tuple(range(1, k + 1)))
    return R


def correlate_with_W_shiftsum(W: np.ndarray, X: np.ndarray) -> np.ndarray:
    k = W.ndim - 2
    A = W.shape[2:]
    Xp = _pad_atoms(X, A)
    Dp = tuple(d + a - 1 for d, a in zip(X.shape[2:], A))
    out = np.zeros((X.shape[0], W.shape[0]) + Dp, dtype=np.result_type(W, X))
    for a in _offsets(A):
        win = (slice(None), slice(None)) + tuple(slice(ai, ai + di) for ai, di in zip(a, Dp))
        w = W[(slice(None), slice(None)) + tuple(a)]                                          # [m, c]
        out += np.tensordot(Xp[win], w, axes=([1], [1])).transpose((0, k + 1) + tuple(range(1, k + 1)))
    return out


def correlate_H_with_shiftsum(X: np.ndarray, H: np.ndarray, A: Sequence[int]) -> np.ndarray:
    D = X.shape[2:]
    out = np.zeros((H.shape[1], X.shape[1]) + tuple(A), dtype=np.result_type(X, H))
    for a in _offsets(A):
        win = (slice(None), slice(None)) + tuple(slice(Ai - 1 - ai, Ai - 1 - ai + di) for Ai, ai, di in zip(A, a, D))
        Hn = H[win].reshape(H.shape[0], H.shape[1], -1)
        Xn = X.reshape(X.shape[0], X.shape[1], -1)
        out[(slice(None), slice(None)) + tuple(a)] = np.einsum('nmp,ncp->mc', Hn, Xn)
    return out


# ----------------------------------------------------------------------------------------------------------
# FFT form of the three primitives ('valid' mode) -- restatement of the reference's default backend 'numpy_fft'
# (backends/NumPy_FFT.py:16-40 `_fft_convolve`, parameter tables backends/_NumPyFFTBackend.py:43-88).  Only used as the
# second, stronger CPU comparator of bench.py; pinned by the same golden vectors.
# ----------------------------------------------------------------------------------------------------------
def _fft_shape(sample_shape, transform_shape):
    from scipy.fft import next_fast_len
    return tuple(next_fast_len(int(d + t - 1)) for d, t in zip(sample_shape, transform_shape))


def _fft_convolve(arrs, arr2, subscripts, slices, axes, fft_shape, correlate):
    """rfftn both operands, contract per frequency, irfftn, slice (NumPy_FFT.py:29-40)."""
    from scipy.fft import irfftn, rfftn
    c2 = np.flip(arr2, axis=axes) if correlate else arr2
    f2 = rfftn(c2, axes=axes, s=fft_shape, workers=-1)
    out = []
    for arr in arrs:
        f1 = rfftn(arr, axes=axes, s=fft_shape, workers=-1)
        fr = np.einsum(subscripts, f1, f2, optimize=True)
        out.append(irfftn(fr, axes=axes, s=fft_shape, workers=-1)[slices].copy())
    return out


def reconstruct_fft(W, H):
    k = W.ndim - 2
    A = W.shape[2:]
    D = tuple(h - a + 1 for h, a in zip(H.shape[2:], A))
    sl = (slice(None), slice(None)) + tuple(slice(a - 1, a - 1 + d) for a, d in zip(A, D))
    return _fft_convolve((H,), W, 'nm...,mc...->nc...', sl, _shift_axes(k), _fft_shape(D, H.shape[2:]), False)[0]


def gradient_H_fft(V, W, H, s=slice(None)):
    k = W.ndim - 2
    Hs = H[s]
    D = V.shape[2:]
    sl = (slice(None), slice(None)) + tuple(slice(0, t) for t in Hs.shape[2:])
    R = reconstruct_fft(W, Hs)
    neg, pos = _fft_convolve((V[s], R), W, 'nc...,mc...->nm...', sl, _shift_axes(k), _fft_shape(D, Hs.shape[2:]), True)
    return neg, pos


def gradient_W_fft(V, W, H, s=slice(None)):
    k = W.ndim - 2
    A = W.shape[2:]
    Hs = H[s]
    D = V.shape[2:]
    lower = tuple(min(d, t) - 1 for d, t in zip(D, Hs.shape[2:]))
    sl = (slice(None), slice(None)) + tuple(slice(lo, lo + a) for lo, a in zip(lower, A))
    R = reconstruct_fft(W, Hs)
    neg, pos = _fft_convolve((V[s], R), Hs, 'nc...,nm...->mc...', sl, _shift_axes(k), _fft_shape(D, Hs.shape[2:]), True)
    return neg, pos


def mu_iteration_fft(V, W, H, eps: float = EPS, sparsity: float = 0.):
    """One full-batch MU iteration with the FFT primitives (TransformInvariantNMF.py:334-340 over 'numpy_fft')."""
    k = W.ndim - 2
    neg, pos = gradient_H_fft(V, W, H)
    multiplicative_update(H, neg.astype(H.dtype, copy=False), pos.astype(H.dtype, copy=False), eps, sparsity)
    neg, pos = gradient_W_fft(V, W, H)
    multiplicative_update(W, neg.astype(W.dtype, copy=False), pos.astype(W.dtype, copy=False), eps,
                          normalization_axes=tuple(range(-k, 0)))


# ----------------------------------------------------------------------------------------------------------
# lateral inhibition helper and the elementwise multiplicative update
# ----------------------------------------------------------------------------------------------------------
def convolve_multi_1d(arr: np.ndarray, kernels: Sequence[np.ndarray], axes: Iterable[int]) -> np.ndarray:
    """
    Separable zero-padded 'same' convolution, one 1-D kernel per axis (odd, symmetric kernels centred on the
    element).  backends/_NumPyBackend.py:56-64 (scipy.ndimage.convolve1d, mode='constant', cval=0).
    """
    out = arr
    for ax, kern in zip(axes, kernels):
        kern = np.asarray(kern)
        r = (len(kern) - 1) // 2
        moved = np.moveaxis(out, ax, -1)
        padded = np.pad(moved, [(0, 0)] * (moved.ndim - 1) + [(r, r)])
        win = sliding_window_view(padded, len(kern), axis=-1)
        moved = np.tensordot(win, kern[::-1].astype(arr.dtype, copy=False), axes=([-1], [0]))
        out = np.moveaxis(moved, -1, ax)
    return out


def inhibition_kernels(inhibition_range: Sequence[int]):
    """1 - (x/(i+1))^2 for x in -i..i.  TransformInvariantNMF.py:163."""
    return tuple(1 - (np.arange(-i, i + 1) / (i + 1)) ** 2 for i in inhibition_range)


def multiplicative_update(arr: np.ndarray, neg, pos, eps: float = EPS, sparsity: float = 0.,
                          normalization_axes=None) -> None:
    """
    ``pos += eps (+ sparsity)`` IN PLACE on pos; ``arr *= neg``; ``arr /= pos``; optional normalisation.
    There is no clip in the reference.  TransformInvariantNMF.py:217-238.
    """
    assert sparsity >= 0
    reg = eps
    if sparsity > 0:
        reg += sparsity
    pos += reg
    arr *= neg
    arr /= pos
    if normalization_axes is not None:
        normalize(arr, normalization_axes)


# ----------------------------------------------------------------------------------------------------------
# the front-end loop (restatement of TransformInvariantNMF.fit_batch / fit_minibatches / fit_stream)
# ----------------------------------------------------------------------------------------------------------
class MiniBatchAlgorithm(enum.Enum):
    """TransformInvariantNMF.py:47-55."""
    Cyclic_MU = 4
    ASG_MU = 5
    GSG_MU = 6
    ASAG_MU = 7
    GSAG_MU = 8


def sequential_minibatches(length: int, batch_size: Optional[int]):
    """TransformInvariantNMF.py:29-37."""
    if batch_size is None:
        return [slice(None)]
    return [slice(lo, min(length, lo + batch_size)) for lo in range(0, length, batch_size)]


def _shuffled(items):
    """np.random.permutation of the batch list, one RNG call per epoch.  TransformInvariantNMF.py:40-44."""
    idx = np.random.permutation(len(items))
    return [items[i] for i in idx]


class OracleNMF:
    """
    Same constructor/fit surface as the reference's TransformInvariantNMF, 'valid' mode only, NumPy only.
    TransformInvariantNMF.py:142-186 (ctor), :282-348 (fit_batch), :350-442 (fit_minibatches), :506-531.
    """

    def __init__(self, n_atoms: int, atom_shape: Sequence[int], inhibition_range=None, impl: str = 'contract',
                 reconstruction_mode: str = 'valid'):
        self.impl = impl
        self.mode = reconstruction_mode
        self.n_atoms = n_atoms
        self.atom_shape = tuple(atom_shape)
        k = len(self.atom_shape)
        if inhibition_range is None:
            rng = tuple(a - 1 for a in self.atom_shape)
        elif isinstance(inhibition_range, int):
            rng = (inhibition_range,) * k
        else:
            rng = tuple(inhibition_range)
        assert len(rng) == k
        self._kernels = inhibition_kernels(rng)
        self._norm_axes = tuple(range(-k, 0))
        self.eps = EPS
        self.W = self.H = self.V = None

    # -- properties mirroring the reference's read-outs (TransformInvariantNMF.py:188-215) --
    @property
    def R(self):
        return reconstruct(self.W, self.H, self.impl, self.mode)

    def R_partial(self, i_atom: int):
        return partial_reconstruct(self.W, self.H, i_atom, self.impl, self.mode)

    def energy(self) -> float:
        return energy(self.V, self.W, self.H, self.impl, self.mode)

    # -- half steps --
    def update_H(self, s=slice(None), sparsity=0., inhibition=0., cross_inhibition=0.):
        """TransformInvariantNMF.py:246-271."""
        neg, pos = gradient_H(self.V, self.W, self.H, s, self.impl, self.mode)
        if inhibition > 0 or cross_inhibition > 0:
            k = len(self.atom_shape)
            g = convolve_multi_1d(self.H[s], self._kernels, range(-k, 0))
            if inhibition > 0:
                pos += inhibition * (g - self.H[s])
            if cross_inhibition > 0:
                pos += (cross_inhibition / (self.n_atoms - 1)) * (g.sum(axis=1, keepdims=True) - g)
        multiplicative_update(self.H[s], neg, pos, self.eps, sparsity)

    def update_W(self, s=slice(None)):
        """TransformInvariantNMF.py:240-244."""
        neg, pos = gradient_W(self.V, self.W, self.H, s, self.impl, self.mode)
        multiplicative_update(self.W, neg, pos, self.eps, normalization_axes=self._norm_axes)

    def _accumulate(self, acc_neg, acc_pos, lam, s):
        """TransformInvariantNMF.py:444-455 (the += / *= forms, including their in-place side effects)."""
        neg, pos = gradient_W(self.V, self.W, self.H, s, self.impl, self.mode)
        if lam == 1:
            acc_neg = acc_neg + neg if np.isscalar(acc_neg) else acc_neg.__iadd__(neg)
            acc_pos = acc_pos + pos if np.isscalar(acc_pos) else acc_pos.__iadd__(pos)
        else:
            if np.isscalar(acc_neg):
                acc_neg, acc_pos = acc_neg * (1 - lam) + lam * neg, acc_pos * (1 - lam) + lam * pos
            else:
                acc_neg *= (1 - lam)
                acc_pos *= (1 - lam)
                acc_neg += lam * neg
                acc_pos += lam * pos
        return acc_neg, acc_pos

    def _init(self, V, keep_W):
        self.V = V
        self.W, self.H = init_matrices(V, self.atom_shape, self.n_atoms, self.W if keep_W else None, self.mode)

    # -- full batch --
    def fit_batch(self, V, n_iterations=1000, update_H=True, update_W=True, keep_W=False, sparsity_H=0.,
                  inhibition_strength=0., cross_atom_inhibition_strength=0.,
                  progress_callback: Optional[Callable] = None):
        assert np.all(V >= 0)
        self._init(V, keep_W)
        for it in range(n_iterations):
            if update_H:
                self.update_H(sparsity=sparsity_H, inhibition=inhibition_strength,
                              cross_inhibition=cross_atom_inhibition_strength)
            if update_W:
                self.update_W()
            if progress_callback is not None and not progress_callback(self, it):
                break
        return self

    # -- mini batches --
    def fit_minibatches(self, V, algorithm=MiniBatchAlgorithm.ASG_MU, batch_size=3, n_epochs=1000, sag_lambda=0.2,
                        keep_W=False, sparsity_H=0., inhibition_strength=0., cross_atom_inhibition_strength=0.,
                        progress_callback: Optional[Callable] = None):
        assert np.all(V >= 0)
        algorithm = MiniBatchAlgorithm(getattr(algorithm, 'value', algorithm))
        # `algorithm in (5, 6, 7, 8)` compares an Enum member with ints and is always False in the reference
        # (TransformInvariantNMF.py:410), so V is never shuffled.
        self._init(V, keep_W)
        batches = sequential_minibatches(len(V), batch_size)
        kw = dict(sparsity=sparsity_H, inhibition=inhibition_strength, cross_inhibition=cross_atom_inhibition_strength)
        stat = None
        for epoch in range(n_epochs):
            stat = self._epoch(algorithm, stat, batches, kw, sag_lambda)
            if progress_callback is not None and not progress_callback(self, epoch):
                break
        return self

    def _mu_W(self, neg, pos):
        multiplicative_update(self.W, neg, pos, self.eps, normalization_axes=self._norm_axes)

    def _epoch(self, algorithm, stat, batches, kw, lam):
        A = MiniBatchAlgorithm
        if algorithm is A.Cyclic_MU:          # TransformInvariantNMF.py:457-465
            g = (0, 0)
            for b in batches:
                self.update_H(b, **kw)
                g = self._accumulate(*g, 1., b)
            self._mu_W(*g)
            return None
        if algorithm is A.ASG_MU:             # :467-472
            for b in _shuffled(batches):
                self.update_H(b, **kw)
                self.update_W(b)
            return None
        if algorithm is A.GSG_MU:             # :474-479
            b = slice(0, 0)
            for b in _shuffled(batches):
                self.update_H(b, **kw)
            self.update_W(b)
            return None
        if algorithm is A.ASAG_MU:            # :481-491  (pos accumulator gains eps on every W update, :232,490)
            if stat is None:
                stat = (0, 0)
            for b in _shuffled(batches):
                self.update_H(b, **kw)
                stat = self._accumulate(*stat, lam, b)
                self._mu_W(*stat)
            return stat
        if algorithm is A.GSAG_MU:            # :493-504
            if stat is None:
                stat = (0, 0)
            b = slice(0, 0)
            for b in _shuffled(batches):
                self.update_H(b, **kw)
            stat = self._accumulate(*stat, lam, b)
            self._mu_W(*stat)
            return stat
        raise ValueError(algorithm)

    # -- streaming --
    def fit_stream(self, V, subsample_size=3, max_subsamples=None, **kwargs):
        """TransformInvariantNMF.py:506-523: only W carries over between subsamples."""
        it = iter(V)
        isub = 0
        while True:
            chunk = list(islice(it, subsample_size))
            if not chunk:
                return self
            self.fit(np.asarray(chunk), keep_W=True, **kwargs)
            if max_subsamples is not None and isub == max_subsamples - 1:
                return self
            isub += 1

    def fit(self, V, **kwargs):
        """TransformInvariantNMF.py:525-531."""
        if 'subsample_size' in kwargs or 'max_subsamples' in kwargs:
            return self.fit_stream(iter(V), **kwargs)
        if 'batch_size' in kwargs or 'algorithm' in kwargs:
            return self.fit_minibatches(V, **kwargs)
        return self.fit_batch(V, **kwargs)


# ----------------------------------------------------------------------------------------------------------
# sample-chunked MU iteration: identical math, bounded im2col temporary (bench.py cpu_baseline at large sizes)
# ----------------------------------------------------------------------------------------------------------
def mu_iteration_chunked(V, W, H, chunk: int, eps: float = EPS, sparsity: float = 0.):
    """
    One full-batch MU iteration (H update, then W update, TransformInvariantNMF.py:334-340) evaluated sample
    chunk by sample chunk: R and the H gradient are per-sample, the W gradient sums over chunks.
    """
    k = W.ndim - 2
    N = V.shape[0]
    for lo in range(0, N, chunk):
        s = slice(lo, min(N, lo + chunk))
        neg, pos = gradient_H(V, W, H, s)
        multiplicative_update(H[s], neg, pos, eps, sparsity)
    acc_n = np.zeros_like(W)
    acc_p = np.zeros_like(W)
    for lo in range(0, N, chunk):
        s = slice(lo, min(N, lo + chunk))
        neg, pos = gradient_W(V, W, H, s)
        acc_n += neg
        acc_p += pos
    multiplicative_update(W, acc_n, acc_p, eps, normalization_axes=tuple(range(-k, 0)))
