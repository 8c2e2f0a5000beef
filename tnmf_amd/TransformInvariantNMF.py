"""
Front end of the MI355X shift-invariant NMF: the multiplicative-update schedules that drive a backend.

Public surface = the reference's (tnmf/TransformInvariantNMF.py): ``TransformInvariantNMF(n_atoms, atom_shape,
inhibition_range, backend, logger, verbose, **backend_kwargs)`` with ``fit`` / ``fit_batch`` / ``fit_minibatches`` /
``fit_stream``, the read-outs ``W``, ``H``, ``V``, ``R``, ``R_partial(i)`` and the ``MiniBatchAlgorithm`` enum.
The only backend shipped is ``'hip'`` (tnmf_amd/backends/HIP.py); any object implementing
tnmf_amd.backends._Backend.Backend can be passed instead of a name.

Differences from the reference, all deliberate:
  * when the backend offers the fused hooks (``fused_update_H`` / ``fused_update_W`` / ``multiplicative_update``)
    they replace the three elementwise passes of TransformInvariantNMF.py:232-235 -- same arithmetic, less traffic;
    ``use_fused_updates=False`` restores the call-by-call form of the reference;
  * the per-iteration energy is evaluated only if the logger is enabled for INFO (the reference formats it eagerly,
    TransformInvariantNMF.py:346, which costs a reconstruction per iteration).
"""
import enum
import itertools
import logging
from typing import Callable, Iterable, Iterator, List, Optional, Tuple, Union

import numpy as np

from .backends._Backend import Backend, sliceNone


class MiniBatchAlgorithm(enum.Enum):
    """Mini-batch MU schedules of Serizel et al. 2016 (reference: TransformInvariantNMF.py:47-55)."""
    Cyclic_MU = 4
    ASG_MU = 5
    GSG_MU = 6
    ASAG_MU = 7
    GSAG_MU = 8


def _sequential_minibatches(length: int, batch_size: Optional[int]) -> List[slice]:
    if batch_size is None:
        return [sliceNone]
    return [slice(lo, min(lo + batch_size, length)) for lo in range(0, length, batch_size)]


def _permuted(items: list) -> list:
    # one np.random.permutation call per epoch, like the reference's _random_shuffle (:40-44)
    order = np.random.permutation(len(items))
    return [items[i] for i in order]


def _joined(batches: list) -> list:
    """A run of H half steps on pairwise disjoint sample slices commutes (the H update of a sample reads that sample and W
    only, reference :246-271, and W does not change inside the run): the slices sorted and joined where they touch.
    Overlapping or strided slices: the list as it is."""
    spans = []
    for b in batches:
        if b.step not in (None, 1) or b.start is None or b.stop is None or b.start < 0 or b.stop < 0:
            return list(batches)
        if b.stop > b.start:
            spans.append((b.start, b.stop))
    spans.sort()
    if any(spans[i][0] < spans[i - 1][1] for i in range(1, len(spans))):
        return list(batches)
    out = []
    for lo, hi in spans:
        if out and out[-1][1] == lo:
            out[-1][1] = hi
        else:
            out.append([lo, hi])
    return [slice(lo, hi) for lo, hi in out]


def _backend_registry():
    from .backends.HIP import HIP_Backend
    return {'hip': HIP_Backend}


ProgressCallback = Callable[['TransformInvariantNMF', int], bool]


class TransformInvariantNMF:
    r"""
    Shift-invariant non-negative matrix factorisation ``V ~ sum_m H[:, m] (*) W[m]`` by multiplicative updates.

    Parameters
    ----------
    n_atoms, atom_shape : dictionary size; ``W`` has shape ``(n_atoms, n_channels, *atom_shape)``
    inhibition_range : lateral inhibition range per shift axis (default ``atom_shape - 1``)
    backend : ``'hip'`` or a :class:`~tnmf_amd.backends._Backend.Backend` instance
    logger, verbose : as in the reference (0 errors .. 3 debug)
    use_fused_updates : use the backend's fused half-step kernels when no inhibition term is requested
    **kwargs : forwarded to the backend constructor (``reconstruction_mode``, ``device``, ``path``, ``init``,
               ``process_group``)
    """

    def __init__(self, n_atoms: int, atom_shape: Tuple[int, ...], inhibition_range: Union[int, Tuple[int, ...]] = None,
                 backend: Union[str, Backend] = 'hip', logger: logging.Logger = None, verbose: int = 0,
                 use_fused_updates: bool = True, **kwargs):
        self.atom_shape = tuple(atom_shape)
        self.n_atoms = n_atoms
        k = len(self.atom_shape)
        if inhibition_range is None:
            self._inhibition_range = tuple(a - 1 for a in self.atom_shape)
        elif isinstance(inhibition_range, int):
            self._inhibition_range = (inhibition_range,) * k
        else:
            self._inhibition_range = tuple(inhibition_range)
        assert len(self._inhibition_range) == k
        # parabolic 1-D kernels 1 - (x / (i + 1))^2, x = -i..i  (reference :163)
        self._inhibition_kernels_1D = tuple(1 - (np.arange(-i, i + 1) / (i + 1)) ** 2 for i in self._inhibition_range)
        self._axes_W_normalization = tuple(range(-k, 0))
        self.eps = 1.e-9

        if isinstance(backend, str):
            registry = _backend_registry()
            if backend.lower() not in registry:
                raise KeyError(f'unknown backend {backend!r}; this package provides {sorted(registry)}')
            self._backend = registry[backend.lower()](**kwargs)
        else:
            self._backend = backend

        self._logger = logger if logger is not None else logging.getLogger(self.__class__.__name__)
        self._logger.setLevel([logging.ERROR, logging.WARNING, logging.INFO, logging.DEBUG][verbose])
        self._use_fused = bool(use_fused_updates)
        self._use_schedules = bool(use_fused_updates)   # mini-batch epochs as one backend call where the backend can
        self._iteration_acc = None

        self._W = None
        self._H = None
        self._V = None
        self._shuffle_idx = None

    # -- read-outs (reference :188-215) ---------------------------------------------------------------------
    @property
    def W(self) -> np.ndarray:
        return self._backend.to_ndarray(self._W)

    @property
    def H(self) -> np.ndarray:
        H = self._backend.to_ndarray(self._H)
        return H if self._shuffle_idx is None else H[np.argsort(self._shuffle_idx)]

    @property
    def V(self) -> np.ndarray:
        return self._V if self._shuffle_idx is None else self._V[np.argsort(self._shuffle_idx)]

    @property
    def R(self) -> np.ndarray:
        return self._backend.to_ndarray(self._backend.reconstruct(self._W, self._H))

    def R_partial(self, i_atom: int) -> np.ndarray:
        return self._backend.to_ndarray(self._backend.partial_reconstruct(self._W, self._H, i_atom))

    def _energy_function(self) -> float:
        return self._backend.reconstruction_energy(self._V, self._W, self._H)

    # -- elementwise multiplicative update (reference :217-238) ----------------------------------------------
    def _multiplicative_update(self, arr, neg, pos, sparsity: float = 0., normalization_axes=None):
        assert sparsity >= 0
        regularization = self.eps + (sparsity if sparsity > 0 else 0.)
        hook = getattr(self._backend, 'multiplicative_update', None)
        if hook is not None:
            hook(arr, neg, pos, regularization)
        else:
            pos += regularization
            arr *= neg
            arr /= pos
        if normalization_axes is not None:
            self._backend.normalize(arr, axis=normalization_axes)

    # -- half steps (reference :240-271) ----------------------------------------------------------------------
    def _fused(self, name: str):
        return getattr(self._backend, name, None) if self._use_fused else None

    def _update_W(self, s: slice = sliceNone):
        fused = self._fused('fused_update_W')
        if fused is not None:
            fused(self._V, self._W, self._H, s, eps=self.eps)
            return
        neg, pos = self._backend.reconstruction_gradient_W(self._V, self._W, self._H, s)
        assert neg.shape == self._W.shape and pos.shape == self._W.shape
        self._multiplicative_update(self._W, neg, pos, normalization_axes=self._axes_W_normalization)

    def _update_H(self, s: slice = sliceNone, sparsity: float = 0., inhibition: float = 0., cross_inhibition: float = 0.):
        lateral = inhibition > 0 or cross_inhibition > 0
        fused = self._fused('fused_update_H')
        if fused is not None:
            try:
                if lateral:
                    fused(self._V, self._W, self._H, s, sparsity=sparsity, eps=self.eps, inhibition=inhibition,
                          cross_inhibition=cross_inhibition, inhibition_kernels=self._inhibition_kernels_1D)
                else:
                    fused(self._V, self._W, self._H, s, sparsity=sparsity, eps=self.eps)
                return
            except NotImplementedError:
                # (inhibition kernels longer than the backend's fused kernel takes; lateral terms or reconstruction modes
                # of volumes: nothing has been written, the reference's own lines below do the step)
                pass
        neg, pos = self._backend.reconstruction_gradient_H(self._V, self._W, self._H, s)
        Hs = self._H[s]
        assert neg.shape == Hs.shape and pos.shape == Hs.shape
        if lateral:
            axes = tuple(range(-len(self.atom_shape), 0))
            g = self._backend.convolve_multi_1d(Hs, self._inhibition_kernels_1D, axes)
            if inhibition > 0:
                term = g - Hs               # an activation does not inhibit itself at its own position
                term *= inhibition
                pos += term
            if cross_inhibition > 0:
                term = g.sum(axis=1, keepdims=True) - g   # what all OTHER atoms contribute at this shift
                term *= cross_inhibition / (self.n_atoms - 1)
                pos += term
        self._multiplicative_update(Hs, neg, pos, sparsity=sparsity)

    def _iteration(self, h_args, update_H: bool = True, update_W: bool = True):
        """One full-batch MU iteration (reference :334-340).  A problem small enough to be launch-latency bound goes to the
        backend as one operation list (one persistent kernel launch per iteration, HIP_Backend.run_schedule)."""
        run = self._scheduler(h_args)
        if run is not None and getattr(self._backend, 'prefers_schedule', lambda *_: False)(self._H):
            ops = ([('H', sliceNone)] if update_H else []) + ([('G', sliceNone, 0., 1.), ('W',)] if update_W else [])
            acc = self._iteration_acc
            if acc is None or acc.shape[1:] != self._W.shape or acc.dtype != self._W.dtype or acc.device != self._W.device:
                self._iteration_acc = self._backend.new_gradient_accumulator(self._W)
            run(self._V, self._W, self._H, ops, self._iteration_acc, sparsity=h_args['sparsity'], eps=self.eps)
            return
        if update_H:
            self._update_H(**h_args)
        if update_W:
            self._update_W()

    def _initialize_matrices(self, V: np.ndarray, keep_W: bool):
        self._V = V
        self._iteration_acc = None    # (sized and typed for the W of ONE fit: a refit may change dtype or device)
        self._W, self._H = self._backend.initialize(self._V, self.atom_shape, self.n_atoms,
                                                    self._W if keep_W else None, self._axes_W_normalization)

    def _report(self, what: str, step: int, progress_callback: Optional[ProgressCallback]) -> bool:
        """Returns False when the callback asks to stop."""
        if progress_callback is not None:
            return bool(progress_callback(self, step))
        if self._logger.isEnabledFor(logging.INFO):
            self._logger.info(f'{what}: {step}\tEnergy function: {self._energy_function()}')
        return True

    # -- full batch (reference :282-348) ------------------------------------------------------------------------
    def fit_batch(self, V: np.ndarray, n_iterations: int = 1000, update_H: bool = True, update_W: bool = True,
                  keep_W: bool = False, sparsity_H: float = 0., inhibition_strength: float = 0.,
                  cross_atom_inhibition_strength: float = 0., progress_callback: ProgressCallback = None):
        assert np.all(V >= 0)
        assert update_H or update_W
        assert sparsity_H >= 0 and inhibition_strength >= 0 and cross_atom_inhibition_strength >= 0
        self._initialize_matrices(V, keep_W)
        h_args = dict(sparsity=sparsity_H, inhibition=inhibition_strength,
                      cross_inhibition=cross_atom_inhibition_strength)
        for iteration in range(n_iterations):
            self._iteration(h_args, update_H, update_W)
            if not self._report('Iteration', iteration, progress_callback):
                break
        self._logger.info('TNMF finished.')

    # -- mini batches (reference :350-504) ----------------------------------------------------------------------
    def fit_minibatches(self, V: np.ndarray, algorithm: MiniBatchAlgorithm = MiniBatchAlgorithm.ASG_MU,
                        batch_size: int = 3, n_epochs: int = 1000, sag_lambda: float = 0.2, keep_W: bool = False,
                        sparsity_H: float = 0., inhibition_strength: float = 0.,
                        cross_atom_inhibition_strength: float = 0., progress_callback: ProgressCallback = None):
        assert np.all(V >= 0)
        assert sparsity_H >= 0 and inhibition_strength >= 0 and cross_atom_inhibition_strength >= 0
        assert isinstance(algorithm, MiniBatchAlgorithm)
        # The reference decides whether to shuffle V with `algorithm in (5, 6, 7, 8)` (:410), an Enum-vs-int test
        # that is never true, so V is never shuffled; this front end keeps that behaviour.
        self._initialize_matrices(V, keep_W)
        plan = getattr(self._backend, 'minibatch_slices', None)
        batches = plan(batch_size) if plan is not None else _sequential_minibatches(len(self._V), batch_size)
        h_args = dict(sparsity=sparsity_H, inhibition=inhibition_strength,
                      cross_inhibition=cross_atom_inhibition_strength)
        epoch_fn = {
            MiniBatchAlgorithm.Cyclic_MU: self._epoch_cyclic,
            MiniBatchAlgorithm.ASG_MU: self._epoch_asg,
            MiniBatchAlgorithm.GSG_MU: self._epoch_gsg,
            MiniBatchAlgorithm.ASAG_MU: self._epoch_asag,
            MiniBatchAlgorithm.GSAG_MU: self._epoch_gsag,
        }[algorithm]
        state = None
        for epoch in range(n_epochs):
            state = epoch_fn(state, batches, h_args, sag_lambda)
            if not self._report('Epoch', epoch, progress_callback):
                break
        self._logger.info('MiniBatch TNMF finished.')

    def _blend_gradient_W(self, acc, lam: float, s: slice):
        """acc <- (1 - lam) * acc + lam * grad_W(batch s); lam == 1 is a plain sum (reference :444-455)."""
        neg, pos = self._backend.reconstruction_gradient_W(self._V, self._W, self._H, s)
        if acc is None:
            # the reference starts from the integers (0, 0): `0 + g` / `0 * (1 - lam) + lam * g`
            if lam == 1:
                return [neg, pos]
            return [lam * neg, lam * pos]
        if lam == 1:
            acc[0] += neg
            acc[1] += pos
        else:
            acc[0] *= (1 - lam)
            acc[1] *= (1 - lam)
            acc[0] += lam * neg
            acc[1] += lam * pos
        return acc

    def _apply_accumulated_W(self, acc):
        # NB: like the reference (:232), the update adds eps to the `pos` accumulator in place
        self._multiplicative_update(self._W, acc[0], acc[1], normalization_axes=self._axes_W_normalization)

    # One epoch of a mini-batch schedule as ONE call of the backend (HIP_Backend.run_schedule -> tnmf_hip_run_schedule):
    # the epoch functions below describe the epoch as a list of operations -- ('H', batch), ('G', batch, a, b) for
    # acc = a * acc + b * gradient_W(batch), ('W',) for the W update from acc -- where the backend offers that and no
    # lateral term is on (those go through tnmf_hip_update_H_ex batch by batch).
    def _scheduler(self, h_args):
        run = self._fused('run_schedule') if self._use_schedules else None
        if run is None or not getattr(self._backend, 'supports_schedules', False):
            return None
        if h_args['inhibition'] > 0 or h_args['cross_inhibition'] > 0:
            return None
        return run

    @staticmethod
    def _blend_coefficients(first: bool, lam: float):
        """(a, b) of acc = a * acc + b * g for reference :444-455; `first`: acc is still the integer 0 of :482/:495."""
        if first:
            return 0., (1. if lam == 1 else lam)
        return (1., 1.) if lam == 1 else (1. - lam, lam)

    def _epoch_cyclic(self, _state, batches, h_args, _lam):
        """Algorithm 4: H per batch, W once per epoch from the summed gradient (reference :457-465)."""
        run = self._scheduler(h_args)
        if run is not None and len(batches):
            ops = []
            for i, batch in enumerate(batches):
                ops += [('H', batch), ('G', batch, 0. if i == 0 else 1., 1.)]
            run(self._V, self._W, self._H, ops + [('W',)], self._backend.new_gradient_accumulator(self._W),
                sparsity=h_args['sparsity'], eps=self.eps)
            return None
        local = self._fused('local_gradient_W')
        lateral = h_args['inhibition'] > 0 or h_args['cross_inhibition'] > 0
        if local is not None and not lateral:
            # sum this rank's [neg | pos] over its batches, ONE all-reduce per epoch, then the fused MU
            total = None
            blend = self._fused('blend_gradient_W')
            for batch in batches:
                self._update_H(batch, **h_args)
                part = local(self._V, self._W, self._H, batch)
                if total is None:
                    total = part
                elif blend is not None:
                    blend(total, part, 1., 1.)
                else:
                    total += part
            total = self._backend.all_reduce_gradient_W(total)
            self._backend.apply_W(self._W, total, eps=self.eps)
            return None
        acc = None
        for batch in batches:
            self._update_H(batch, **h_args)
            acc = self._blend_gradient_W(acc, 1., batch)
        self._apply_accumulated_W(acc)
        return None

    def _epoch_asg(self, _state, batches, h_args, _lam):
        """Algorithm 5: H and W after every (shuffled) batch (reference :467-472)."""
        run = self._scheduler(h_args)
        if run is not None:
            ops = []
            for batch in _permuted(batches):
                ops += [('H', batch), ('G', batch, 0., 1.), ('W',)]
            run(self._V, self._W, self._H, ops, self._backend.new_gradient_accumulator(self._W),
                sparsity=h_args['sparsity'], eps=self.eps)
            return None
        for batch in _permuted(batches):
            self._update_H(batch, **h_args)
            self._update_W(batch)
        return None

    def _epoch_gsg(self, _state, batches, h_args, _lam):
        """Algorithm 6: H for every shuffled batch, W from the last batch only (reference :474-479)."""
        batch = slice(0, 0)
        run = self._scheduler(h_args)
        if run is not None:
            order = _permuted(batches)
            ops = [('H', b) for b in order] + [('G', order[-1] if len(order) else batch, 0., 1.), ('W',)]
            run(self._V, self._W, self._H, ops, self._backend.new_gradient_accumulator(self._W),
                sparsity=h_args['sparsity'], eps=self.eps)
            return None
        order = _permuted(batches)
        batch = order[-1] if len(order) else batch
        for run_ in _joined(order):     # (the H steps of an epoch commute: one call per contiguous run of samples)
            self._update_H(run_, **h_args)
        self._update_W(batch)
        return None

    def _epoch_asag(self, state, batches, h_args, lam):
        """Algorithm 7: running average of the W gradient over batches AND epochs (reference :481-491)."""
        run = self._scheduler(h_args)
        if run is not None:
            first = state is None
            acc = self._backend.new_gradient_accumulator(self._W) if first else state
            ops = []
            for batch in _permuted(batches):
                ops += [('H', batch), ('G', batch) + self._blend_coefficients(first, lam), ('W',)]
                first = False
            run(self._V, self._W, self._H, ops, acc, sparsity=h_args['sparsity'], eps=self.eps)
            return acc if len(ops) else state
        for batch in _permuted(batches):
            self._update_H(batch, **h_args)
            state = self._blend_gradient_W(state, lam, batch)
            self._apply_accumulated_W(state)
        return state

    def _epoch_gsag(self, state, batches, h_args, lam):
        """Algorithm 8: H for every batch, averaged gradient refreshed from the last one (reference :493-504)."""
        batch = slice(0, 0)
        run = self._scheduler(h_args)
        if run is not None:
            first = state is None
            acc = self._backend.new_gradient_accumulator(self._W) if first else state
            order = _permuted(batches)
            ops = [('H', b) for b in order]
            ops += [('G', order[-1] if len(order) else batch) + self._blend_coefficients(first, lam), ('W',)]
            run(self._V, self._W, self._H, ops, acc, sparsity=h_args['sparsity'], eps=self.eps)
            return acc
        order = _permuted(batches)
        batch = order[-1] if len(order) else batch
        for run_ in _joined(order):
            self._update_H(run_, **h_args)
        state = self._blend_gradient_W(state, lam, batch)
        self._apply_accumulated_W(state)
        return state

    # -- streaming (reference :506-523) ---------------------------------------------------------------------------
    def fit_stream(self, V: Iterator[np.ndarray], subsample_size: int = 3, max_subsamples: int = None, **kwargs):
        for isub in itertools.count(0):
            subsample = list(itertools.islice(V, subsample_size))
            if not subsample:
                self._logger.info('Sample iterator exhausted. TNMF on full iterator finished.')
                return
            self._logger.info(f'Processing subsample {isub}.')
            self.fit(np.asarray(subsample), keep_W=True, **kwargs)   # only W carries over
            if max_subsamples is not None and isub == max_subsamples - 1:
                self._logger.info(f'Processed {max_subsamples} subsamples. TNMF on iterator will stop.')
                return

    def fit(self, V: Union[np.ndarray, Iterable[np.ndarray]], **kwargs):
        """Dispatch on the keyword arguments exactly like the reference (:525-531)."""
        if 'subsample_size' in kwargs or 'max_subsamples' in kwargs:
            self.fit_stream(iter(V), **kwargs)
        elif 'batch_size' in kwargs or 'algorithm' in kwargs:
            self.fit_minibatches(V, **kwargs)
        else:
            self.fit_batch(V, **kwargs)
