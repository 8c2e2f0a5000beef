"""
TEST-ONLY backend: plugs the CPU oracle (oracle/tnmf_oracle.py) under the product front-end
(tnmf_amd.TransformInvariantNMF) so that the front-end's schedules and the host-side sharding logic can be checked
on CPU -- against the reference's known answers and, with gloo, across two ranks.  Never used by the product.
"""
import numpy as np
import torch

from oracle import tnmf_oracle as orc
from tnmf_amd import sharding
from tnmf_amd.backends._Backend import Backend, sliceNone


class OracleBackend(Backend):
    def __init__(self, reconstruction_mode='valid', impl='c', process_group=None, hooks=False, schedules=False):
        if reconstruction_mode != 'valid':
            raise NotImplementedError
        super().__init__(reconstruction_mode)
        self.impl = impl
        self._group = process_group
        self._rank, self._world = 0, 1
        if process_group is not None:
            import torch.distributed as dist
            self._rank, self._world = dist.get_rank(process_group), dist.get_world_size(process_group)
        self._V_local = None
        self._shard = (0, 0)
        if hooks:   # expose the same optional hooks as the hip backend
            self.local_gradient_W = self._local_gradient_W
            self.all_reduce_gradient_W = self._all_reduce_gradient_W
            self.apply_W = self._apply_W
            self.fused_update_W = self._fused_update_W
            self.fused_update_H = self._fused_update_H
            self.minibatch_slices = self._minibatch_slices
        self.schedule_calls = []
        if hooks and schedules:   # ... and the operation-list hook (HIP_Backend.run_schedule), interpreted with the oracle
            self.supports_schedules = process_group is None
            self.run_schedule = self._run_schedule
            self.prefers_schedule = lambda H: True
            self.new_gradient_accumulator = lambda W: np.empty((2,) + W.shape, dtype=W.dtype)

    # -- set-up: same helpers as HIP_Backend._initialize_matrices --
    def _initialize_matrices(self, V, atom_shape, n_atoms, W=None, axes_W_normalization=None):
        N = self.n_samples
        n0, n1 = self._shard = sharding.shard_bounds(N, self._rank, self._world)
        self._V_local = np.ascontiguousarray(V[n0:n1])
        H = np.empty((n1 - n0, n_atoms) + self._transform_shape, dtype=V.dtype)
        for i, h in sharding.reference_init_stream(N, (n_atoms,) + self._transform_shape, self._shard, V.dtype):
            H[i] = h
        if W is None:
            W = sharding.reference_init_W(n_atoms, self.n_channels, self.atom_shape, V.dtype)
        return W, H

    def _minibatch_slices(self, batch_size):
        return sharding.local_minibatches(self.n_samples, self._rank, self._world, batch_size)

    def _reduce(self, arr):
        if self._group is not None:
            t = torch.from_numpy(arr)
            sharding.all_reduce_sum(t, self._group)
        return arr

    # -- primitives --
    def reconstruct(self, W, H):
        return orc.reconstruct(W, H, self.impl)

    def reconstruction_gradient_H(self, V, W, H, s=sliceNone):
        return orc.gradient_H(self._V_local, W, H, s, self.impl)

    def reconstruction_gradient_W(self, V, W, H, s=sliceNone):
        neg, pos = orc.gradient_W(self._V_local, W, H, s, self.impl)
        negpos = np.ascontiguousarray(np.stack([neg, pos]))
        self._reduce(negpos)
        return negpos[0], negpos[1]

    def reconstruction_energy(self, V, W, H):
        e = np.array([orc.energy(self._V_local, W, H, self.impl)])
        self._reduce(e)
        return float(e[0])

    def normalize(self, arr, axis=None):
        orc.normalize(arr, axis)

    def convolve_multi_1d(self, arr, kernels, axes):
        return orc.convolve_multi_1d(arr, kernels, axes)

    @staticmethod
    def to_ndarray(arr):
        return arr

    # -- optional hooks (enabled with hooks=True) --
    def _local_gradient_W(self, V, W, H, s=sliceNone):
        neg, pos = orc.gradient_W(self._V_local, W, H, s, self.impl)
        return np.ascontiguousarray(np.stack([neg, pos]))

    def _all_reduce_gradient_W(self, negpos):
        return self._reduce(negpos)

    def _apply_W(self, W, negpos, eps=1e-9):
        orc.multiplicative_update(W, negpos[0], negpos[1], eps,
                                  normalization_axes=tuple(range(-len(self.atom_shape), 0)))

    def _fused_update_W(self, V, W, H, s=sliceNone, eps=1e-9):
        self._apply_W(W, self._reduce(self._local_gradient_W(V, W, H, s)), eps)

    def _run_schedule(self, V, W, H, ops, acc, sparsity=0., eps=1e-9):
        """The contract of tnmf_hip_run_schedule (include/tnmf_hip.h), step by step with the oracle's primitives."""
        self.schedule_calls.append([op[0] for op in ops])
        for op in ops:
            if op[0] == 'H':
                self._fused_update_H(V, W, H, op[1], sparsity=sparsity, eps=eps)
            elif op[0] == 'G':
                g = self._local_gradient_W(V, W, H, op[1])
                a, b = op[2], op[3]
                acc[...] = b * g if a == 0 else a * acc + b * g
            elif op[0] == 'W':
                self._apply_W(W, acc, eps)            # (leaves acc[1] incremented by eps, like the reference's :232)
            else:
                raise ValueError(op)

    def _fused_update_H(self, V, W, H, s=sliceNone, sparsity=0., eps=1e-9, inhibition=0., cross_inhibition=0.,
                        inhibition_kernels=None):
        neg, pos = orc.gradient_H(self._V_local, W, H, s, self.impl)
        if inhibition > 0 or cross_inhibition > 0:      # the reference's lines, TransformInvariantNMF.py:253-269
            Hs = H[s]
            g = orc.convolve_multi_1d(Hs, inhibition_kernels, tuple(range(-len(self.atom_shape), 0)))
            if inhibition > 0:
                pos += inhibition * (g - Hs)
            if cross_inhibition > 0:
                pos += cross_inhibition / (H.shape[1] - 1) * (g.sum(axis=1, keepdims=True) - g)
        orc.multiplicative_update(H[s], neg, pos, eps, sparsity)
