"""
Host-side logic of the sample-sharded (data-parallel) mode -- no device code, testable on CPU with gloo.

The MU path shards over the sample axis: R, the H gradient and the H update of sample n touch only sample n, so
V[n0:n1] and H[n0:n1] stay resident on their rank and never move.  The W gradient is a sum over samples: every rank
computes its partial [neg | pos] and ONE all-reduce (sum) per W update makes it global; the replicated W update that
follows is identical on every rank.  (the reference itself is single-process.)
"""
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np

sliceNone = slice(None)


def shard_bounds(n_samples: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [n0, n1) of the sample axis owned by `rank` (blocks of ceil(N / world), tail may be short)."""
    per = -(-n_samples // world)
    return min(n_samples, rank * per), min(n_samples, (rank + 1) * per)


def local_minibatches(n_samples: int, rank: int, world: int, batch_size: Optional[int],
                      counts: Optional[Sequence[int]] = None) -> List[slice]:
    """
    Sequential mini-batches in the rank's LOCAL sample coordinates.  Global batch j is the union of every rank's
    local batch j; each rank contributes ceil(batch_size / world) of its own samples and all ranks get the same
    number of batches (possibly empty ones at the tail) so that their all-reduces pair up.
    world == 1 reproduces the reference's sequential split (tnmf/TransformInvariantNMF.py:29-37).
    `counts`: samples per rank when the ranks brought their own blocks (HIP_Backend(sharded_input=True)) instead of the
    even split of shard_bounds.
    """
    if batch_size is None:
        return [sliceNone]
    if counts is not None:
        assert len(counts) == world and sum(counts) == n_samples
        n_local, largest = int(counts[rank]), int(max(counts))
    else:
        n0, n1 = shard_bounds(n_samples, rank, world)
        n_local = n1 - n0
        largest = -(-n_samples // world)
    b = max(1, -(-int(batch_size) // world))
    return [slice(min(lo, n_local), min(lo + b, n_local)) for lo in range(0, largest, b)]


def reference_init_stream(n_samples: int, per_sample_shape: Sequence[int], shard: Tuple[int, int],
                          dtype) -> Iterator[Tuple[int, np.ndarray]]:
    """
    Yields (local index, 1 - rand(per_sample_shape) cast to dtype) for the samples of `shard`, consuming the global
    legacy NumPy RNG for ALL samples in order: consecutive ``np.random.rand`` calls continue one stream, so the values
    equal the rows of ``1 - np.random.rand(N, *per_sample_shape)`` (tnmf/backends/_Backend.py:92) on every rank.
    """
    n0, n1 = shard
    for n in range(n_samples):
        h = np.random.rand(*per_sample_shape)
        if n0 <= n < n1:
            yield n - n0, np.asarray(1 - h, dtype=dtype)


def reference_init_W(n_atoms: int, n_channels: int, atom_shape: Sequence[int], dtype) -> np.ndarray:
    """1 - rand cast to dtype, THEN normalised over the atom axes in that dtype (tnmf/backends/_Backend.py:95-96)."""
    W = np.asarray(1 - np.random.rand(n_atoms, n_channels, *atom_shape), dtype=dtype)
    W /= W.sum(axis=tuple(range(-len(atom_shape), 0)), keepdims=True)
    return W


def all_reduce_sum(tensor, group) -> None:
    """In-place sum over the ranks of `group` (RCCL on GPU tensors, gloo on CPU tensors); no-op without a group."""
    if group is None:
        return
    import torch.distributed as dist
    if dist.get_world_size(group) > 1:
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group)


def all_gather(tensor, group):
    """The ranks' tensors one behind the other, [world, *tensor.shape], in rank order (RCCL / gloo all-gather)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    flat = tensor.contiguous().reshape(-1)
    out = torch.empty(world * flat.numel(), dtype=tensor.dtype, device=tensor.device)
    dist.all_gather_into_tensor(out, flat, group=group)   # (flat on both sides: gloo insists on matching ranks)
    return out.reshape((world,) + tuple(tensor.shape))
