"""
Sample-sharded (N > 1 ranks) path on CPU: two gloo ranks run the product front-end + the host-side sharding logic
(tnmf_amd/sharding.py) over the TEST-ONLY oracle backend and must reproduce the single-process factorisation.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tnmf_amd import sharding


def test_shard_bounds_cover_the_sample_axis():
    for N in (0, 1, 5, 8, 17):
        for world in (1, 2, 3, 8):
            blocks = [sharding.shard_bounds(N, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == N
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))


def test_local_minibatches_pair_up():
    for N, world, bs in ((17, 2, 3), (16, 8, 8), (5, 4, 2), (12, 1, 5)):
        plans = [sharding.local_minibatches(N, r, world, bs) for r in range(world)]
        assert len({len(p) for p in plans}) == 1                      # same number of batches on every rank
        for r, plan in enumerate(plans):                              # every local sample exactly once, in order
            n0, n1 = sharding.shard_bounds(N, r, world)
            covered = [i for s in plan for i in range(*s.indices(n1 - n0))]
            assert covered == list(range(n1 - n0))
    assert sharding.local_minibatches(7, 0, 1, 3) == [slice(0, 3), slice(3, 6), slice(6, 7)]   # reference split


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _spawn(fn, args_of_port, nprocs=2):
    """mp.spawn on a fresh port; the port is probed and released before the ranks bind it, so a rare race with another
    process's ephemeral port gets one more attempt on another port."""
    for attempt in range(2):
        try:
            mp.spawn(fn, args=args_of_port(_free_port()), nprocs=nprocs, join=True)
            return
        except Exception as exc:   # noqa: BLE001
            # only a lost race for the port is worth another port; anything else (an assertion inside a rank, a parity
            # mismatch) is a genuine failure and surfaces on the first attempt
            text = str(exc).lower()
            port_race = any(k in text for k in ('address already in use', 'eaddrinuse', 'errno 98',
                                                'connection refused', 'connection reset'))
            if attempt or not port_race:
                raise


def _worker(rank, world, port, mode, out):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [here, os.path.dirname(here)]
    from oracle_backend import OracleBackend
    from tnmf_amd.TransformInvariantNMF import MiniBatchAlgorithm, TransformInvariantNMF
    os.environ['OMP_NUM_THREADS'] = '2'
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(7)
        V = rng.random((6, 2, 12, 14))
        np.random.seed(42)                       # identical global RNG state on every rank
        nmf = TransformInvariantNMF(n_atoms=3, atom_shape=(3, 4),
                                    backend=OracleBackend(process_group=dist.group.WORLD, hooks=True))
        if mode == 'batch':
            nmf.fit(V, n_iterations=4, sparsity_H=0.05)
        else:
            nmf.fit(V, algorithm=MiniBatchAlgorithm.Cyclic_MU, batch_size=4, n_epochs=3)
        np.savez(os.path.join(out, f'rank{rank}.npz'), W=nmf.W, H=nmf.H, E=nmf._energy_function(),
                 shard=np.array(nmf._backend._shard))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('mode', ['batch', 'cyclic'])
def test_two_ranks_match_single_process(tmp_path, mode):
    from oracle_backend import OracleBackend
    from tnmf_amd.TransformInvariantNMF import MiniBatchAlgorithm, TransformInvariantNMF
    _spawn(_worker, lambda port: (2, port, mode, str(tmp_path)))
    r0, r1 = (np.load(tmp_path / f'rank{r}.npz') for r in range(2))

    rng = np.random.default_rng(7)
    V = rng.random((6, 2, 12, 14))
    np.random.seed(42)
    ref = TransformInvariantNMF(n_atoms=3, atom_shape=(3, 4), backend=OracleBackend(hooks=True))
    if mode == 'batch':
        ref.fit(V, n_iterations=4, sparsity_H=0.05)
    else:
        # 2 ranks x local batch 2 == sequential global batches re-ordered; Cyclic-MU sums all of them per epoch,
        # H updates are per sample, so the result equals the single-process Cyclic-MU run up to summation order
        ref.fit(V, algorithm=MiniBatchAlgorithm.Cyclic_MU, batch_size=2, n_epochs=3)

    assert tuple(r0['shard']) == (0, 3) and tuple(r1['shard']) == (3, 6)
    np.testing.assert_allclose(r0['W'], r1['W'], rtol=0, atol=0)          # replicated W is bit-identical
    np.testing.assert_allclose(r0['W'], ref.W, rtol=1e-12)
    np.testing.assert_allclose(np.concatenate([r0['H'], r1['H']]), ref.H, rtol=1e-12)
    assert np.isclose(float(r0['E']), ref._energy_function(), rtol=1e-12)
    assert float(r0['E']) == float(r1['E'])


def _gather_worker(rank, world, port, out):
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    try:
        t = torch.arange(6, dtype=torch.float32).reshape(2, 3) + 10 * rank
        parts = sharding.all_gather(t, dist.group.WORLD)
        np.save(os.path.join(out, f'gather{rank}.npy'), parts.numpy())
    finally:
        dist.destroy_process_group()


def test_all_gather_is_in_rank_order(tmp_path):
    """The gather of the 'ordered' reduction (HIP_Backend(reduce='ordered')): [world, *shape], rank r at index r."""
    _spawn(_gather_worker, lambda port: (2, port, str(tmp_path)))
    want = np.stack([np.arange(6, dtype=np.float32).reshape(2, 3) + 10 * r for r in range(2)])
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f'gather{r}.npy'), want)
