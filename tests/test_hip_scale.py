"""
GPU parity at the sizes that are benchmarked, and the sample-sharded branch of HIP_Backend on one GPU.

  * BASELINE config 3 (256 x 1 x 256 x 256, 32 atoms 12 x 12) end to end: 5 float32 iterations of the default dispatch
    against the float64 C oracle (oracle/tnmf_oracle_c.c, pinned to the reference): W, H and the energy within 1e-5
    (north star: "W,H must match the reference NumPy backend on identical seeds within 1e-5 relative fp32").
  * Cyclic-MU (reference tnmf/TransformInvariantNMF.py:444-465) in float32 at the shard geometries of configs 4 and 5.
  * world_size 2 inside one process: two HIP_Backend objects as rank 0 / rank 1 with an injected collective
    (tests/local_collective.py), full batch and Cyclic-MU with an uneven split and an empty tail batch, against the
    unsharded run and the oracle.
  * the spectrum cache of the FFT family survives a foreign write to H.
"""
import os
import threading

import numpy as np
import pytest
import torch

from local_collective import run_ranks
from oracle import tnmf_oracle as orc

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from tnmf_amd.backends.HIP import HIP_Backend
    from tnmf_amd.TransformInvariantNMF import MiniBatchAlgorithm, TransformInvariantNMF


def relmax(got, want):
    want = np.asarray(want, dtype=np.float64)
    scale = np.abs(want).max()
    return np.abs(np.asarray(got, dtype=np.float64) - want).max() / (scale if scale > 0 else 1.0)


def planted_V(N, C, D, M, A, seed, dtype=np.float32, density=0.01):
    """V = reconstruct(W*, H*) + 0.01 U with a sparse H* (SURVEY 8d's synthetic model), built with the oracle."""
    rng = np.random.default_rng(seed)
    Hs = tuple(d + a - 1 for d, a in zip(D, A))
    Wt = rng.random((M, C) + A)
    Wt /= Wt.sum(axis=tuple(range(-len(A), 0)), keepdims=True)
    V = np.empty((N, C) + D, dtype=dtype)
    for lo in range(0, N, 32):          # bounded temporaries
        n = min(32, N - lo)
        Ht = rng.random((n, M) + Hs) * (rng.random((n, M) + Hs) < density)
        V[lo:lo + n] = orc.reconstruct(Wt, Ht, 'c' if len(A) < 3 else 'contract') + 0.01 * rng.random((n, C) + D)
    return V


def oracle_threads():
    orc.set_threads(orc.default_threads(cap=64))


# full size: 256 samples.  TNMF_TEST_CONFIG3_N overrides the sample count on a box with few host cores (the float64
# oracle needs ~3 thread-seconds per sample and iteration).
CONFIG3_N = int(os.environ.get('TNMF_TEST_CONFIG3_N', '256'))


@pytest.mark.parametrize('N', [CONFIG3_N], ids=[f'N{CONFIG3_N}'])
def test_config3_end_to_end_against_f64_oracle(N):
    C, D, M, A = 1, (256, 256), 32, (12, 12)
    oracle_threads()
    V = planted_V(N, C, D, M, A, seed=1234)
    np.random.seed(42)
    nmf = TransformInvariantNMF(n_atoms=M, atom_shape=A, backend='hip', path='auto')
    nmf.fit(V, n_iterations=5, progress_callback=lambda *_: True)
    W, H, E = nmf.W, nmf.H, nmf._energy_function()
    del nmf
    torch.cuda.empty_cache()
    np.random.seed(42)
    ref = orc.OracleNMF(n_atoms=M, atom_shape=A, impl='c').fit(V.astype(np.float64), n_iterations=5)
    dW, dH = relmax(W, ref.W), relmax(H, ref.H)
    gap = abs(E - ref.energy()) / ref.energy()
    print(f'config 3, N={N}: dW={dW:.2e} dH={dH:.2e} energy gap={gap:.2e}')
    assert dW < 1e-5 and dH < 1e-5 and gap < 1e-5, (dW, dH, gap)


def test_config2_end_to_end_against_f64_oracle():
    """BASELINE configs[1] at its full 64 samples (64 x 1 x 128 x 128, 16 atoms 9 x 9): 5 float32 iterations of the
    default dispatch from the reference's seeded start against the float64 C oracle -- W, H and the energy within 1e-5."""
    N, C, D, M, A = 64, 1, (128, 128), 16, (9, 9)
    oracle_threads()
    V = planted_V(N, C, D, M, A, seed=1234)
    np.random.seed(42)
    nmf = TransformInvariantNMF(n_atoms=M, atom_shape=A, backend='hip', path='auto')
    nmf.fit(V, n_iterations=5, progress_callback=lambda *_: True)
    fams = nmf._backend.last_path
    W, H, E = nmf.W, nmf.H, nmf._energy_function()
    nmf._update_H()
    assert fams == 'fft' and nmf._backend.last_path == 'split'     # the default dispatch at this size (hybrid + split)
    del nmf
    torch.cuda.empty_cache()
    np.random.seed(42)
    ref = orc.OracleNMF(n_atoms=M, atom_shape=A, impl='c').fit(V.astype(np.float64), n_iterations=5)
    dW, dH = relmax(W, ref.W), relmax(H, ref.H)
    gap = abs(E - ref.energy()) / ref.energy()
    print(f'config 2, N={N}: dW={dW:.2e} dH={dH:.2e} energy gap={gap:.2e}')
    assert dW < 1e-5 and dH < 1e-5 and gap < 1e-5, (dW, dH, gap)


def test_refit_across_dtypes_on_the_schedule_path():
    """One model object fitted in float32 and then in float64 (and back) on a problem small enough for the persistent
    schedule kernel (BASELINE config 1's geometry): the gradient accumulator the front end keeps between iterations
    belongs to ONE fit -- a float32 buffer handed to a float64 run would be overrun by the library."""
    rng = np.random.default_rng(3)
    V = rng.random((10, 3, 60))
    nmf = TransformInvariantNMF(n_atoms=8, atom_shape=(20,), backend='hip')
    for dtype, tol in ((np.float32, 1e-5), (np.float64, 1e-10), (np.float32, 1e-5)):
        np.random.seed(42)
        nmf.fit(V.astype(dtype), n_iterations=4, progress_callback=lambda *_: True)
        assert nmf._backend.prefers_schedule(nmf._H)
        assert nmf._iteration_acc is not None and nmf._iteration_acc.dtype == nmf._W.dtype
        np.random.seed(42)
        ref = orc.OracleNMF(n_atoms=8, atom_shape=(20,)).fit(V.astype(dtype).astype(np.float64), n_iterations=4)
        assert relmax(nmf.W, ref.W) < tol and relmax(nmf.H, ref.H) < tol
    # the backend refuses an accumulator of another type outright
    nmf.fit(V, n_iterations=1, progress_callback=lambda *_: True)
    bad = torch.zeros((2,) + tuple(nmf._W.shape), dtype=torch.float32, device=nmf._W.device)
    with pytest.raises(AssertionError):
        nmf._backend.run_schedule(nmf._V, nmf._W, nmf._H, [('G', slice(None), 0., 1.), ('W',)], bad)


@pytest.mark.parametrize('dtype,tol', [(np.float32, 1e-5), (np.float64, 1e-10)], ids=['f32', 'f64'])
def test_persistent_schedule_kernel_is_a_checked_option(dtype, tol):
    """The persistent schedule kernel (one launch per iteration of a tiny problem, grid-wide barriers) needs its whole
    grid resident: the library sizes the grid by an occupancy query and falls back to the per-operation path when told to
    (persistent=0: a caller that shares the GPU) or when the grid cannot be resident.  All three modes -- plain launch of
    the occupancy-sized grid, cooperative launch, forced fallback -- compute the same factorisation (the two launch
    flavours of the persistent kernel bit for bit), and the library reports which way it went."""
    rng = np.random.default_rng(8)
    V = rng.random((10, 3, 60)).astype(dtype)     # BASELINE config 1's geometry
    res = {}
    for mode, want_persistent in ((1, True), (2, True), (0, False)):
        np.random.seed(42)
        nmf = TransformInvariantNMF(n_atoms=8, atom_shape=(20,), backend='hip', persistent=mode)
        nmf.fit(V, n_iterations=6, progress_callback=lambda *_: True)
        assert nmf._backend.prefers_schedule(nmf._H)
        assert nmf._backend.last_schedule_persistent is want_persistent, mode
        res[mode] = (nmf.W, nmf.H)
        nmf.fit(V, algorithm=MiniBatchAlgorithm.ASG_MU, batch_size=3, n_epochs=2, progress_callback=lambda *_: True)
        assert nmf._backend.last_schedule_persistent is want_persistent, mode
        res[mode] += (nmf.W, nmf.H)
    for a, b in zip(res[1], res[2]):
        assert np.array_equal(a, b)               # the same kernel, launched cooperatively: the same bits
    for a, b in zip(res[1], res[0]):
        # the per-operation path may split the pixel sum of the W gradient differently (and picks its kernels per
        # operation): the same arithmetic up to the order of additions
        assert relmax(a, b) < (1e-12 if dtype == np.float64 else 2e-6)
    np.random.seed(42)
    ref = orc.OracleNMF(n_atoms=8, atom_shape=(20,)).fit(V.astype(np.float64), n_iterations=6)
    assert relmax(res[0][0], ref.W) < tol and relmax(res[0][1], ref.H) < tol


@pytest.mark.parametrize('C,D,M,A', [(3, (256, 256), 32, (12, 12)), (3, (512, 512), 64, (16, 16))],
                         ids=['config4_shard_geometry', 'config5_shard_geometry'])
def test_cyclic_mu_f32_at_shard_geometry(C, D, M, A):
    """Cyclic-MU, N = 8, batch 4, 2 epochs, float32 default dispatch vs the float64 oracle's Cyclic-MU."""
    oracle_threads()
    V = planted_V(8, C, D, M, A, seed=77)
    np.random.seed(42)
    nmf = TransformInvariantNMF(n_atoms=M, atom_shape=A, backend='hip')
    nmf.fit(V, algorithm=MiniBatchAlgorithm.Cyclic_MU, batch_size=4, n_epochs=2, progress_callback=lambda *_: True)
    np.random.seed(42)
    ref = orc.OracleNMF(n_atoms=M, atom_shape=A, impl='c').fit(
        V.astype(np.float64), algorithm=orc.MiniBatchAlgorithm.Cyclic_MU, batch_size=4, n_epochs=2)
    dW, dH = relmax(nmf.W, ref.W), relmax(nmf.H, ref.H)
    gap = abs(nmf._energy_function() - ref.energy()) / ref.energy()
    print(f'cyclic {C}x{D} m{M} a{A}: dW={dW:.2e} dH={dH:.2e} gap={gap:.2e}')
    assert dW < 1e-5 and dH < 1e-5 and gap < 1e-5, (dW, dH, gap)


@pytest.mark.parametrize('N,C,D,M,A', [(256, 3, (256, 256), 32, (12, 12)), (128, 3, (512, 512), 64, (16, 16))],
                         ids=['config4_shard', 'config5_shard'])
def test_full_shard_sizes_of_configs_4_and_5(N, C, D, M, A):
    """The per-GPU shards of BASELINE configs 4 and 5 at their FULL sample counts (256 x 3 x 256^2, 128 x 3 x 512^2),
    default dispatch, float32.  The float64 oracle referees a 4-sample window (the H half step of a sample depends on
    that sample and W only; the W gradient is additive over samples); size-independent properties cover the rest:
    additivity of the W gradient over mini-batch slices, mini-batch H updates == the full-batch one, unit atom norms."""
    oracle_threads()
    rng = np.random.default_rng(41)
    V = rng.random((N, C) + D, dtype=np.float32)
    V *= rng.random((N, C) + D, dtype=np.float32) > 0.3            # exact zeros among the samples
    torch.cuda.manual_seed(99)
    np.random.seed(42)
    nmf = TransformInvariantNMF(n_atoms=M, atom_shape=A, backend='hip', init='device')
    nmf._initialize_matrices(V, keep_W=False)
    be = nmf._backend
    win = slice(N // 2 - 1, N // 2 + 3)                                # 4 samples from the middle of the shard
    W0 = be.to_ndarray(nmf._W).astype(np.float64)
    H0w = be.to_ndarray(nmf._H[win]).astype(np.float64)
    H_before = nmf._H.clone()

    # H half step on the whole shard; the window against the oracle
    nmf._update_H(sparsity=0.01)
    assert be.last_path == 'split'
    Vw = V[win].astype(np.float64)
    on, op = orc.gradient_H(Vw, W0, H0w, slice(None), 'c')
    want = H0w * on / (op + 1e-9 + 0.01)
    dH = relmax(be.to_ndarray(nmf._H[win]), want)
    # the same half step batch by batch gives the same bits as the full-batch launch (per-sample independence)
    H_full = nmf._H.clone()
    nmf._H.copy_(H_before)
    for b in be.minibatch_slices(N // 4):
        nmf._update_H(b, sparsity=0.01)
    same = torch.equal(nmf._H, H_full)
    del H_before, H_full

    # W gradient: the window slice against the oracle, and additivity over the slices of the whole shard
    H1w = be.to_ndarray(nmf._H[win]).astype(np.float64)
    part = be.local_gradient_W(V, nmf._W, nmf._H, win)
    gn, gp = orc.gradient_W(Vw, W0, H1w, slice(None), 'c')
    dWn, dWp = relmax(part[0].cpu().numpy(), gn), relmax(part[1].cpu().numpy(), gp)
    whole = be.local_gradient_W(V, nmf._W, nmf._H, slice(None)).double()
    pieces = sum(be.local_gradient_W(V, nmf._W, nmf._H, b).double() for b in be.minibatch_slices(N // 4))
    add = float(((whole - pieces).abs().max() / whole.abs().max()).item())
    be.apply_W(nmf._W, whole.to(nmf._W.dtype), eps=1e-9)
    norms = nmf._W.sum(dim=(-2, -1))
    print(f'{N}x{C}x{D} m{M} a{A}: dH={dH:.2e} dWneg={dWn:.2e} dWpos={dWp:.2e} additivity={add:.2e} same_bits={same}')
    assert dH < 1e-5 and dWn < 1e-5 and dWp < 1e-5, (dH, dWn, dWp)
    assert same, 'mini-batch H updates differ from the full-batch launch'
    assert add < 1e-5
    assert float((norms - 1).abs().max().item()) < 1e-5
    assert np.isfinite(nmf._energy_function())


# ---------------------------------------------------------------------------------------------------------------
# world_size 2 on one GPU
# ---------------------------------------------------------------------------------------------------------------
_init_lock = threading.Lock()


def _fit(V, M, A, mode, pg=None, **kw):
    """One model; the seeded initialisation reads the GLOBAL NumPy RNG (reference behaviour), so concurrent ranks take
    turns for it: each seeds and draws under a lock."""
    nmf = TransformInvariantNMF(n_atoms=M, atom_shape=A, backend='hip', process_group=pg, **kw)
    plain_init = nmf._initialize_matrices

    def seeded_init(V_, keep_W):
        with _init_lock:
            np.random.seed(42)
            plain_init(V_, keep_W)

    nmf._initialize_matrices = seeded_init
    cb = lambda *_: True  # noqa: E731
    if mode == 'batch':
        nmf.fit(V, n_iterations=3, sparsity_H=0.05, progress_callback=cb)
    else:
        nmf.fit(V, algorithm=MiniBatchAlgorithm.Cyclic_MU, batch_size=2, n_epochs=3, sparsity_H=0.05,
                progress_callback=cb)
    return nmf


@pytest.mark.parametrize('mode', ['batch', 'cyclic'])
@pytest.mark.parametrize('dtype,geom,tol', [
    (np.float64, (2, (20, 24), 5, (4, 5)), 1e-10),
    (np.float32, (1, (96, 80), 32, (12, 12)), 1e-5),      # large enough for the hybrid dispatch on every rank
    (np.float64, (1, (6, 7, 9), 3, (2, 3, 3)), 1e-10),    # three shift axes: the volume kernels behind the same sharding
], ids=['f64', 'f32_hybrid', 'f64_volume'])
def test_two_ranks_in_one_process(mode, dtype, geom, tol):
    """HIP_Backend's world > 1 branch: shard bounds, local slices, empty tail batches, the [neg|pos] all-reduce, the
    energy all-reduce.  N = 7 over 2 ranks is an uneven split (4 + 3); Cyclic-MU with batch_size 2 gives every rank
    four local batches of one sample, the last one EMPTY on rank 1."""
    C, D, M, A = geom
    N = 7
    oracle_threads()
    V = planted_V(N, C, D, M, A, seed=5, dtype=dtype, density=0.05)

    def rank_body(rank, coll):
        torch.cuda.set_device(0)
        nmf = _fit(V, M, A, mode, pg=coll)
        be = nmf._backend
        return dict(W=nmf.W, H=nmf.H, E=nmf._energy_function(), shard=be.shard,
                    batches=be.minibatch_slices(2) if mode == 'cyclic' else None, family=be.last_path)

    (r0, r1), group = run_ranks(2, rank_body)
    assert r0['shard'] == (0, 4) and r1['shard'] == (4, 7)
    if mode == 'cyclic':
        assert len(r0['batches']) == len(r1['batches']) == 4
        last = r1['batches'][-1]
        assert last.start == last.stop == 3                               # the empty tail batch
        assert group.calls == 3 + 1                                       # one W all-reduce per epoch + the energy
    else:
        assert group.calls == 3 + 1                                       # one per iteration + the energy
    assert np.array_equal(r0['W'], r1['W'])                               # replicated W is bit-identical
    assert r0['E'] == r1['E']

    single = _fit(V, M, A, mode)
    H2 = np.concatenate([r0['H'], r1['H']])
    np.random.seed(42)
    ref = orc.OracleNMF(n_atoms=M, atom_shape=A, impl='c' if len(A) < 3 else 'contract')
    if mode == 'batch':
        ref.fit(V.astype(np.float64), n_iterations=3, sparsity_H=0.05)
    else:
        ref.fit(V.astype(np.float64), algorithm=orc.MiniBatchAlgorithm.Cyclic_MU, batch_size=2, n_epochs=3,
                sparsity_H=0.05)
    for name, got in (('sharded', (r0['W'], H2, r0['E'])), ('single', (single.W, single.H, single._energy_function()))):
        dW, dH = relmax(got[0], ref.W), relmax(got[1], ref.H)
        gap = abs(got[2] - ref.energy()) / ref.energy()
        assert dW < tol and dH < tol and gap < tol, (name, dW, dH, gap)
    # the shards of H only ever see their own samples: the sharded H equals the single-process H to rounding of W
    assert relmax(H2, single.H) < tol
    if dtype == np.float32:
        assert r0['family'] == 'fft'          # last call = energy -> reconstruct on the FFT family: hybrid was active


@pytest.mark.parametrize('mode', ['batch', 'cyclic'])
def test_ranks_bring_their_own_blocks(mode):
    """HIP_Backend(sharded_input=True): every rank hands in ONLY its own samples (bench.py at N > 1: the global array
    exists on no host).  Blocks of different length (5 + 2 of 7 samples): the ranks learn each other's counts through the
    collective, the global sample range of each is its offset in rank order, the mini-batch plans pair up (the short rank
    gets empty tail batches), and the factorisation equals the one of ranks that were handed the global array."""
    C, D, M, A, N = 2, (20, 24), 5, (4, 5), 7
    V = planted_V(N, C, D, M, A, seed=5, dtype=np.float64, density=0.05)
    cuts = [(0, 5), (5, 7)]

    def rank_body(rank, coll):
        torch.cuda.set_device(0)
        lo, hi = cuts[rank]
        nmf = TransformInvariantNMF(n_atoms=M, atom_shape=A, backend='hip', process_group=coll, sharded_input=True)
        plain_init = nmf._initialize_matrices

        def seeded_init(V_, keep_W):
            # the seeded start of the reference: every rank walks the GLOBAL stream and keeps its samples, which needs
            # the counts of the ranks before it -- a collective, done BEFORE the lock that lets the ranks of this one
            # process take turns at the global NumPy RNG
            nmf._backend.exchange_sample_counts(V_.shape[0])
            with _init_lock:
                np.random.seed(42)
                plain_init(V_, keep_W)

        nmf._initialize_matrices = seeded_init
        cb = lambda *_: True  # noqa: E731
        if mode == 'batch':
            nmf.fit(V[lo:hi], n_iterations=3, sparsity_H=0.05, progress_callback=cb)
        else:
            nmf.fit(V[lo:hi], algorithm=MiniBatchAlgorithm.Cyclic_MU, batch_size=2, n_epochs=3, sparsity_H=0.05,
                    progress_callback=cb)
        be = nmf._backend
        return dict(W=nmf.W, H=nmf.H, E=nmf._energy_function(), shard=be.shard, n=be.n_samples,
                    batches=be.minibatch_slices(2))

    (r0, r1), _group = run_ranks(2, rank_body)
    assert r0['shard'] == (0, 5) and r1['shard'] == (5, 7) and r0['n'] == r1['n'] == 7
    assert len(r0['batches']) == len(r1['batches']) == 5          # local batch 1; rank 1's last three are empty
    assert [b.stop - b.start for b in r1['batches']] == [1, 1, 0, 0, 0]
    assert np.array_equal(r0['W'], r1['W']) and r0['E'] == r1['E']
    np.random.seed(42)
    ref = orc.OracleNMF(n_atoms=M, atom_shape=A, impl='c')
    if mode == 'batch':
        ref.fit(V, n_iterations=3, sparsity_H=0.05)
    else:
        ref.fit(V, algorithm=orc.MiniBatchAlgorithm.Cyclic_MU, batch_size=2, n_epochs=3, sparsity_H=0.05)
    assert relmax(r0['W'], ref.W) < 1e-10 and relmax(np.concatenate([r0['H'], r1['H']]), ref.H) < 1e-10
    assert abs(r0['E'] - ref.energy()) / ref.energy() < 1e-10


@pytest.mark.parametrize('dtype', [np.float32, np.float64], ids=['f32', 'f64'])
def test_ordered_reduction_is_the_rank_order_sum_bit_for_bit(dtype):
    """reduce='ordered' (all-gather, then sum in rank order): three ranks on one GPU; the W gradient every
    rank gets back equals ((p0 + p1) + p2) of the ranks' own partial [neg | pos] buffers BIT FOR BIT, on every rank, and
    the whole fit stays within tolerance of the all-reduce flavour."""
    C, D, M, A, N = 1, (24, 20), 5, (4, 5), 8
    V = planted_V(N, C, D, M, A, seed=11, dtype=dtype, density=0.05)

    def rank_body(rank, coll):
        torch.cuda.set_device(0)
        be = HIP_Backend(process_group=coll, reduce='ordered')
        with _init_lock:
            np.random.seed(42)
            W, H = be.initialize(V, A, M, None, (-2, -1))
        part = be.local_gradient_W(V, W, H, slice(None)).clone()
        neg, pos = be.reconstruction_gradient_W(V, W, H, slice(None))
        return dict(part=part.cpu().numpy(), neg=neg.cpu().numpy(), pos=pos.cpu().numpy())

    res, group = run_ranks(3, rank_body)
    total = res[0]['part'].copy()
    for r in (1, 2):
        total = total + res[r]['part']          # the element type's own addition, rank order
    for r in range(3):
        assert np.array_equal(res[r]['neg'], total[0]) and np.array_equal(res[r]['pos'], total[1]), r

    def fit_body(flavour):
        def body(rank, coll):
            torch.cuda.set_device(0)
            return _fit(V, M, A, 'cyclic', pg=coll, reduce=flavour).W
        return run_ranks(3, body)[0]

    Wo, Wa = fit_body('ordered'), fit_body('all_reduce')
    assert np.array_equal(Wo[0], Wo[1]) and np.array_equal(Wo[0], Wo[2])
    assert relmax(Wo[0], Wa[0]) < (1e-5 if dtype == np.float32 else 1e-12)


# ---------------------------------------------------------------------------------------------------------------
# spectrum cache ownership (stale cache hit after a torch-side write to H)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('path', ['fft', 'hybrid'])
def test_foreign_write_to_H_between_fused_calls(path):
    N, C, D, M, A = 3, 1, (40, 48), 6, (7, 7)
    rng = np.random.default_rng(9)
    V = rng.random((N, C) + D)
    Wn = rng.random((M, C) + A)
    Wn /= Wn.sum(axis=(-2, -1), keepdims=True)
    Hn = rng.random((N, M) + tuple(d + a - 1 for d, a in zip(D, A)))
    be = HIP_Backend(path=path)
    np.random.seed(1)
    be.initialize(V.astype(np.float32), A, M, None, (-2, -1))
    W = torch.from_numpy(Wn.astype(np.float32)).cuda()
    H = torch.from_numpy(Hn.astype(np.float32)).cuda()
    be.fused_update_H(V, W, H, slice(None), sparsity=0., eps=1e-9)       # leaves the row spectra of the new H cached
    on, op = orc.gradient_H(V, Wn, Hn, slice(None), 'c')
    H1 = Hn * on / (op + 1e-9)
    # (path='fft' makes no parity claim on H in float32 -- DESIGN.md 4b; W below it does)
    if path != 'fft':
        assert relmax(be.to_ndarray(H), H1) < 4e-5
    H1 = be.to_ndarray(H).astype(np.float64)                            # continue from what the device holds
    H.mul_(0.5)                                                           # a torch-side write the library cannot see
    H[1].add_(0.25)
    H2 = H1 * 0.5
    H2[1] += 0.25
    Wf = W.clone()
    be.fused_update_W(V, Wf, H, slice(None), eps=1e-9)
    on, op = orc.gradient_W(V, Wn, H2, slice(None), 'c')
    Wo = Wn * on / (op + 1e-9)
    Wo /= Wo.sum(axis=(-2, -1), keepdims=True)
    assert relmax(be.to_ndarray(Wf), Wo) < 4e-5
    # a different tensor at (possibly) the same address: freed and re-allocated by the caching allocator
    ptr = H.data_ptr()
    del H
    Hnew = torch.from_numpy((Hn * 0.75).astype(np.float32)).cuda()
    be.fused_update_W(V, W.clone(), Hnew, slice(None), eps=1e-9)
    got = W.clone()
    be.fused_update_W(V, got, Hnew, slice(None), eps=1e-9)
    on, op = orc.gradient_W(V, Wn, Hn * 0.75, slice(None), 'c')
    Wo = Wn * on / (op + 1e-9)
    Wo /= Wo.sum(axis=(-2, -1), keepdims=True)
    assert relmax(be.to_ndarray(got), Wo) < 4e-5, f'same address: {Hnew.data_ptr() == ptr}'


@pytest.mark.parametrize('path', ['hybrid', 'fft'])
def test_minibatch_slices_share_one_spectrum_cache(path):
    """Cyclic-MU over mini-batch slices of the backend's own H (reference TransformInvariantNMF.py:457-465; per-slice
    caches: NumPy_CachingFFT.py:143-158): every batch is row-transformed ONCE per epoch -- like a full-batch iteration --
    and a torch-side write that hits one batch only is seen (the whole cache is dropped: conservative)."""
    N, C, D, M, A, B = 6, 1, (40, 48), 6, (7, 7), 2
    rng = np.random.default_rng(19)
    V = rng.random((N, C) + D).astype(np.float32)
    be = HIP_Backend(path=path)
    np.random.seed(3)
    W, H = be.initialize(V, A, M, None, (-2, -1))
    Wn, Hn = be.to_ndarray(W).astype(np.float64), be.to_ndarray(H).astype(np.float64)
    Vn = V.astype(np.float64)
    batches = [slice(lo, lo + B) for lo in range(0, N, B)]
    tol_H = np.inf if path == 'fft' else 4e-5  # (path='fft' makes no parity claim on H in float32)

    def epoch():
        total = None
        for b in batches:
            be.fused_update_H(V, W, H, b, sparsity=0., eps=1e-9)
            part = be.local_gradient_W(V, W, H, b)
            total = part if total is None else total.add_(part)
        be.apply_W(W, total, eps=1e-9)

    def oracle_epoch(Wn, Hn):
        neg = pos = 0.
        for b in batches:
            on, op = orc.gradient_H(Vn, Wn, Hn, b, 'c')
            Hn[b] = Hn[b] * on / (op + 1e-9)
            gn, gp = orc.gradient_W(Vn, Wn, Hn, b, 'c')
            neg, pos = neg + gn, pos + gp
        Wn = Wn * neg / (pos + 1e-9)
        return Wn / Wn.sum(axis=(-2, -1), keepdims=True), Hn

    c0 = be.cache_counters
    epoch()
    Wn, Hn = oracle_epoch(Wn, Hn)
    c1 = be.cache_counters
    epoch()
    Wn, Hn = oracle_epoch(Wn, Hn)
    c2 = be.cache_counters
    assert relmax(be.to_ndarray(W), Wn) < 4e-5 and relmax(be.to_ndarray(H), Hn) < tol_H
    per_epoch = c2['h_runs'] - c1['h_runs']
    # hybrid: the direct H update changes H, so each batch is transformed once (for the W half step) and the next
    # epoch's reconstruct of the same batch finds it cached; pure FFT: the fused update leaves the new spectra behind
    assert per_epoch == (len(batches) if path == 'hybrid' else 0), (c0, c1, c2)
    assert c2['h_hits'] - c1['h_hits'] >= len(batches)
    assert c2['v_runs'] == c1['v_runs'], 'the samples never change: their spectra are computed once'
    # a torch-side write that hits ONE batch only
    H[2:4].mul_(0.5)
    Hn[2:4] *= 0.5
    epoch()
    Wn, Hn = oracle_epoch(Wn, Hn)
    assert relmax(be.to_ndarray(W), Wn) < 4e-5 and relmax(be.to_ndarray(H), Hn) < tol_H
    # ... and a write through a slice view of one sample, between the two half steps of a batch
    be.fused_update_H(V, W, H, batches[1], sparsity=0., eps=1e-9)
    on, op = orc.gradient_H(Vn, Wn, Hn, batches[1], 'c')
    Hn[batches[1]] = Hn[batches[1]] * on / (op + 1e-9)
    H[3].add_(0.125)
    Hn[3] += 0.125
    got = be.local_gradient_W(V, W, H, batches[1])
    gn, gp = orc.gradient_W(Vn, Wn, Hn, batches[1], 'c')
    assert relmax(got[0].cpu().numpy(), gn) < 4e-5 and relmax(got[1].cpu().numpy(), gp) < 4e-5
    # the dictionary's spectra are kept between W updates as well: a torch-side write to W must be seen
    W.mul_(1.5)
    Wn = Wn * 1.5
    be.fused_update_H(V, W, H, batches[0], sparsity=0., eps=1e-9)
    on, op = orc.gradient_H(Vn, Wn, Hn, batches[0], 'c')
    Hn[batches[0]] = Hn[batches[0]] * on / (op + 1e-9)
    assert relmax(be.to_ndarray(H), Hn) < tol_H


def test_reflect_mode_rejects_a_pad_as_long_as_the_row():
    """torch's reflect pad needs pad < size (_PyTorchBackend.py:42-52): atom 5 on 4 shifts must fail, not read past H."""
    be = HIP_Backend(reconstruction_mode='reflect')
    V = np.random.default_rng(0).random((1, 1, 4))
    from tnmf_amd._lib import TnmfHipError
    np.random.seed(0)
    W, H = be.initialize(V, (5,), 2, None, (-1,))
    with pytest.raises(TnmfHipError):
        be.reconstruct(W, H)


def test_reference_inhibition_line_needs_backend_native_H():
    """tnmf/TransformInvariantNMF.py:258 computes `inhibition_gradient - self.H[s]` with self.H the ndarray property:
    on a ROCm tensor that raises, which is why INTEGRATION.md asks for `self._H[s]` there (one line) when the
    reference's own front end drives this backend with inhibition > 0; everything else of the array contract holds
    on the device (in-place *=, /=, += with floats and arrays, writable slices)."""
    H = torch.rand(4, 3, 9, device='cuda')
    g = torch.rand_like(H)
    with pytest.raises(TypeError):
        _ = g - H.cpu().numpy()
    view = H[1:3]
    view *= g[1:3]
    view /= (g[1:3] + 1e-9)
    pos = 0 + g
    pos += 1e-9
    pos += 0.2 * g
    pos *= 0.8
    assert view.data_ptr() == H[1].data_ptr() and pos.shape == H.shape


def test_bench_two_rank_path_rehearsal(tmp_path):
    """bench.py's N > 1 path end to end with the real backend: two processes (torch.distributed.run) share the one GPU
    of the test box, the collective goes through gloo (RCCL refuses two ranks on one device).  Not a performance number
    -- it checks that the sharded bench runs, that both ranks take part and that the JSON line is well formed."""
    import json
    import socket
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TNMF_BENCH_DIST_BACKEND='gloo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2',
           '--warmup', '1', '--config', '2', '--samples', '8', '--algorithm', 'cyclic', '--batch-size', '4']
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 2 and line['steps'] == 2 and line['value'] > 0 and line['scaling'] == 'weak'
    assert line['config']['global_samples'] == 16 and line['config']['algorithm'] == 'cyclic'
    assert 'sample-sharded x2' in line['config']['parallelism']
    assert line['rccl_ranks_seen'] == 2 and line['distributed']['backend'] == 'gloo'


def test_bench_line_contract():
    """The one JSON line of `python bench.py` (here on BASELINE config 2, a few steps, a two-second CPU leg): the keys the
    driver reads, the roofline object of the dominant kernel (algorithmic work / measured launch time against the peak, PMC
    traffic from the committed profile with its staleness stamp) and the CPU baseline timed on this box's host cores."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--config', '2', '--steps', '4', '--warmup', '1',
                          '--cpu-budget', '2', '--no-fft-variant'], env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in d, key
    assert d['n_gpus'] == 1 and d['steps'] == 4 and d['warmup'] == 1 and d['higher_is_better'] is True
    assert d['vs_baseline'] is None and d['dtype'] == 'f32' and d['data'] == 'synthetic' and 'workload' in d['config']
    assert abs(d['value'] - 1e3 / d['ms_per_step']) < 1e-6 * d['value']
    r = d['roofline']
    assert r['bound'] in ('hbm', 'mfma') and r['unit'] in ('GB/s', 'TFLOP/s')
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9 and 0 < r['frac'] < 1
    assert 'k_split_corr_W' in r['kernel'] and r['avg_launch_ms'] > 0
    assert r['traffic'] is None or r['traffic'] > 0
    assert r['traffic_file_is_current'] in (True, False, None) and r['rocprof_file_is_current'] in (True, False, None)
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] == os.cpu_count() and c['value'] > 0 and 'samples' in c['sample']
    p = d['parity']
    assert p['W_rel_diff_vs_oracle'] < 1e-5 and p['H_rel_diff_vs_oracle'] < 1e-5 and p['energy_gap_vs_oracle'] < 1e-5


def test_bench_launches_its_own_ranks(tmp_path):
    """The form the driver's SCALE run takes: plain `python bench.py --gpus 2 ...` with no WORLD_SIZE in the environment.
    bench.py starts the two ranks itself as CHILD processes (torch.distributed.run) before it touches the GPU and relays
    rank 0's line; the line carries the weak-scaling headline, the strong-scaling leg of the same configuration (global
    sample count fixed, half per rank, with the one-GPU run of the same problem beside it), the Cyclic-MU legs on the shards
    of configs 4 and 5 (one collective per epoch), and the number of ranks the collective reached.  Two ranks share the one
    GPU of the test box, so the collective is gloo (TNMF_BENCH_DIST_BACKEND) and no number here is a measurement."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env['TNMF_BENCH_DIST_BACKEND'] = 'gloo'
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--config', '2',
           '--samples', '8']
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['scaling'] == 'weak' and line['value'] > 0
    assert line['config']['global_samples'] == 16 and line['config']['samples_per_gpu'] == 8
    assert line['rccl_ranks_seen'] == 2 and line['distributed']['world_size'] == 2
    assert sorted(r['rank'] for r in line['distributed']['ranks']) == [0, 1]
    assert len({r['pid'] for r in line['distributed']['ranks']}) == 2
    st = line['strong_scaling']
    assert st['scaling'] == 'strong' and st['global_samples'] == 8 and st['samples_on_rank0'] == 4 and st['value'] > 0
    assert st['same_problem_on_one_gpu_same_run']['value'] > 0 and st['speedup_over_one_gpu'] > 0
    for leg, n_per_gpu in (('config4_cyclic', 8), ('config5_cyclic', 8)):
        assert line[leg]['global_samples'] == 2 * n_per_gpu and line[leg]['value'] > 0
        assert np.isfinite(line[leg]['energy_after_run'])


# ---------------------------------------------------------------------------------------------------------------
# row-padded activations (tnmf_hip_geom.h_row_stride)
# ---------------------------------------------------------------------------------------------------------------
def _padded_like(Hc, ld):
    """A view with the values of the contiguous [N, M, Hy, Hx] tensor Hc in storage whose rows are ld elements long."""
    store = torch.zeros(tuple(Hc.shape[:3]) + (ld,), dtype=Hc.dtype, device=Hc.device)
    Hp = store[..., :Hc.shape[3]]
    Hp.copy_(Hc)
    return Hp


def test_row_padded_activations_match_contiguous():
    """What initialize() allocates for the hybrid dispatch -- activation rows padded to whole 128-byte lines, H a view
    of that storage -- gives the SAME bits as C-contiguous activations in every primitive and in both fused half steps;
    a kernel family that wants contiguous activations (path='mfma') answers TNMF_E_STRIDE and the backend goes through a
    copy, with the in-place update landing in the padded tensor."""
    rng = np.random.default_rng(21)
    N, C, D, M, A = 4, 1, (96, 80), 32, (12, 12)
    V = rng.random((N, C) + D).astype(np.float32)
    Wn = rng.random((M, C) + A).astype(np.float32)
    Hn = rng.random((N, M, D[0] + A[0] - 1, D[1] + A[1] - 1)).astype(np.float32)
    for path, family in (('auto', 'split'), ('mfma', 'mfma')):
        be = HIP_Backend(path=path)
        np.random.seed(1)
        W0, H0 = be.initialize(V, A, M, None, (-2, -1))
        if path == 'auto':   # the backend's own H is padded: 91 -> 96 floats per row
            assert not H0.is_contiguous() and H0.stride(2) == 96 and H0.stride(1) == H0.shape[2] * 96
            assert tuple(H0.shape) == Hn.shape
        else:
            assert H0.is_contiguous()
        W = torch.from_numpy(Wn).cuda()
        Hc = torch.from_numpy(Hn).cuda()
        Hp = _padded_like(Hc, 96)
        assert not Hp.is_contiguous()
        Rc, Rp = be.reconstruct(W, Hc), be.reconstruct(W, Hp)
        assert torch.equal(Rc, Rp)
        gc, gp = be.reconstruction_gradient_W(V, W, Hc), be.reconstruction_gradient_W(V, W, Hp)
        assert torch.equal(gc[0], gp[0]) and torch.equal(gc[1], gp[1])
        hc, hp = be.reconstruction_gradient_H(V, W, Hc), be.reconstruction_gradient_H(V, W, Hp)
        assert hp[0].is_contiguous() and torch.equal(hc[0], hp[0]) and torch.equal(hc[1], hp[1])
        assert be.reconstruction_energy(V, W, Hc) == be.reconstruction_energy(V, W, Hp)
        Hc2, Hp2 = Hc.clone(), _padded_like(Hc, 96)
        for _ in range(2):   # the second round runs on what the library may have cached about the first
            be.fused_update_H(V, W, Hc2, slice(None), sparsity=0.05, eps=1e-9)
            assert be.last_path == family
            be.fused_update_H(V, W, Hp2, slice(None), sparsity=0.05, eps=1e-9)
            assert be.last_path == family
            assert torch.equal(Hc2, Hp2)
            Wc, Wp = W.clone(), W.clone()
            be.fused_update_W(V, Wc, Hc2, slice(None), eps=1e-9)
            be.fused_update_W(V, Wp, Hp2, slice(None), eps=1e-9)
            assert torch.equal(Wc, Wp)
        # the pad columns of the storage never leak into results: poison them and repeat one reconstruct
        Hp2._base[..., 91:] = float('nan')
        assert torch.equal(be.reconstruct(W, Hp2), be.reconstruct(W, Hc2))
        # a mini-batch slice of the padded tensor is still the padded layout
        be.fused_update_H(V, W, Hc2, slice(1, 3), sparsity=0., eps=1e-9)
        Hp3 = _padded_like(Hc, 96)
        Hp3.copy_(Hp2)
        Hc3 = Hp2.contiguous()
        be.fused_update_H(V, W, Hp3, slice(1, 3), sparsity=0., eps=1e-9)
        be.fused_update_H(V, W, Hc3, slice(1, 3), sparsity=0., eps=1e-9)
        assert torch.equal(Hp3, Hc3)
        del be


def test_front_end_runs_on_row_padded_activations():
    """fit() on the default dispatch: the activations the front end holds are the padded view; W, H and the energy agree
    with the float64 oracle as they do for contiguous activations, and nmf.H comes back as a plain contiguous ndarray."""
    oracle_threads()
    N, C, D, M, A = 4, 1, (96, 80), 32, (12, 12)
    V = planted_V(N, C, D, M, A, seed=5)
    np.random.seed(42)
    nmf = TransformInvariantNMF(n_atoms=M, atom_shape=A, backend='hip')
    nmf.fit(V, n_iterations=5, sparsity_H=0.02, progress_callback=lambda *_: True)
    assert not nmf._H.is_contiguous() and nmf._H.stride(2) == 96
    H = nmf.H
    assert isinstance(H, np.ndarray) and H.flags['C_CONTIGUOUS'] and H.shape == (N, M, 107, 91)
    np.random.seed(42)
    ref = orc.OracleNMF(n_atoms=M, atom_shape=A, impl='c').fit(V.astype(np.float64), n_iterations=5, sparsity_H=0.02)
    dW, dH = relmax(nmf.W, ref.W), relmax(H, ref.H)
    gap = abs(nmf._energy_function() - ref.energy()) / ref.energy()
    print(f'padded rows: dW={dW:.2e} dH={dH:.2e} gap={gap:.2e}')
    assert dW < 1e-5 and dH < 1e-5 and gap < 1e-5, (dW, dH, gap)


@pytest.mark.parametrize('kw', [dict(inhibition_strength=0.1, cross_atom_inhibition_strength=0.05, sparsity_H=0.02),
                                dict(algorithm='asg', batch_size=2, n_epochs=2),
                                dict(keep='W')],
                         ids=['inhibition', 'stochastic_minibatches', 'refit_keeping_W'])
def test_front_end_branches_on_row_padded_activations(kw):
    """The front-end branches that touch H outside the fused kernels -- inhibition (separable convolutions of H, sums
    over the atom axis, `g - H[s]`), a stochastic mini-batch schedule (H slices, accumulators) and a second fit that
    keeps W -- on the padded view, against the float64 oracle."""
    oracle_threads()
    N, C, D, M, A = 4, 1, (96, 80), 32, (12, 12)
    V = planted_V(N, C, D, M, A, seed=9)
    hip_kw, ref_kw = dict(kw), dict(kw)
    if kw.get('algorithm') == 'asg':
        hip_kw['algorithm'] = MiniBatchAlgorithm.ASG_MU
        ref_kw['algorithm'] = orc.MiniBatchAlgorithm.ASG_MU
    else:
        hip_kw.setdefault('n_iterations', 3)
        ref_kw.setdefault('n_iterations', 3)
    refit = hip_kw.pop('keep', None) is not None
    ref_kw.pop('keep', None)
    cb = lambda *_: True  # noqa: E731
    np.random.seed(42)
    nmf = TransformInvariantNMF(n_atoms=M, atom_shape=A, backend='hip')
    nmf.fit(V, progress_callback=cb, **hip_kw)
    if refit:
        nmf.fit(V, keep_W=True, progress_callback=cb, **hip_kw)
    assert not nmf._H.is_contiguous() and nmf._H.stride(2) == 96
    np.random.seed(42)
    ref = orc.OracleNMF(n_atoms=M, atom_shape=A, impl='c')
    ref.fit(V.astype(np.float64), **ref_kw)
    if refit:
        ref.fit(V.astype(np.float64), keep_W=True, **ref_kw)
    dW, dH = relmax(nmf.W, ref.W), relmax(nmf.H, ref.H)
    gap = abs(nmf._energy_function() - ref.energy()) / ref.energy()
    print(f'{kw}: dW={dW:.2e} dH={dH:.2e} gap={gap:.2e}')
    assert dW < 1e-5 and dH < 2e-5 and gap < 1e-5, (dW, dH, gap)


@pytest.mark.parametrize('path,mode,lateral', [
    ('auto', 'valid', True),        # split kernel with the extra-term epilogue (row-padded activations)
    ('mfma', 'valid', True),        # f32 MFMA family: unfused gradient + one update kernel
    ('generic', 'valid', True),     # generic fused kernel with the extra term
    ('fft', 'valid', True),         # FFT family: unfused gradient + one update kernel
    ('auto', 'circular', True), ('auto', 'reflect', False), ('auto', 'full', True),
    ('auto16', 'valid', True),      # 16 x 16 atoms, three channels: the eight-wave workgroups of the split kernel, extra term
], ids=lambda v: str(v))
def test_lateral_terms_and_modes_run_inside_the_library(path, mode, lateral):
    """TransformInvariantNMF._update_H in full (reference :246-271) through tnmf_hip_update_H_ex: lateral inhibition and
    cross-atom inhibition as an extra term of the fused update's denominator, the reconstruction modes with pad, gradient,
    fold and update as kernels of the library -- float32, against the float64 oracle's front end."""
    oracle_threads()
    N, C, D, M, A = 4, 1, (96, 80), 32, (12, 12)
    if path == 'auto16':
        path, (N, C, D, M, A) = 'auto', (6, 3, (90, 100), 33, (16, 16))
    V = planted_V(N, C, D, M, A, seed=13)
    kw = dict(n_iterations=3, sparsity_H=0.02)
    if lateral:
        kw.update(inhibition_strength=0.1, cross_atom_inhibition_strength=0.05)
    np.random.seed(42)
    nmf = TransformInvariantNMF(n_atoms=M, atom_shape=A, backend='hip', path=path, reconstruction_mode=mode,
                                inhibition_range=(3, 5))
    calls = []
    inner = nmf._backend._lib.tnmf_hip_update_H_ex

    class Spy:   # (ctypes function objects cannot be patched in place)
        def __getattr__(self, name):
            if name == 'tnmf_hip_update_H_ex':
                def counted(*a):
                    calls.append(1)
                    return inner(*a)
                return counted
            return getattr(lib, name)

    lib = nmf._backend._lib
    nmf._backend._lib = Spy()
    nmf.fit(V, progress_callback=lambda *_: True, **kw)
    nmf._backend._lib = lib
    assert len(calls) == 3, 'the H half steps did not go through tnmf_hip_update_H_ex'
    np.random.seed(42)
    ref = orc.OracleNMF(n_atoms=M, atom_shape=A, impl='c', reconstruction_mode=mode, inhibition_range=(3, 5))
    ref.fit(V.astype(np.float64), **kw)
    dW, dH = relmax(nmf.W, ref.W), relmax(nmf.H, ref.H)
    gap = abs(nmf._energy_function() - ref.energy()) / ref.energy()
    print(f'{path} {mode} lateral={lateral}: dW={dW:.2e} dH={dH:.2e} gap={gap:.2e}')
    assert dW < 1e-5 and gap < 1e-5, (dW, dH, gap)
    if path != 'fft':        # (path='fft' makes no parity claim on H in float32)
        assert dH < 1e-5, dH


@pytest.mark.parametrize('algorithm', ['Cyclic_MU', 'ASG_MU', 'GSG_MU', 'ASAG_MU', 'GSAG_MU'])
@pytest.mark.parametrize('dtype,tol', [(np.float64, 1e-11), (np.float32, 1e-5)], ids=['f64', 'f32'])
def test_schedules_in_one_call_match_the_batch_by_batch_path(algorithm, dtype, tol):
    """tnmf_hip_run_schedule (one library call per epoch) against the same schedule driven batch by batch from Python and
    against the oracle's front end (reference TransformInvariantNMF.py:444-504): small batches (2 of 7 samples, a ragged
    last batch), 3 epochs, sag_lambda 0.8 -- the accumulator of ASAG / GSAG persists across the calls."""
    N, C, D, M, A = 7, 2, (20, 24), 5, (4, 5)
    V = planted_V(N, C, D, M, A, seed=21, dtype=dtype, density=0.05)
    kw = dict(batch_size=2, n_epochs=3, sag_lambda=0.8, sparsity_H=0.05)
    got = {}
    for flavour in ('one_call', 'batch_by_batch'):
        np.random.seed(42)
        nmf = TransformInvariantNMF(n_atoms=M, atom_shape=A, backend='hip')
        nmf._use_schedules = flavour == 'one_call'
        calls = []
        if flavour == 'one_call':
            inner = nmf._backend.run_schedule
            nmf._backend.run_schedule = lambda *a, **k: (calls.append(len(a[3])), inner(*a, **k))[1]
        nmf.fit(V, algorithm=getattr(MiniBatchAlgorithm, algorithm), progress_callback=lambda *_: True, **kw)
        if flavour == 'one_call':
            assert len(calls) == 3 and min(calls) >= 4, calls          # one call per epoch, a whole epoch of operations
        got[flavour] = (nmf.W, nmf.H, nmf._energy_function())
    np.random.seed(42)
    ref = orc.OracleNMF(n_atoms=M, atom_shape=A, impl='c')
    ref.fit(V.astype(np.float64), algorithm=getattr(orc.MiniBatchAlgorithm, algorithm), **kw)
    for name, (W, H, E) in got.items():
        dW, dH, gap = relmax(W, ref.W), relmax(H, ref.H), abs(E - ref.energy()) / ref.energy()
        assert dW < tol and dH < tol and gap < tol, (name, dW, dH, gap)
    assert relmax(got['one_call'][0], got['batch_by_batch'][0]) < (1e-13 if dtype == np.float64 else 1e-6)


@pytest.mark.parametrize('dtype,tol', [(np.float64, 1e-11), (np.float32, 2e-6)], ids=['f64', 'f32'])
def test_runs_of_h_steps_are_joined_only_when_they_commute(dtype, tol):
    """tnmf_hip_run_schedule executes a run of consecutive H half steps on pairwise DISJOINT sample ranges as the H half
    step of their union (GSG / GSAG: one launch chain per epoch instead of one per batch) -- the same result as one call
    per step, in any order of the steps; a run with OVERLAPPING ranges is left alone (an H step is not idempotent: the
    second step on a sample must see the first)."""
    N, C, D, M, A = 40, 1, (32, 32), 10, (7, 7)     # (large enough not to be a 'tiny' problem: the per-operation path)
    V = planted_V(N, C, D, M, A, seed=31, dtype=dtype, density=0.05)

    def fresh():
        be = HIP_Backend()
        np.random.seed(3)
        W, H = be.initialize(V, A, M, None, (-2, -1))
        return be, W, H

    def steps(ops_lists):
        be, W, H = fresh()
        acc = be.new_gradient_accumulator(W)
        for ops in ops_lists:
            be.run_schedule(V, W, H, ops, acc, sparsity=0.05)
        return be.to_ndarray(H), be.to_ndarray(W)

    tail = [('G', slice(5, 9), 0., 1.), ('W',)]
    # disjoint, shuffled, with a gap (samples 20..24 are not updated) and an empty slice
    run = [('H', slice(30, 40)), ('H', slice(0, 5)), ('H', slice(9, 9)), ('H', slice(10, 20)), ('H', slice(5, 10)),
           ('H', slice(25, 30))]
    H1, W1 = steps([run + tail])
    H2, W2 = steps([[op] for op in run] + [tail])
    assert relmax(H1, H2) < tol and relmax(W1, W2) < tol
    be0, _, H0 = fresh()
    assert np.array_equal(H1[20:25], be0.to_ndarray(H0)[20:25])              # the gap was left alone
    # overlapping: samples 10..14 are updated twice, in order
    lap = [('H', slice(0, 15)), ('H', slice(10, 25))]
    H3, _ = steps([lap])
    H4, _ = steps([[lap[0]], [lap[1]]])
    H5, _ = steps([[('H', slice(0, 25))]])
    assert relmax(H3, H4) < tol
    assert relmax(H3[10:15], H5[10:15]) > 1e-3                               # (... which is NOT what one joined step gives)


def test_inhibition_kernels_beyond_the_fused_kernel_fall_back_to_the_reference_lines():
    """A 127-tap inhibition kernel (range 63) in float64 does not fit the LDS tile of the lateral-term kernel: the library
    answers TNMF_E_UNSUPPORTED before touching H, the backend turns that into NotImplementedError and the front end walks
    the reference's own lines (TransformInvariantNMF.py:253-269) on the backend's primitives -- same result."""
    N, C, D, M, A = 2, 1, (40, 44), 4, (5, 5)
    V = planted_V(N, C, D, M, A, seed=4, dtype=np.float64, density=0.05)
    kw = dict(n_iterations=3, inhibition_strength=0.2, cross_atom_inhibition_strength=0.1)
    np.random.seed(42)
    nmf = TransformInvariantNMF(n_atoms=M, atom_shape=A, backend='hip', inhibition_range=63)
    nmf.fit(V, progress_callback=lambda *_: True, **kw)
    np.random.seed(42)
    ref = orc.OracleNMF(n_atoms=M, atom_shape=A, impl='c', inhibition_range=63)
    ref.fit(V, **kw)
    assert relmax(nmf.W, ref.W) < 1e-10 and relmax(nmf.H, ref.H) < 1e-10
