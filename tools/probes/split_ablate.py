"""
Timing-only ablations of the split H-update kernel (k_split_corr_W) on the DIAGNOSTIC build of the library
(`make -C tnmf_amd/csrc DIAG=1` -> tnmf_amd/lib/libtnmf_hip_diag.so; results are wrong by design, only times matter),
on the data of bench.py (planted model, three real MU iterations first), so that clocks and memory behaviour are those
of the benchmark.
    python tools/probes/split_ablate.py [config] [masks]    masks: 1 window staging, 4 MFMA loop, 8 epilogue, 128 H loads
    TNMF_HIP_STAMPS=1 adds the per-phase cycle stamps (stderr)
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tnmf_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, 'tnmf_amd', 'lib', 'libtnmf_hip_diag.so')
import bench  # noqa: E402
from tnmf_amd.TransformInvariantNMF import TransformInvariantNMF  # noqa: E402

cfg_id = int(sys.argv[1]) if len(sys.argv) > 1 else 3
masks = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0, 1, 8, 128, 136, 137, 5]
cfg = bench.CONFIGS[cfg_id]
dev = torch.device('cuda', 0)
V = bench.synth_V_on_device(cfg, cfg['N'], 1234, dev)
np.random.seed(42)
torch.cuda.manual_seed(4242)
nmf = TransformInvariantNMF(n_atoms=cfg['M'], atom_shape=tuple(cfg['A']), backend='hip', device=dev, path='auto',
                            init='device')
nmf._initialize_matrices(V, keep_W=False)
for _ in range(3):
    nmf._update_H()
    nmf._update_W()
be = nmf._backend
W, H = nmf._W, nmf._H
R = be.reconstruct(W, H)
H0 = H.clone()
g = be._geom(H.shape[0], W.shape[0], be._row_stride(H) or 0)   # (the backend's H has row-padded storage)
p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
be._lib.tnmf_hip_diag_set_ablate.argtypes = [ctypes.c_void_p, ctypes.c_int]
for mask in masks:
    be._lib.tnmf_hip_diag_set_ablate(be._ctx, mask)
    ms = []
    for it in range(7):
        H.copy_(H0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(be._lib.tnmf_hip_update_H(be._ctx, ctypes.byref(g), p(be._V_dev), p(W), p(H), p(R), 1, 1e-9, 0.0,
                                             be._stream()), 'update_H')
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    print(f'ablate {mask:4d}: {np.median(ms[2:]):.3f} ms  (min {min(ms[2:]):.3f})  last_path={be.last_path}', flush=True)
