"""
Timing-only ablations of the split H-update kernel (k_split_corr_W) on the DIAGNOSTIC build of the library
(`make -C tnmf_amd/csrc DIAG=1` -> tnmf_amd/lib/libtnmf_hip_diag.so; results are wrong by design, only times matter).
    python tools/probes/split_ablate.py [config]      masks: 1 window staging, 4 MFMA loop, 8 epilogue, 128 H loads
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tnmf_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, 'tnmf_amd', 'lib', 'libtnmf_hip_diag.so')
from tnmf_amd.backends.HIP import HIP_Backend  # noqa: E402

CONFIGS = {3: (256, 1, (256, 256), 32, (12, 12)), 4: (256, 3, (256, 256), 32, (12, 12)), 5: (128, 3, (512, 512), 64, (16, 16)),
           2: (64, 1, (128, 128), 16, (9, 9))}
N, C, D, M, A = CONFIGS[int(sys.argv[1]) if len(sys.argv) > 1 else 3]
masks = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0, 1, 4, 8, 128, 136, 137, 5, 141]
V = np.random.default_rng(0).random((N, C) + D).astype(np.float32)
for mask in masks:
    os.environ['TNMF_HIP_ABLATE'] = str(mask)
    be = HIP_Backend(path='split', init='device')
    W, H = be.initialize(V, A, M, None, (-2, -1))
    be._R_scratch.uniform_(0.5, 1.0)
    ms = []
    for it in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        be._lib.tnmf_hip_reconstruct  # noqa: B018
        R = be._R_scratch
        g = be._geom(N, M)
        import ctypes
        e0.record()
        _lib.check(be._lib.tnmf_hip_update_H(be._ctx, ctypes.byref(g), ctypes.c_void_p(be._V_dev.data_ptr()),
                                             ctypes.c_void_p(W.data_ptr()), ctypes.c_void_p(H.data_ptr()),
                                             ctypes.c_void_p(R.data_ptr()), 1, 1e-9, 0.0, be._stream()), 'update_H')
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    print(f'ablate {mask:4d}: {np.median(ms[2:]):.3f} ms  (min {min(ms[2:]):.3f})  last_path={be.last_path}', flush=True)
    del be, W, H
    torch.cuda.empty_cache()
