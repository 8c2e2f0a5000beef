"""The C-ABI library loads and exports every symbol include/tnmf_hip.h declares (no GPU needed, no compute calls)."""
import os
import re

import pytest

from conftest import ROOT
from tnmf_amd import _lib

HEADER = os.path.join(ROOT, 'include', 'tnmf_hip.h')


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(tnmf_hip_[a-z_0-9A-Z]+)\s*\(', text)))


def test_library_exports_header():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f'{n} declared in tnmf_hip.h but not exported'
    assert sorted(_lib.EXPORTS) == names


def test_abi_version_and_strerror():
    lib = _lib.load()
    assert lib.tnmf_hip_abi_version() == _lib.ABI_VERSION
    assert b'geometry' in lib.tnmf_hip_strerror(-2)
    assert lib.tnmf_hip_strerror(0) == b'ok'


def test_no_cpu_fallback():
    """Without a GPU the backend must refuse to construct (and never route through the oracle)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from tnmf_amd.backends.HIP import HIP_Backend
    with pytest.raises(RuntimeError):
        HIP_Backend()
    with pytest.raises(ValueError):
        HIP_Backend(reconstruction_mode='same')          # unknown mode: ValueError like _PyTorchBackend.py:50-52


def test_product_never_imports_oracle():
    """The package must not import (or dlopen) anything under oracle/."""
    pkg = os.path.join(ROOT, 'tnmf_amd')
    pat = re.compile(r'^\s*(from|import)\s+oracle\b|tnmf_oracle|libtnmf_oracle', re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                assert not pat.search(open(os.path.join(dirpath, f)).read()), f


def test_product_library_has_no_diagnostic_switches():
    """Phase ablation, cycle stamps and forced kernel variants exist only in -DTNMF_DIAG builds (make DIAG=1): the
    product library must not look at the environment at all."""
    blob = open(_lib.LIB_PATH, 'rb').read()
    for name in (b'TNMF_HIP_ABLATE', b'TNMF_HIP_STAMPS', b'TNMF_FFT_NO_MIXED', b'TNMF_FFT_NO_RESIDENT',
                 b'TNMF_MIX_GROUPS', b'TNMF_FFT_WINDOW_MB'):
        assert name not in blob, name
    import subprocess
    syms = subprocess.run(['nm', '-D', '--undefined-only', _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert 'getenv' not in syms


def test_array_contract_of_the_reference_front_end_on_torch_tensors():
    """which of the array-object operations the REFERENCE front end applies to backend-native
    arrays (tnmf/TransformInvariantNMF.py:232-235, :271, :447-453, :458, :483, :258-268) a torch.Tensor
    satisfies.  CPU tensors stand in for ROCm tensors: the operator semantics are the same, only `tensor - ndarray`
    additionally fails for a device tensor."""
    import numpy as np
    import torch
    H = torch.rand(4, 3, 9, dtype=torch.float64)
    neg, pos = torch.rand_like(H), torch.rand_like(H)
    s = slice(1, 3)
    view = H[s]
    assert view.data_ptr() == H[1].data_ptr()           # H[s] is a writable view (:271)
    pos += 1e-9                                         # += Python float (:232)
    before = H.clone()
    view *= neg[s]                                      # in-place *=, /= with same-type arrays (:234-235)
    view /= pos[s]
    assert not torch.equal(H[s], before[s]) and torch.equal(H[0], before[0])
    acc = 0 + neg                                       # 0 + arr start values (:458, :483)
    acc += pos                                          # += arrays (:447-448)
    acc *= 0.8                                          # *= float (:450-451)
    acc += 0.2 * neg                                    # float * arr (:452-453)
    assert acc.shape == H.shape
    g = -H                                              # inhibition branch (:258-268): -arr, .sum(axis=1, keepdims=True)
    assert g.sum(axis=1, keepdims=True).shape == (4, 1, 9)
    # `inhibition_gradient - self.H[s]` (:258) subtracts the NDARRAY property self.H from a backend-native array.  A CPU
    # tensor takes that (the reference's own PyTorch backends rely on it); a ROCm tensor raises TypeError (GPU test
    # tests/test_hip_scale.py::test_reference_inhibition_line_needs_backend_native_H), so under the reference's
    # UNMODIFIED front end the inhibition terms need a one-line change (self._H[s] in place of self.H[s]) -- stated in
    # INTEGRATION.md section 2.  This repository's front end uses the backend-native H (TransformInvariantNMF.py:170).
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', DeprecationWarning)
        assert isinstance(H - np.zeros(H.shape), torch.Tensor)
