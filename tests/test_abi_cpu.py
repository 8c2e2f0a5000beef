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
