"""
ctypes binding of libtnmf_hip.so (C ABI: include/tnmf_hip.h).

There is no CPU fallback: if the shared library is missing or does not export the full ABI, importing a symbol
raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C tnmf_amd/csrc``.
"""
import ctypes
import os
from typing import Optional, Sequence

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libtnmf_hip.so')

ABI_VERSION = 7

# every symbol include/tnmf_hip.h declares
EXPORTS = (
    'tnmf_hip_abi_version', 'tnmf_hip_strerror', 'tnmf_hip_ctx_create', 'tnmf_hip_ctx_destroy',
    'tnmf_hip_ctx_reserve', 'tnmf_hip_ctx_set_path', 'tnmf_hip_ctx_last_path', 'tnmf_hip_reconstruct',
    'tnmf_hip_grad_H', 'tnmf_hip_grad_W', 'tnmf_hip_mu_update', 'tnmf_hip_normalize_W', 'tnmf_hip_energy',
    'tnmf_hip_convolve_multi_1d', 'tnmf_hip_update_H', 'tnmf_hip_grad_W_fused', 'tnmf_hip_apply_W',
    'tnmf_hip_pad_H', 'tnmf_hip_fold_H', 'tnmf_hip_ctx_set_cache', 'tnmf_hip_ctx_invalidate',
    'tnmf_hip_ctx_set_split', 'tnmf_hip_ctx_h_row_stride', 'tnmf_hip_ctx_bind', 'tnmf_hip_ctx_set_persistent',
    'tnmf_hip_ctx_last_schedule_persistent',
    'tnmf_hip_ctx_cache_counters', 'tnmf_hip_sum_parts',
    'tnmf_hip_update_H_ex', 'tnmf_hip_run_schedule', 'tnmf_hip_axpby', 'tnmf_hip_convolve_axis',
)

MODES = {'valid': 0, 'full': 1, 'circular': 2, 'reflect': 3}

PATHS = {'auto': 0, 'generic': 1, 'mfma': 2, 'fft': 3, 'hybrid': 4, 'split': 5}


class Geom(ctypes.Structure):
    """tnmf_hip_geom"""
    _fields_ = [('ndim', ctypes.c_int), ('N', ctypes.c_int), ('M', ctypes.c_int), ('C', ctypes.c_int),
                ('D', ctypes.c_int * 3), ('A', ctypes.c_int * 3), ('dtype', ctypes.c_int),
                ('h_row_stride', ctypes.c_int)]


class Op(ctypes.Structure):
    """tnmf_hip_op"""
    _fields_ = [('kind', ctypes.c_int), ('n0', ctypes.c_int), ('n1', ctypes.c_int), ('a', ctypes.c_double),
                ('b', ctypes.c_double)]


OP_UPDATE_H, OP_GRAD_W, OP_APPLY_W = 0, 1, 2

E_UNSUPPORTED = -5
E_STRIDE = -6   # TNMF_E_STRIDE: the kernel family of this call wants C-contiguous H


class TnmfHipError(RuntimeError):
    def __init__(self, code: int, where: str, text: str):
        super().__init__(f'{where} failed with code {code}: {text}')
        self.code = code


_lib: Optional[ctypes.CDLL] = None


def load() -> ctypes.CDLL:
    """Load the library once; fail loudly when it is absent or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'{LIB_PATH} not found: the HIP extension is not built. The hip backend has no CPU fallback. '
            'Run `make -C tnmf_amd/csrc` (needs hipcc) first.')
    lib = ctypes.CDLL(LIB_PATH)
    missing = [name for name in EXPORTS if not hasattr(lib, name)]
    if missing:
        raise RuntimeError(f'{LIB_PATH} does not export {missing}')
    vp, ci, cd, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_size_t
    gp = ctypes.POINTER(Geom)
    lib.tnmf_hip_abi_version.restype = ci
    lib.tnmf_hip_strerror.restype = ctypes.c_char_p
    lib.tnmf_hip_strerror.argtypes = [ci]
    lib.tnmf_hip_ctx_create.argtypes = [ci, ctypes.POINTER(vp)]
    lib.tnmf_hip_ctx_destroy.argtypes = [vp]
    lib.tnmf_hip_ctx_reserve.argtypes = [vp, gp]
    lib.tnmf_hip_ctx_set_path.argtypes = [vp, ci]
    lib.tnmf_hip_ctx_set_cache.argtypes = [vp, ci]
    lib.tnmf_hip_ctx_set_split.argtypes = [vp, ci]
    lib.tnmf_hip_ctx_set_persistent.argtypes = [vp, ci]
    lib.tnmf_hip_ctx_last_schedule_persistent.argtypes = [vp]
    lib.tnmf_hip_ctx_last_schedule_persistent.restype = ci
    lib.tnmf_hip_ctx_h_row_stride.argtypes = [vp, gp, ctypes.POINTER(ci)]
    lib.tnmf_hip_ctx_invalidate.argtypes = [vp]
    lib.tnmf_hip_ctx_bind.argtypes = [vp, gp, vp, vp]
    lib.tnmf_hip_ctx_cache_counters.argtypes = [vp, ctypes.POINTER(ctypes.c_ulonglong * 4)]
    lib.tnmf_hip_ctx_last_path.restype = ctypes.c_char_p
    lib.tnmf_hip_ctx_last_path.argtypes = [vp]
    lib.tnmf_hip_reconstruct.argtypes = [vp, gp, vp, vp, vp, vp]
    lib.tnmf_hip_grad_H.argtypes = [vp, gp, vp, vp, vp, vp, vp, vp, vp]
    lib.tnmf_hip_grad_W.argtypes = [vp, gp, vp, vp, vp, vp, vp, vp, vp]
    lib.tnmf_hip_mu_update.argtypes = [vp, ci, vp, vp, vp, cd, sz, vp]
    lib.tnmf_hip_normalize_W.argtypes = [vp, gp, vp, vp]
    lib.tnmf_hip_energy.argtypes = [vp, gp, vp, vp, vp, ctypes.POINTER(cd), vp]
    lib.tnmf_hip_convolve_multi_1d.argtypes = [vp, ci, ci, sz, ctypes.POINTER(ci), vp, vp, vp,
                                               ctypes.POINTER(cd), ci, ctypes.POINTER(cd), ci, vp]
    lib.tnmf_hip_convolve_axis.argtypes = [vp, ci, sz, ci, sz, vp, vp, ctypes.POINTER(cd), ci, vp]
    lib.tnmf_hip_update_H.argtypes = [vp, gp, vp, vp, vp, vp, ci, cd, cd, vp]
    lib.tnmf_hip_update_H_ex.argtypes = [vp, gp, ci, vp, vp, vp, vp, cd, cd, cd, cd, ctypes.POINTER(cd), ci,
                                         ctypes.POINTER(cd), ci, ctypes.POINTER(cd), ci, vp]
    lib.tnmf_hip_run_schedule.argtypes = [vp, gp, vp, vp, vp, vp, vp, ctypes.POINTER(Op), ci, cd, cd, vp]
    lib.tnmf_hip_grad_W_fused.argtypes = [vp, gp, vp, vp, vp, vp, ci, vp, vp]
    lib.tnmf_hip_apply_W.argtypes = [vp, gp, vp, vp, cd, vp]
    lib.tnmf_hip_sum_parts.argtypes = [vp, ci, vp, ci, sz, vp, vp]
    lib.tnmf_hip_axpby.argtypes = [vp, ci, vp, vp, cd, cd, sz, vp]
    lib.tnmf_hip_pad_H.argtypes = [vp, gp, ci, vp, vp, vp]
    lib.tnmf_hip_fold_H.argtypes = [vp, gp, ci, vp, vp, vp]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if fn.restype is ctypes.c_int or name not in ('tnmf_hip_strerror', 'tnmf_hip_ctx_last_path'):
            fn.restype = ci
    if lib.tnmf_hip_abi_version() != ABI_VERSION:
        raise RuntimeError(f'{LIB_PATH}: ABI version {lib.tnmf_hip_abi_version()} != expected {ABI_VERSION}')
    _lib = lib
    return lib


def check(code: int, where: str) -> None:
    if code != 0:
        text = load().tnmf_hip_strerror(code)
        raise TnmfHipError(code, where, text.decode() if text else '?')


def make_geom(n: int, m: int, c: int, sample_shape: Sequence[int], atom_shape: Sequence[int], dtype_code: int,
              h_row_stride: int = 0) -> Geom:
    k = len(atom_shape)
    if k not in (1, 2, 3) or len(sample_shape) != k:
        raise NotImplementedError('the hip backend supports 1, 2 or 3 shift dimensions')
    g = Geom()
    g.ndim, g.N, g.M, g.C, g.dtype = k, int(n), int(m), int(c), int(dtype_code)
    for i in range(k):
        g.D[i] = int(sample_shape[i])
        g.A[i] = int(atom_shape[i])
    g.h_row_stride = int(h_row_stride)   # 0: C-contiguous activations
    return g
