"""
tnmf_amd -- an MI355X (gfx950) native backend for the multiplicative-update loop of shift-invariant NMF.

    from tnmf_amd.TransformInvariantNMF import TransformInvariantNMF
    nmf = TransformInvariantNMF(n_atoms=32, atom_shape=(12, 12), backend='hip')
    nmf.fit(V, n_iterations=100)

The compute lives in tnmf_amd/lib/libtnmf_hip.so (sources: tnmf_amd/csrc, C ABI: include/tnmf_hip.h).
"""
__version__ = '0.1.0'
