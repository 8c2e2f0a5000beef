// Host-side entry points of the FFT kernel family (fft.hip + one fft_len.hip object per transform length).
//
// FFT formulation of the three primitives in 'valid' mode (the reference's counterpart is its
// default backend, backends/NumPy_FFT.py:16-40 with the slice tables of backends/_NumPyFFTBackend.py:43-88).  This is
// not that code: transforms are our own mixed-radix LDS kernels (fft_engine.h) with the spectral contractions, the
// crop, the multiplicative update and the next forward transform fused into them, so that one MU iteration moves
// about ten activation-sized streams through HBM instead of the ~30 a library FFT pipeline needs.
#pragma once
#include "common.h"

// one argument block for every kernel of the family; the meaning of the generic slots is documented per op in
// fft_kernels.h
struct FftArgs {
    const void *src0, *src1, *src2;
    void *dst0, *dst1;
    long ps_src, ps_dst;     // plane strides (elements of the respective array)
    int planes, rows, cols;  // rows / cols that hold data (the rest of the transform length is zero padding)
    int ld_src, ld_dst;      // row strides of real arrays
    int KX, KXP;             // stored kx count (Lx/2+1) and the row stride of every spectrum-side array
    int yoff, xoff;          // crop offsets of the inverse kernels
    int N, M, C, Hy;         // contraction kernels
    int n0, ngroups, nper;   // sample window / split of the sample sum
    int mgroups, mper;       // split of the atom loop of the H-gradient kernel over blocks
    int clamp0;              // plain inverse row kernel: clamp the output at zero from below
    double reg;              // eps (+ sparsity) of the fused update
};

enum FftOp {
    kFftRowsFwd = 0,   // real rows -> row spectra T
    kFftRowsInv,       // row spectra -> real rows (cropped)
    kFftRowsInv2,      // two row spectra (neg, pos) -> two real arrays
    kFftRowsMu,        // two row spectra + H: H = H*neg/(pos+reg) in place, then row spectra of the new H
    kFftColsFwd,       // row spectra -> full spectra S (ky in digit-reversed order)
    kFftColsInv,       // full spectra -> row spectra (cropped rows)
    kFftContractR,     // T (all atoms of a sample) x W spectra -> R spectra
    kFftGradH,         // (V, R) spectra x flipped-W spectra -> row spectra of neg, pos for every atom
    kFftGradW,         // T x conj (V, R) spectra summed over a group of samples -> partial W-gradient spectra
};

typedef int (*fft_run_fn)(int op, int dtype, const FftArgs *a, hipStream_t s);
// per-length objects (fft_len.hip compiled with -DTNMF_FFT_L=<L>)
#define TNMF_FFT_DECL(L) int fft_run_##L(int op, int dtype, const FftArgs *a, hipStream_t s)
TNMF_FFT_DECL(32);
TNMF_FFT_DECL(48);
TNMF_FFT_DECL(64);
TNMF_FFT_DECL(96);
TNMF_FFT_DECL(144);
TNMF_FFT_DECL(192);
TNMF_FFT_DECL(270);
TNMF_FFT_DECL(288);
TNMF_FFT_DECL(384);
TNMF_FFT_DECL(540);
TNMF_FFT_DECL(576);
#undef TNMF_FFT_DECL

// mixed contractions (fft_mixed.hip): transform along x only, the atom rows directly
bool mixed_has_reconstruct(const Geo &g, int dtype);
bool mixed_has_grad_W(const Geo &g, int dtype);
int mixed_reconstruct(const Geo &g, const void *Tsp, const void *WT, void *OT, int KX, int KXP, hipStream_t s);
int mixed_grad_W(const Geo &g, const void *Tsp, const void *VT, const void *RT, void *Gn, void *Gp, int KX, int KXP,
                 int ngroups, int nper, hipStream_t s);

// streaming contractions on resident full spectra of H (fft_spectral.hip): no transform inside
int spectral_contract_R(const Geo &g, int dtype, const void *SH, const void *SW, void *SR, int Ly, int KX, int KXP,
                        hipStream_t s);
int spectral_grad_W(const Geo &g, int dtype, const void *SH, const void *SV, const void *SR, void *Gn, void *Gp, int Ly,
                    int KX, int KXP, int ngroups, int nper, hipStream_t s);

// shape support (2-D problems, transform length available for the activation shape, dtype instantiated)
bool fft_has(const Geo &g, int dtype);
void fft_invalidate(tnmf_hip_ctx *ctx);     // H and V may have changed (every sample of the binding)
// the g.N samples at H have changed (a slice of the bound activations: only their spectra are dropped; anything else: all)
void fft_invalidate_H(tnmf_hip_ctx *ctx, const Geo &g, int dtype, const void *H);
// The resident problem whose spectra the workspace mirrors (tnmf_hip_ctx_bind): g.N samples of activations at H and of
// samples at V.  Calls on whole-sample slices of it share one cache with per-sample validity.
void fft_bind(tnmf_hip_ctx *ctx, const Geo &g, int dtype, const void *H, const void *V);
void fft_unbind(tnmf_hip_ctx *ctx);
void fft_invalidate_W(tnmf_hip_ctx *ctx);   // the dictionary has changed: its cached spectra are stale
void fft_release(tnmf_hip_ctx *ctx);
// pre-size the family's workspace for `g` (with_window: including the buffers of the fused FFT H update)
int fft_reserve(tnmf_hip_ctx *ctx, const Geo &g, int dtype, bool with_window);

// nonneg: clamp R at zero from below (W, H >= 0 make R >= 0; this removes transform rounding noise below zero so
// that the direct H-gradient kernel keeps its invariant pos >= 0 -- used by the hybrid dispatch)
int fft_reconstruct(tnmf_hip_ctx *ctx, const Geo &g, int dtype, const void *W, const void *H, void *R, bool nonneg,
                    hipStream_t s);
// neg/pos of the H gradient from V and a given R
int fft_grad_H(tnmf_hip_ctx *ctx, const Geo &g, int dtype, const void *V, const void *R, const void *W, void *neg,
               void *pos, hipStream_t s);
// fused: H = H*neg/(pos+reg) in place (R given); leaves the row spectra of the new H cached
int fft_update_H(tnmf_hip_ctx *ctx, const Geo &g, int dtype, const void *V, const void *R, const void *W, void *H,
                 double reg, hipStream_t s);
// neg/pos of the W gradient (reference orientation) from V, a given R and H
// nonneg: clamp neg/pos at zero from below (sums of non-negative products: removes transform rounding noise)
int fft_grad_W(tnmf_hip_ctx *ctx, const Geo &g, int dtype, const void *V, const void *R, const void *H, void *neg,
               void *pos, bool nonneg, hipStream_t s);
