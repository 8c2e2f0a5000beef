// volume.h -- three shift axes (volumes).  The reference is k-generic in NumPy (backends/NumPy.py:69-132 contracts over
// however many shift axes there are) and dispatches conv1d / conv2d / conv3d in PyTorch (backends/PyTorch.py:13-17).
#pragma once
#include "common.h"

struct Vol {
    int N, M, C;
    int D[3];   // sample shape (z, y, x)
    int A[3];   // atom shape
    int H[3];   // shift shape D + A - 1 (what every kernel works on; the other modes pad to it)
};

int vol_reconstruct(const Vol &v, int dtype, const void *W, const void *H, void *R, hipStream_t s);
// fused: H <- H * neg / (pos + reg) in place; otherwise neg / pos (shaped like H) are written
int vol_corr_W(const Vol &v, int dtype, const void *V, const void *R, const void *W, void *Hio, void *neg, void *pos,
               bool fused, double reg, hipStream_t s);
// partials: P * M * C * |A| * 2 doubles of scratch (P = vol_corr_H_chunks)
int vol_corr_H_chunks(const tnmf_hip_ctx *ctx, const Vol &v);
int vol_corr_H(const Vol &v, int dtype, const void *V, const void *R, const void *H, void *neg, void *pos,
               double *partials, int P, hipStream_t s);
int vol_pad_fold(const tnmf_hip_ctx *ctx, const Vol &v, int dtype, int mode, bool fold, const void *in, void *out,
                 hipStream_t s);
// G [N][M][vox] <- inh * (G - H) + xc * (sum over the atoms of G - G), in place (H laid out like G)
int vol_lateral(const tnmf_hip_ctx *ctx, int dtype, size_t N, int M, size_t vox, void *G, const void *H, double inh,
                double xc, hipStream_t s);
