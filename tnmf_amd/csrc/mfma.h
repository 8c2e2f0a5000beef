// Host-side entry points of mfma.hip: the f32 MFMA kernel family for gfx950.
#pragma once
#include "common.h"

// per-primitive shape support of the MFMA family (float32 only)
bool mfma_has_reconstruct(const Geo &g, int dtype);
bool mfma_has_corr_W(const Geo &g, int dtype);
bool mfma_has_corr_H(const Geo &g, int dtype);
int mfma_reconstruct(tnmf_hip_ctx *ctx, const Geo &g, const float *W, const float *H, float *R, hipStream_t s);
int mfma_corr_W(tnmf_hip_ctx *ctx, const Geo &g, const float *V, const float *R, const float *W, float *H_inout,
                float *neg, float *pos, bool fused, float reg, hipStream_t s);
int mfma_corr_H_chunks(const tnmf_hip_ctx *ctx, const Geo &g);
int mfma_corr_H(tnmf_hip_ctx *ctx, const Geo &g, const float *V, const float *R, const float *H, double *partials,
                int P, hipStream_t s);
// sets the per-device kernel attributes (dynamic LDS above 64 KB); call once per context after hipSetDevice
int mfma_prepare_device();
