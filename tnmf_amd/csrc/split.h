// Host-side entry points of split.hip: the H-gradient / fused H-update kernel on the bf16 matrix cores with every
// float32 operand split exactly into three bf16 terms (f32-grade products at 16/6 of the f32 MFMA rate).
#pragma once
#include "common.h"

// shape support: float32, 2-D, atoms up to 16 x 16 (the specialised instantiations are listed in split.hip)
// (every atom up to 16 x 16 runs on the smallest covering instantiation; only_if_worth: not when the zero padding of that
// instantiation costs more than the exact f32 kernels would)
bool split_has_corr_W(const Geo &g, int dtype, bool only_if_worth = false);
// neg/pos of the H gradient (fused == false) or H = H * neg / (pos + reg) in place (fused == true); R is given
// extra (fused, row-padded H only; may be NULL): a further term of the denominator, laid out like H
int split_corr_W(tnmf_hip_ctx *ctx, const Geo &g, const float *V, const float *R, const float *W, float *H_inout,
                 float *neg, float *pos, bool fused, float reg, hipStream_t s, const float *extra = nullptr);
void split_release(tnmf_hip_ctx *ctx);
// per-device kernel attributes (dynamic LDS above 64 KB); once per context after hipSetDevice
int split_prepare_device();
