// One translation unit per (atom rows, runs of four taps per atom row) instantiation of the split H-gradient kernel
// (compiled with -DTNMF_SPLIT_AY=<rows> -DTNMF_SPLIT_NR4=<runs>): the fully unrolled MFMA loop takes hipcc about half a
// minute per kernel, so the shapes are built in parallel.
#include "split_kernels.h"

#define TNMF_SPLIT_CAT2(a, b, c) a##b##_##c
#define TNMF_SPLIT_CAT(a, b, c) TNMF_SPLIT_CAT2(a, b, c)

int TNMF_SPLIT_CAT(split_launch_, TNMF_SPLIT_AY, TNMF_SPLIT_NR4)(tnmf_hip_ctx *ctx, const Geo &g, const float *V,
                                                                 const float *R, const float *W, float *H_inout,
                                                                 float *neg, float *pos, bool fused, float reg,
                                                                 hipStream_t s, const float *extra) {
    return launch<TNMF_SPLIT_AY, TNMF_SPLIT_NR4>(ctx, g, V, R, W, H_inout, neg, pos, fused, reg, s, extra);
}

int TNMF_SPLIT_CAT(split_prepare_, TNMF_SPLIT_AY, TNMF_SPLIT_NR4)() {
    return prepare_one<TNMF_SPLIT_AY, TNMF_SPLIT_NR4>();
}
