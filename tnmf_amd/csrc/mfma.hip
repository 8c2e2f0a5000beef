// mfma.hip -- f32 MFMA kernels (placeholder until the first generic path is parity-green on the GPU).
#include "mfma.h"

bool mfma_supported(const Geo &, int) { return false; }
int mfma_reconstruct(tnmf_hip_ctx *, const Geo &, const float *, const float *, float *, hipStream_t) {
    return TNMF_E_UNSUPPORTED;
}
int mfma_corr_W(tnmf_hip_ctx *, const Geo &, const float *, const float *, const float *, float *, float *, float *,
                bool, float, hipStream_t) {
    return TNMF_E_UNSUPPORTED;
}
int mfma_corr_H_chunks(const tnmf_hip_ctx *, const Geo &) { return 0; }
int mfma_corr_H(tnmf_hip_ctx *, const Geo &, const float *, const float *, const float *, double *, int,
                hipStream_t) {
    return TNMF_E_UNSUPPORTED;
}
