// api.hip -- the extern "C" surface of libtnmf_hip.so (include/tnmf_hip.h): argument checks, scratch management and
// dispatch between the kernel families.  No torch types, no exceptions, no allocation inside a call once
// tnmf_hip_ctx_reserve() has sized the scratch.
#include <cstdlib>
#include <cstring>
#include <new>
#include <algorithm>
#include <vector>

#include "fft.h"
#include "generic.h"
#include "mfma.h"
#include "split.h"
#include "volume.h"

namespace {

int to_geo(const tnmf_hip_geom *in, Geo *g) {
    if (!in) return TNMF_E_NULL;
    if (in->dtype != 0 && in->dtype != 1) return TNMF_E_DTYPE;
    if (in->ndim != 1 && in->ndim != 2) return TNMF_E_GEOM;
    g->N = in->N;
    g->M = in->M;
    g->C = in->C;
    if (in->ndim == 1) {
        g->Dy = 1;
        g->Dx = in->D[0];
        g->Ay = 1;
        g->Ax = in->A[0];
    } else {
        g->Dy = in->D[0];
        g->Dx = in->D[1];
        g->Ay = in->A[0];
        g->Ax = in->A[1];
    }
    if (g->N < 0 || g->M <= 0 || g->C <= 0 || g->Dy <= 0 || g->Dx <= 0 || g->Ay <= 0 || g->Ax <= 0)
        return TNMF_E_GEOM;
    g->Hy = g->Dy + g->Ay - 1;
    g->Hx = g->Dx + g->Ax - 1;
    g->Hs = in->h_row_stride > 0 ? in->h_row_stride : g->Hx;
    if (g->Hs < g->Hx) return TNMF_E_GEOM;
    return TNMF_OK;
}

inline size_t esize(int dtype) { return dtype == 0 ? 4 : 8; }

// three shift axes (volumes): a kernel family of its own (volume.hip); H is C-contiguous there
inline bool is_vol(const tnmf_hip_geom *in) { return in && in->ndim == 3; }

int to_vol(const tnmf_hip_geom *in, Vol *v) {
    if (!in) return TNMF_E_NULL;
    if (in->dtype != 0 && in->dtype != 1) return TNMF_E_DTYPE;
    v->N = in->N;
    v->M = in->M;
    v->C = in->C;
    if (v->N < 0 || v->M <= 0 || v->C <= 0) return TNMF_E_GEOM;
    for (int i = 0; i < 3; ++i) {
        v->D[i] = in->D[i];
        v->A[i] = in->A[i];
        if (v->D[i] <= 0 || v->A[i] <= 0) return TNMF_E_GEOM;
        v->H[i] = v->D[i] + v->A[i] - 1;
    }
    if (in->h_row_stride > 0 && in->h_row_stride != v->H[2]) return TNMF_E_STRIDE;
    return TNMF_OK;
}
inline size_t vol_vox(const Vol &v) { return (size_t)v.D[0] * v.D[1] * v.D[2]; }
inline size_t vol_hvox(const Vol &v) { return (size_t)v.H[0] * v.H[1] * v.H[2]; }
inline size_t vol_avox(const Vol &v) { return (size_t)v.A[0] * v.A[1] * v.A[2]; }
// the dictionary of a volume problem as the (M, C, flattened atom) rows the normalisation kernel works on
inline Geo vol_dict_geo(const Vol &v) {
    Geo g = {};
    g.N = v.N;
    g.M = v.M;
    g.C = v.C;
    g.Ay = v.A[0] * v.A[1];
    g.Ax = v.A[2];
    return g;
}

// scratch layout: [R (N*C*D elements)] [split-K partials (doubles)] [energy partials + result (doubles)]
struct Scratch {
    size_t r_off, r_bytes;
    size_t part_off, part_bytes;
    size_t red_off, red_bytes;
    size_t total;
    int P;
};

Scratch plan_scratch(const tnmf_hip_ctx *ctx, const Geo &g, int dtype) {
    Scratch s;
    s.r_off = 0;
    s.r_bytes = align_up((size_t)g.N * g.C * g.Dy * g.Dx * esize(dtype), 256);
    int P = generic_corr_H_chunks(ctx, g);
    const int Pm = mfma_corr_H_chunks(ctx, g);
    if (Pm > P) P = Pm;
    s.P = P;
    s.part_off = s.r_off + s.r_bytes;
    s.part_bytes = align_up((size_t)P * g.M * g.C * g.Ay * g.Ax * 2 * sizeof(double), 256);
    s.red_off = s.part_off + s.part_bytes;
    s.red_bytes = align_up((size_t)(kEnergyPartials + 8) * sizeof(double), 256);
    s.total = s.red_off + s.red_bytes;
    return s;
}

int ensure_buffer(void **buf, size_t *have, size_t bytes, bool slack);

int ensure_scratch(tnmf_hip_ctx *ctx, size_t bytes) { return ensure_buffer(&ctx->ws, &ctx->ws_bytes, bytes, true); }
int ensure_hwork(tnmf_hip_ctx *ctx, size_t bytes) { return ensure_buffer(&ctx->hw, &ctx->hw_bytes, bytes, false); }

int ensure_buffer(void **buf, size_t *have, size_t bytes, bool slack) {
    if (bytes <= *have) return TNMF_OK;
    // the larger buffer is allocated BEFORE the current one is released, so a failure leaves the working one in place;
    // when both do not fit side by side the old order (release, then allocate) is tried once
    const size_t want = align_up(bytes + (slack ? bytes / 8 : 0), 1 << 20);
    void *bigger = nullptr;
    if (hipMalloc(&bigger, want) != hipSuccess) {
        (void)hipGetLastError();
        bigger = nullptr;
        if (!*buf) return TNMF_E_WORKSPACE;
    }
    if (*buf) {
        TNMF_HIP_TRY(hipDeviceSynchronize());
        TNMF_HIP_TRY(hipFree(*buf));
        *buf = nullptr;
        *have = 0;
    }
    if (!bigger && hipMalloc(&bigger, want) != hipSuccess) {
        (void)hipGetLastError();
        return TNMF_E_WORKSPACE;
    }
    *buf = bigger;
    *have = want;
    return TNMF_OK;
}

inline char *ws_at(tnmf_hip_ctx *ctx, size_t off) { return static_cast<char *>(ctx->ws) + off; }

enum Prim { kReconstruct, kCorrW, kCorrH };

// FFT family for reconstruct and the W gradient (the hybrid dispatch): forced by TNMF_PATH_HYBRID wherever the family
// covers the shape; chosen by TNMF_PATH_AUTO for float32 problems that are not tiny (the family costs ~20 launches per iteration).
// Measured: W, H and the energy stay as close to the float64 oracle as with the direct kernels alone,
// because the H gradient -- the only place where float32 transform error matters -- stays on the direct kernels.
// A mini-batch slice decides by its OWN size: small batches of a large resident problem (the stochastic schedules with
// batch_size 3) take the direct kernels -- 6 launches per batch step instead of ~40 -- and find the resident problem's
// row-padded H readable there: the generic kernels and the split kernel take the row stride, only the f32 MFMA kernels
// want C-contiguous rows and are skipped for padded ones.
bool use_fft_hybrid(const tnmf_hip_ctx *ctx, const Geo &g, int dtype) {
    if (ctx->path == TNMF_PATH_HYBRID) return fft_has(g, dtype);
    if (ctx->path != TNMF_PATH_AUTO || dtype != 0 || !fft_has(g, dtype)) return false;
    // measured crossover against the direct kernels at 128x128 samples, 16 atoms: one sample (2^18 entries) is a tie,
    // two are 20 % ahead, 64 samples (config 2) 1.9x
    return (size_t)g.N * g.M * g.Hy * g.Hx >= ((size_t)1 << 19);
}

// H gradient on the bf16 matrix cores with exact 3 x bf16 operand splits: forced by TNMF_PATH_SPLIT, default under AUTO
// and HYBRID (tnmf_hip_ctx_set_split), never under MFMA (the exact-f32 family), GENERIC or FFT.
bool use_split(const tnmf_hip_ctx *ctx, const Geo &g, int dtype) {
    if (ctx->path == TNMF_PATH_SPLIT) return split_has_corr_W(g, dtype);
    if (ctx->path != TNMF_PATH_AUTO && ctx->path != TNMF_PATH_HYBRID) return false;
    // (tiny calls -- the batches of the stochastic schedules -- are launch latency: the split kernel needs an operand
    // preparation launch in front of it, the generic kernel does not)
    if (ctx->path == TNMF_PATH_AUTO && (size_t)g.N * g.M * g.Hy * g.Hx < ((size_t)1 << 16)) return false;
    return ctx->split && split_has_corr_W(g, dtype, true);
}

bool use_mfma(const tnmf_hip_ctx *ctx, const Geo &g, int dtype, Prim prim) {
    if (ctx->path == TNMF_PATH_GENERIC || ctx->path == TNMF_PATH_FFT) return false;   // (FFT never gets here)
    switch (prim) {
        case kReconstruct: return mfma_has_reconstruct(g, dtype);
        case kCorrW: return mfma_has_corr_W(g, dtype);
        default: return mfma_has_corr_H(g, dtype);
    }
}

#define ENTER(ctx, geom)                                   \
    if (!(ctx)) return TNMF_E_NULL;                        \
    Geo g;                                                 \
    {                                                      \
        const int _rc = to_geo((geom), &g);                \
        if (_rc != TNMF_OK) return _rc;                    \
    }                                                      \
    const int dtype = (geom)->dtype;                       \
    hipStream_t s = static_cast<hipStream_t>(stream);      \
    TNMF_HIP_TRY(hipSetDevice((ctx)->device));

#define CHECK(rc_expr)                  \
    do {                                \
        const int _rc = (rc_expr);      \
        if (_rc != TNMF_OK) return _rc; \
    } while (0)

#define VOL_ENTER(ctx, geom)                               \
    if (!(ctx)) return TNMF_E_NULL;                        \
    Vol v;                                                 \
    {                                                      \
        const int _rc = to_vol((geom), &v);                \
        if (_rc != TNMF_OK) return _rc;                    \
    }                                                      \
    const int dtype = (geom)->dtype;                       \
    hipStream_t s = static_cast<hipStream_t>(stream);      \
    TNMF_HIP_TRY(hipSetDevice((ctx)->device));             \
    (ctx)->last_path = "volume";

// scratch of a volume call: [R of the slice | energy partials + result | partial sums of the W gradient]
int vol_scratch(tnmf_hip_ctx *ctx, const Vol &v, int dtype, void **R, double **red, double **part = nullptr,
                int *P = nullptr, size_t extra = 0) {
    const size_t r_bytes = align_up((size_t)v.N * v.C * vol_vox(v) * esize(dtype), 256);
    const size_t e_bytes = align_up((size_t)(kEnergyPartials + 8) * sizeof(double), 256);
    const int chunks = vol_corr_H_chunks(ctx, v);
    const size_t p_bytes = align_up((size_t)chunks * v.M * v.C * vol_avox(v) * 2 * sizeof(double), 256);
    const int rc = ensure_scratch(ctx, r_bytes + e_bytes + p_bytes + extra);   // (extra: behind the partial sums)
    if (rc != TNMF_OK) return rc;
    if (R) *R = ws_at(ctx, 0);
    if (red) *red = reinterpret_cast<double *>(ws_at(ctx, r_bytes));
    if (part) *part = reinterpret_cast<double *>(ws_at(ctx, r_bytes + e_bytes));
    if (P) *P = chunks;
    return TNMF_OK;
}

int do_reconstruct(tnmf_hip_ctx *ctx, const Geo &g, int dtype, const void *W, const void *H, void *R, hipStream_t s) {
    if (g.N == 0) return TNMF_OK;
    // the FFT family serves non-negative factorisations: outputs that are non-negative by construction are clamped
    if (ctx->path == TNMF_PATH_FFT || use_fft_hybrid(ctx, g, dtype)) {
        const int rc = fft_reconstruct(ctx, g, dtype, W, H, R, true, s);
        // AUTO only chose the family for speed: when its workspace does not fit, the direct kernels still do the job
        if (!(rc == TNMF_E_WORKSPACE && ctx->path == TNMF_PATH_AUTO)) return rc;
    }
    // (the f32 MFMA kernels read C-contiguous H; the generic kernels take the row stride)
    if (g.Hs == g.Hx && use_mfma(ctx, g, dtype, kReconstruct)) {
        ctx->last_path = "mfma";
        return mfma_reconstruct(ctx, g, (const float *)W, (const float *)H, (float *)R, s);
    }
    if (ctx->path == TNMF_PATH_MFMA) return g.Hs != g.Hx ? TNMF_E_STRIDE : TNMF_E_UNSUPPORTED;
    ctx->last_path = "generic";
    return generic_reconstruct(ctx, g, dtype, W, H, R, s);
}

// extra (fused only, may be NULL): a further term of the denominator, laid out like H (the lateral inhibition terms).
// Kernel families without an epilogue for it answer TNMF_E_UNSUPPORTED / TNMF_E_STRIDE before touching anything; the
// caller then takes the unfused gradient and launch_mu_update_extra.
int do_corr_W(tnmf_hip_ctx *ctx, const Geo &g, int dtype, const void *V, const void *R, const void *W, void *Hio,
              void *neg, void *pos, bool fused, double reg, hipStream_t s, const void *extra = nullptr) {
    if (g.N == 0) return TNMF_OK;
    if (extra && (!fused || ctx->path == TNMF_PATH_FFT)) return TNMF_E_UNSUPPORTED;
    // (AUTO never takes the FFT family here, although it is the faster one for heavy atoms -- 46 vs 53 ms per iteration
    // at config 5's shard, C*Ay*Ax = 768: its float32 transform noise leaves 4e-4 of max|H| in the activations, and the
    // parity bar of the default path is 1e-5 on W AND H.  path = FFT is the opt-in.)
    if (ctx->path == TNMF_PATH_FFT)
        return fused ? fft_update_H(ctx, g, dtype, V, R, W, Hio, reg, s) : fft_grad_H(ctx, g, dtype, V, R, W, neg, pos, s);
    if (fused) fft_invalidate_H(ctx, g, dtype, Hio);   // the direct kernels are about to change H: its cached spectra are stale
    if (use_split(ctx, g, dtype)) {
        const int rc = split_corr_W(ctx, g, (const float *)V, (const float *)R, (const float *)W, (float *)Hio,
                                    (float *)neg, (float *)pos, fused, (float)reg, s, (const float *)extra);
        if (rc == TNMF_OK) ctx->last_path = "split";
        if (rc != TNMF_E_UNSUPPORTED) return rc;
    }
    if ((!fused || g.Hs == g.Hx) && use_mfma(ctx, g, dtype, kCorrW)) {
        if (extra) return TNMF_E_UNSUPPORTED;   // (the f32 MFMA kernel has no extra-term epilogue)
        ctx->last_path = "mfma";
        return mfma_corr_W(ctx, g, (const float *)V, (const float *)R, (const float *)W, (float *)Hio, (float *)neg,
                           (float *)pos, fused, (float)reg, s);
    }
    if (ctx->path == TNMF_PATH_MFMA || ctx->path == TNMF_PATH_SPLIT)
        return fused && g.Hs != g.Hx ? TNMF_E_STRIDE : TNMF_E_UNSUPPORTED;   // (nothing has been written)
    ctx->last_path = "generic";
    return generic_corr_W(ctx, g, dtype, V, R, W, Hio, neg, pos, fused, reg, s, extra);
}

// the split-K kernel of the direct families alone: partials[P][M*C][Ay*Ax][2] (doubles), P returned.  g.N > 0.
int do_corr_H_partials(tnmf_hip_ctx *ctx, const Geo &g, int dtype, const void *V, const void *R, const void *H,
                       double *partials, int *P_out, hipStream_t s) {
    if (g.Hs == g.Hx && use_mfma(ctx, g, dtype, kCorrH)) {
        ctx->last_path = "mfma";
        *P_out = mfma_corr_H_chunks(ctx, g);
        return mfma_corr_H(ctx, g, (const float *)V, (const float *)R, (const float *)H, partials, *P_out, s);
    }
    if (ctx->path == TNMF_PATH_MFMA) return g.Hs != g.Hx ? TNMF_E_STRIDE : TNMF_E_UNSUPPORTED;
    ctx->last_path = "generic";
    *P_out = generic_corr_H_chunks(ctx, g);
    return generic_corr_H(ctx, g, dtype, V, R, H, partials, *P_out, s);
}

bool corr_H_on_fft(const tnmf_hip_ctx *ctx, const Geo &g, int dtype) {
    return ctx->path == TNMF_PATH_FFT || use_fft_hybrid(ctx, g, dtype);
}

int do_corr_H(tnmf_hip_ctx *ctx, const Geo &g, int dtype, const Scratch &sc, const void *V, const void *R,
              const void *H, void *neg, void *pos, hipStream_t s) {
    double *partials = reinterpret_cast<double *>(ws_at(ctx, sc.part_off));
    int P;
    if (g.N == 0) {
        // empty slice: the sums are zero
        TNMF_HIP_TRY(hipMemsetAsync(neg, 0, (size_t)g.M * g.C * g.Ay * g.Ax * esize(dtype), s));
        TNMF_HIP_TRY(hipMemsetAsync(pos, 0, (size_t)g.M * g.C * g.Ay * g.Ax * esize(dtype), s));
        return TNMF_OK;
    }
    if (corr_H_on_fft(ctx, g, dtype)) {
        const int rc = fft_grad_W(ctx, g, dtype, V, R, H, neg, pos, true, s);
        if (!(rc == TNMF_E_WORKSPACE && ctx->path == TNMF_PATH_AUTO)) return rc;
    }
    CHECK(do_corr_H_partials(ctx, g, dtype, V, R, H, partials, &P, s));
    return finalize_corr_H(g, dtype, partials, P, neg, pos, s);
}


// ---- three shift axes: the entry points below hand over to these (same argument meaning, C-contiguous H)
int vol_api_reconstruct(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom, const void *W, const void *H, void *R, void *stream) {
    VOL_ENTER(ctx, geom);
    if (!W || (v.N > 0 && (!H || !R))) return TNMF_E_NULL;
    return vol_reconstruct(v, dtype, W, H, R, s);
}

int vol_api_grad_H(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom, const void *V, const void *R_or_null, const void *W,
                   const void *H, void *neg, void *pos, void *stream) {
    VOL_ENTER(ctx, geom);
    if (!W || (v.N > 0 && (!V || !neg || !pos))) return TNMF_E_NULL;
    const void *R = R_or_null;
    if (!R && v.N > 0) {
        if (!H) return TNMF_E_NULL;
        void *Rs;
        CHECK(vol_scratch(ctx, v, dtype, &Rs, nullptr));
        CHECK(vol_reconstruct(v, dtype, W, H, Rs, s));
        R = Rs;
    }
    return vol_corr_W(v, dtype, V, R, W, nullptr, neg, pos, false, 0.0, s);
}

int vol_api_grad_W(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom, const void *V, const void *R_scratch, int r_is_valid,
                   const void *W, const void *H, void *neg, void *pos, void *stream) {
    VOL_ENTER(ctx, geom);
    if (!neg || !pos || (v.N > 0 && (!V || !H))) return TNMF_E_NULL;
    const void *R = R_scratch;
    void *Rws;
    double *part;
    int P;
    CHECK(vol_scratch(ctx, v, dtype, &Rws, nullptr, &part, &P));
    if (v.N > 0 && !r_is_valid) {
        if (!W) return TNMF_E_NULL;
        void *Rs = R_scratch ? const_cast<void *>(R_scratch) : Rws;
        CHECK(vol_reconstruct(v, dtype, W, H, Rs, s));
        R = Rs;
    }
    return vol_corr_H(v, dtype, V, R, H, neg, pos, part, P, s);
}

int vol_api_energy(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom, const void *V, const void *W, const void *H,
                   double *out_host, void *stream) {
    VOL_ENTER(ctx, geom);
    if (!out_host) return TNMF_E_NULL;
    *out_host = 0.0;
    if (v.N == 0) return TNMF_OK;
    if (!V || !W || !H) return TNMF_E_NULL;
    void *Rs;
    double *red;
    CHECK(vol_scratch(ctx, v, dtype, &Rs, &red));
    CHECK(vol_reconstruct(v, dtype, W, H, Rs, s));
    CHECK(launch_half_sqdiff(ctx, dtype, V, Rs, (size_t)v.N * v.C * vol_vox(v), red, red + kEnergyPartials, s));
    TNMF_HIP_TRY(hipMemcpyAsync(out_host, red + kEnergyPartials, sizeof(double), hipMemcpyDeviceToHost, s));
    TNMF_HIP_TRY(hipStreamSynchronize(s));
    return TNMF_OK;
}

int vol_api_update_H(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom, const void *V, const void *W, void *H_inout,
                     void *R_scratch, int r_is_valid, double eps, double sparsity, void *stream) {
    VOL_ENTER(ctx, geom);
    if (v.N == 0) return TNMF_OK;
    if (!V || !W || !H_inout) return TNMF_E_NULL;
    void *Rs = R_scratch;
    if (!Rs) {
        if (r_is_valid) return TNMF_E_NULL;
        CHECK(vol_scratch(ctx, v, dtype, &Rs, nullptr));
    }
    if (!r_is_valid) CHECK(vol_reconstruct(v, dtype, W, H_inout, Rs, s));
    const double reg = eps + (sparsity > 0 ? sparsity : 0.0);   // TransformInvariantNMF.py:227-230
    return vol_corr_W(v, dtype, V, Rs, W, H_inout, nullptr, nullptr, true, reg, s);
}

// TransformInvariantNMF._update_H in full for volumes: the lateral terms (three passes of the 1-D convolution, then
// k_vol_lateral) and the padded modes (pad, 'valid' kernels, fold) on work arrays of the library, one update kernel
int vol_api_update_H_ex(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom, int mode, const void *V, const void *W,
                        void *H_inout, void *R_scratch, double eps, double sparsity, double inhibition,
                        double cross_inhibition, const double *const kern[3], const int klen[3], void *stream) {
    VOL_ENTER(ctx, geom);
    if (mode < TNMF_MODE_VALID || mode > TNMF_MODE_REFLECT) return TNMF_E_UNSUPPORTED;
    if (v.N == 0) return TNMF_OK;
    if (!V || !W || !H_inout) return TNMF_E_NULL;
    if (inhibition < 0 || cross_inhibition < 0) return TNMF_E_GEOM;
    const bool lateral = inhibition > 0 || cross_inhibition > 0;
    if (lateral)
        for (int i = 0; i < 3; ++i) {
            if (!kern[i]) return TNMF_E_NULL;
            if (klen[i] < 1 || klen[i] > kMaxTaps || !(klen[i] & 1)) return TNMF_E_UNSUPPORTED;
        }
    const double reg = eps + (sparsity > 0 ? sparsity : 0.0);   // TransformInvariantNMF.py:227-230
    const double xc = cross_inhibition > 0 && v.M > 1 ? cross_inhibition / (v.M - 1) : 0.0;   // (:266-268)
    if (mode == TNMF_MODE_VALID && !lateral)
        return vol_api_update_H(ctx, geom, V, W, H_inout, R_scratch, 0, eps, sparsity, stream);
    int S[3];
    for (int i = 0; i < 3; ++i) {
        S[i] = mode == TNMF_MODE_VALID ? v.H[i] : (mode == TNMF_MODE_FULL ? v.D[i] - v.A[i] + 1 : v.D[i]);
        if (S[i] < 1) return TNMF_E_GEOM;
    }
    const size_t es = esize(dtype), planes = (size_t)v.N * v.M;
    const size_t svox = (size_t)S[0] * S[1] * S[2];
    const size_t nS = align_up(planes * svox * es, 256), nP = align_up(planes * vol_hvox(v) * es, 256);
    // work arrays: [G0 | G1] (lateral), [neg | pos] in the mode's shape, and for the padded modes [Hp | negp | posp]
    const size_t lat = lateral ? 2 * nS : 0, pad = mode == TNMF_MODE_VALID ? 0 : 3 * nP;
    CHECK(ensure_hwork(ctx, lat + 2 * nS + pad));
    char *G0 = static_cast<char *>(ctx->hw), *G1 = G0 + nS;
    char *neg = static_cast<char *>(ctx->hw) + lat, *pos = neg + nS;
    char *Hp = pos + nS, *negp = Hp + nP, *posp = negp + nP;
    void *Rs = R_scratch;
    if (!Rs) CHECK(vol_scratch(ctx, v, dtype, &Rs, nullptr));
    const void *E = nullptr;
    if (lateral) {
        // first shift axis first (_NumPyBackend.py:60-62): H -> G0 -> G1 -> G0
        CHECK(launch_convolve_axis(ctx, dtype, H_inout, G0, planes, S[0], S[1] * S[2], kern[0], klen[0], s));
        CHECK(launch_convolve_axis(ctx, dtype, G0, G1, planes * S[0], S[1], S[2], kern[1], klen[1], s));
        CHECK(launch_convolve_axis(ctx, dtype, G1, G0, planes * S[0] * S[1], S[2], 1, kern[2], klen[2], s));
        CHECK(vol_lateral(ctx, dtype, (size_t)v.N, v.M, svox, G0, H_inout, inhibition, xc, s));
        E = G0;
    }
    if (mode == TNMF_MODE_VALID) {
        CHECK(vol_reconstruct(v, dtype, W, H_inout, Rs, s));
        CHECK(vol_corr_W(v, dtype, V, Rs, W, nullptr, neg, pos, false, 0.0, s));
    } else {
        CHECK(vol_pad_fold(ctx, v, dtype, mode, false, H_inout, Hp, s));
        CHECK(vol_reconstruct(v, dtype, W, Hp, Rs, s));
        CHECK(vol_corr_W(v, dtype, V, Rs, W, nullptr, negp, posp, false, 0.0, s));
        CHECK(vol_pad_fold(ctx, v, dtype, mode, true, negp, neg, s));
        CHECK(vol_pad_fold(ctx, v, dtype, mode, true, posp, pos, s));
    }
    // H <- H * neg / (pos + E + reg): every array is C-contiguous in the mode's shift shape (rows of S[2] elements)
    return launch_mu_update_extra(ctx, dtype, H_inout, neg, pos, E, planes * S[0] * S[1], S[2], S[2], reg, s);
}

// a list of operations on slices of the resident volume problem (tnmf_hip_run_schedule): one host call, one launch chain
int vol_api_run_schedule(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom, const void *V, void *W_inout, void *H_inout,
                         void *R_scratch, void *acc, const tnmf_hip_op *ops, int n_ops, double eps, double sparsity,
                         void *stream) {
    VOL_ENTER(ctx, geom);
    if (n_ops < 0 || (n_ops > 0 && !ops)) return TNMF_E_NULL;
    if (!V || !W_inout || !H_inout || !acc) return TNMF_E_NULL;
    const size_t es = esize(dtype);
    const size_t vs = (size_t)v.C * vol_vox(v) * es, hs = (size_t)v.M * vol_hvox(v) * es;
    const size_t wn = (size_t)v.M * v.C * vol_avox(v);
    int nmax = 0;
    for (int i = 0; i < n_ops; ++i) {
        if (ops[i].n0 < 0 || ops[i].n1 < ops[i].n0 || ops[i].n1 > v.N) return TNMF_E_GEOM;
        if (ops[i].n1 - ops[i].n0 > nmax) nmax = ops[i].n1 - ops[i].n0;
    }
    const double reg = eps + (sparsity > 0 ? sparsity : 0.0);   // TransformInvariantNMF.py:227-230
    // scratch for the largest slice: [R | energy words | partial sums of the W gradient], and one gradient behind it
    Vol vmax = v;
    vmax.N = nmax;
    void *Rws;
    double *part;
    int Pmax;
    CHECK(vol_scratch(ctx, vmax, dtype, &Rws, nullptr, &part, &Pmax, align_up(2 * wn * es, 256)));
    char *grad = reinterpret_cast<char *>(part) + align_up((size_t)Pmax * wn * 2 * sizeof(double), 256);
    const Geo dict = vol_dict_geo(v);
    for (int i = 0; i < n_ops; ++i) {
        const tnmf_hip_op &op = ops[i];
        Vol vs_ = v;
        vs_.N = op.n1 - op.n0;
        const char *Vb = static_cast<const char *>(V) + (size_t)op.n0 * vs;
        char *Hb = static_cast<char *>(H_inout) + (size_t)op.n0 * hs;
        void *Rb = R_scratch ? static_cast<void *>(static_cast<char *>(R_scratch) + (size_t)op.n0 * vs) : Rws;
        switch (op.kind) {
            case TNMF_OP_UPDATE_H:
                if (vs_.N == 0) break;
                CHECK(vol_reconstruct(vs_, dtype, W_inout, Hb, Rb, s));
                CHECK(vol_corr_W(vs_, dtype, Vb, Rb, W_inout, Hb, nullptr, nullptr, true, reg, s));
                break;
            case TNMF_OP_GRAD_W: {
                if (vs_.N > 0) CHECK(vol_reconstruct(vs_, dtype, W_inout, Hb, Rb, s));
                int P = vol_corr_H_chunks(ctx, vs_);
                if (P > Pmax) P = Pmax;   // (the chunk count grows with the slice: never beyond the largest one's)
                CHECK(vol_corr_H(vs_, dtype, Vb, Rb, Hb, grad, grad + wn * es, part, P, s));
                CHECK(launch_axpby(ctx, dtype, acc, grad, op.a, op.b, 2 * wn, s));
                break;
            }
            case TNMF_OP_APPLY_W: {
                char *np = static_cast<char *>(acc);
                CHECK(launch_apply_normalize_W(dict, dtype, W_inout, np, np + wn * es, eps, true, s));
                break;
            }
            default: return TNMF_E_UNSUPPORTED;
        }
    }
    return TNMF_OK;
}

int vol_api_pad_fold(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom, int mode, bool fold, const void *in, void *out,
                     void *stream) {
    VOL_ENTER(ctx, geom);
    if (mode < TNMF_MODE_VALID || mode > TNMF_MODE_REFLECT) return TNMF_E_UNSUPPORTED;
    if (v.N > 0 && (!in || !out)) return TNMF_E_NULL;
    return vol_pad_fold(ctx, v, dtype, mode, fold, in, out, s);
}

}  // namespace

extern "C" {

int tnmf_hip_abi_version(void) { return TNMF_HIP_ABI_VERSION; }

int tnmf_hip_ctx_h_row_stride(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom, int *stride_out) {
    if (!ctx || !stride_out) return TNMF_E_NULL;
    if (is_vol(geom)) {   // volumes: C-contiguous
        Vol v;
        const int vrc = to_vol(geom, &v);
        if (vrc != TNMF_OK) return vrc;
        *stride_out = v.H[2];
        return TNMF_OK;
    }
    Geo g;
    const int rc = to_geo(geom, &g);
    if (rc != TNMF_OK) return rc;
    const int dtype = geom->dtype;
    *stride_out = g.Hx;
    // padded rows pay where the split kernel reads and writes H in 32-pixel tiles and every other reader of H is the
    // FFT family's row transform; a whole number of 128-byte lines per row
    const bool hybrid = ctx->path == TNMF_PATH_AUTO || ctx->path == TNMF_PATH_HYBRID;
    if (hybrid && g.N > 0 && use_fft_hybrid(ctx, g, dtype) && use_split(ctx, g, dtype))
        *stride_out = (int)align_up((size_t)g.Hx, 32);
    return TNMF_OK;
}

const char *tnmf_hip_strerror(int code) {
    switch (code) {
        case TNMF_OK: return "ok";
        case TNMF_E_NULL: return "tnmf_hip: required pointer is NULL";
        case TNMF_E_GEOM: return "tnmf_hip: bad geometry";
        case TNMF_E_DTYPE: return "tnmf_hip: dtype must be 0 (f32) or 1 (f64)";
        case TNMF_E_WORKSPACE: return "tnmf_hip: scratch allocation failed";
        case TNMF_E_UNSUPPORTED: return "tnmf_hip: shape not supported by the selected kernel family";
        case TNMF_E_STRIDE: return "tnmf_hip: this kernel family wants C-contiguous H (h_row_stride == shift width)";
        default: break;
    }
    if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
    return "tnmf_hip: unknown error";
}

int tnmf_hip_ctx_create(int device_id, tnmf_hip_ctx **out) {
    if (!out) return TNMF_E_NULL;
    *out = nullptr;
    TNMF_HIP_TRY(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    TNMF_HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    {
        int rc = mfma_prepare_device();   // per device, hence per context
        if (rc == TNMF_OK) rc = split_prepare_device();
        if (rc != TNMF_OK) return rc;
    }
    tnmf_hip_ctx *ctx = new (std::nothrow) tnmf_hip_ctx();
    if (!ctx) return TNMF_E_WORKSPACE;
    ctx->device = device_id;
    ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    ctx->path = TNMF_PATH_AUTO;
    ctx->last_path = "none";
    {
        const char *ab = tnmf_diag_env("TNMF_HIP_ABLATE");   // -DTNMF_DIAG builds only; nullptr in the product
        ctx->ablate = ab ? atoi(ab) : 0;
    }
    ctx->ws = nullptr;
    ctx->ws_bytes = 0;
    ctx->split = 1;
    ctx->wimg = nullptr;
    ctx->wimg_bytes = 0;
    ctx->fft = FftState();
    *out = ctx;
    return TNMF_OK;
}

int tnmf_hip_ctx_destroy(tnmf_hip_ctx *ctx) {
    if (!ctx) return TNMF_OK;
    int rc = TNMF_OK;
    if (ctx->ws || ctx->fft.ws || ctx->wimg || ctx->hw) {
        (void)hipSetDevice(ctx->device);
        (void)hipDeviceSynchronize();
    }
    fft_release(ctx);
    split_release(ctx);
    if (ctx->hw) (void)hipFree(ctx->hw);
    for (int i = 0; i < tnmf_hip_ctx::kOpSlots; ++i) {
        if (ctx->ops_done[i]) (void)hipEventDestroy(ctx->ops_done[i]);
        if (ctx->ops_pinned[i]) (void)hipHostFree(ctx->ops_pinned[i]);
    }
    if (ctx->ws) {
        const hipError_t e = hipFree(ctx->ws);
        if (e != hipSuccess) rc = (int)e;
    }
    delete ctx;
    return rc;
}

int tnmf_hip_ctx_reserve(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom) {
    void *stream = nullptr;
    if (is_vol(geom)) {
        VOL_ENTER(ctx, geom);
        (void)s;
        return vol_scratch(ctx, v, dtype, nullptr, nullptr);
    }
    ENTER(ctx, geom);
    (void)s;
    if (g.N > 0 && ctx->path == TNMF_PATH_FFT) CHECK(fft_reserve(ctx, g, dtype, true));
    if (g.N > 0 && use_fft_hybrid(ctx, g, dtype)) {
        const int rc = fft_reserve(ctx, g, dtype, false);
        if (rc != TNMF_OK && !(rc == TNMF_E_WORKSPACE && ctx->path == TNMF_PATH_AUTO)) return rc;
    }
    return ensure_scratch(ctx, plan_scratch(ctx, g, dtype).total);
}

int tnmf_hip_ctx_set_path(tnmf_hip_ctx *ctx, int path) {
    if (!ctx) return TNMF_E_NULL;
    if (path < TNMF_PATH_AUTO || path > TNMF_PATH_SPLIT) return TNMF_E_UNSUPPORTED;
    if (path != ctx->path) fft_invalidate(ctx);   // (the FFT family's workspace layout follows the path: column lengths)
    ctx->path = path;
    return TNMF_OK;
}

int tnmf_hip_ctx_set_split(tnmf_hip_ctx *ctx, int enable) {
    if (!ctx) return TNMF_E_NULL;
    ctx->split = enable != 0;
    return TNMF_OK;
}

int tnmf_hip_ctx_set_persistent(tnmf_hip_ctx *ctx, int mode) {
    if (!ctx) return TNMF_E_NULL;
    if (mode < 0 || mode > 2) return TNMF_E_UNSUPPORTED;
    ctx->persistent = mode;
    return TNMF_OK;
}

int tnmf_hip_ctx_last_schedule_persistent(const tnmf_hip_ctx *ctx) { return ctx && ctx->last_schedule_persistent ? 1 : 0; }

int tnmf_hip_ctx_set_cache(tnmf_hip_ctx *ctx, int enable) {
    if (!ctx) return TNMF_E_NULL;
    ctx->fft.cache_enabled = enable != 0;
    fft_invalidate(ctx);
    return TNMF_OK;
}

int tnmf_hip_ctx_bind(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom, const void *H, const void *V) {
    if (!ctx) return TNMF_E_NULL;
    if (!geom || !H) {
        fft_unbind(ctx);
        return TNMF_OK;
    }
    if (is_vol(geom)) {   // (the spectrum cache belongs to the FFT family: nothing to bind for volumes)
        fft_unbind(ctx);
        return TNMF_OK;
    }
    Geo g;
    const int rc = to_geo(geom, &g);
    if (rc != TNMF_OK) return rc;
    fft_bind(ctx, g, geom->dtype, H, V);
    return TNMF_OK;
}

int tnmf_hip_ctx_cache_counters(const tnmf_hip_ctx *ctx, unsigned long long out[4]) {
    if (!ctx || !out) return TNMF_E_NULL;
    out[0] = ctx->fft.h_runs;
    out[1] = ctx->fft.h_hits;
    out[2] = ctx->fft.v_runs;
    out[3] = ctx->fft.v_hits;
    return TNMF_OK;
}

int tnmf_hip_ctx_invalidate(tnmf_hip_ctx *ctx) {
    if (!ctx) return TNMF_E_NULL;
    fft_invalidate(ctx);
    return TNMF_OK;
}

const char *tnmf_hip_ctx_last_path(const tnmf_hip_ctx *ctx) { return ctx ? ctx->last_path : "none"; }

#ifdef TNMF_DIAG
// diagnostic library only (not part of the ABI of include/tnmf_hip.h): change the ablation mask of a live context
int tnmf_hip_diag_set_ablate(tnmf_hip_ctx *ctx, int mask) {
    if (!ctx) return TNMF_E_NULL;
    ctx->ablate = mask;
    return TNMF_OK;
}
#endif

int tnmf_hip_reconstruct(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom, const void *W, const void *H, void *R,
                         void *stream) {
    if (is_vol(geom)) return vol_api_reconstruct(ctx, geom, W, H, R, stream);
    ENTER(ctx, geom);
    if (!W || (g.N > 0 && (!H || !R))) return TNMF_E_NULL;
    return do_reconstruct(ctx, g, dtype, W, H, R, s);
}

int tnmf_hip_grad_H(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom, const void *V, const void *R_or_null,
                    const void *W, const void *H, void *neg, void *pos, void *stream) {
    if (is_vol(geom)) return vol_api_grad_H(ctx, geom, V, R_or_null, W, H, neg, pos, stream);
    ENTER(ctx, geom);
    if (!W || (g.N > 0 && (!V || !neg || !pos))) return TNMF_E_NULL;
    const void *R = R_or_null;
    if (!R && g.N > 0) {
        if (!H) return TNMF_E_NULL;
        const Scratch sc = plan_scratch(ctx, g, dtype);
        CHECK(ensure_scratch(ctx, sc.total));
        void *Rs = ws_at(ctx, sc.r_off);
        CHECK(do_reconstruct(ctx, g, dtype, W, H, Rs, s));
        R = Rs;
    }
    return do_corr_W(ctx, g, dtype, V, R, W, nullptr, neg, pos, false, 0.0, s);
}

int tnmf_hip_grad_W(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom, const void *V, const void *R_or_null,
                    const void *W, const void *H, void *neg, void *pos, void *stream) {
    if (is_vol(geom)) return vol_api_grad_W(ctx, geom, V, R_or_null, R_or_null != nullptr, W, H, neg, pos, stream);
    ENTER(ctx, geom);
    if (!neg || !pos || (g.N > 0 && (!V || !H))) return TNMF_E_NULL;
    const Scratch sc = plan_scratch(ctx, g, dtype);
    CHECK(ensure_scratch(ctx, sc.total));
    const void *R = R_or_null;
    if (!R && g.N > 0) {
        if (!W) return TNMF_E_NULL;
        void *Rs = ws_at(ctx, sc.r_off);
        CHECK(do_reconstruct(ctx, g, dtype, W, H, Rs, s));
        R = Rs;
    }
    return do_corr_H(ctx, g, dtype, sc, V, R, H, neg, pos, s);
}

int tnmf_hip_mu_update(tnmf_hip_ctx *ctx, int dtype, void *arr, const void *neg, void *pos, double reg,
                       size_t n_elems, void *stream) {
    if (!ctx) return TNMF_E_NULL;
    if (dtype != 0 && dtype != 1) return TNMF_E_DTYPE;
    if (n_elems > 0 && (!arr || !neg || !pos)) return TNMF_E_NULL;
    TNMF_HIP_TRY(hipSetDevice(ctx->device));
    fft_invalidate(ctx);   // arr may be (part of) the activations whose row spectra are cached
    return launch_mu_update(ctx, dtype, arr, neg, pos, reg, n_elems, static_cast<hipStream_t>(stream));
}

int tnmf_hip_normalize_W(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom, void *W, void *stream) {
    if (is_vol(geom)) {
        VOL_ENTER(ctx, geom);
        if (!W) return TNMF_E_NULL;
        return launch_apply_normalize_W(vol_dict_geo(v), dtype, W, nullptr, nullptr, 0.0, false, s);
    }
    ENTER(ctx, geom);
    if (!W) return TNMF_E_NULL;
    fft_invalidate_W(ctx);
    return launch_apply_normalize_W(g, dtype, W, nullptr, nullptr, 0.0, false, s);
}

int tnmf_hip_energy(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom, const void *V, const void *W, const void *H,
                    double *out_host, void *stream) {
    if (is_vol(geom)) return vol_api_energy(ctx, geom, V, W, H, out_host, stream);
    ENTER(ctx, geom);
    if (!out_host) return TNMF_E_NULL;
    if (g.N == 0) {
        *out_host = 0.0;
        return TNMF_OK;
    }
    if (!V || !W || !H) return TNMF_E_NULL;
    const Scratch sc = plan_scratch(ctx, g, dtype);
    CHECK(ensure_scratch(ctx, sc.total));
    void *Rs = ws_at(ctx, sc.r_off);
    CHECK(do_reconstruct(ctx, g, dtype, W, H, Rs, s));
    double *red = reinterpret_cast<double *>(ws_at(ctx, sc.red_off));
    CHECK(launch_half_sqdiff(ctx, dtype, V, Rs, (size_t)g.N * g.C * g.Dy * g.Dx, red, red + kEnergyPartials, s));
    TNMF_HIP_TRY(hipMemcpyAsync(out_host, red + kEnergyPartials, sizeof(double), hipMemcpyDeviceToHost, s));
    TNMF_HIP_TRY(hipStreamSynchronize(s));
    return TNMF_OK;
}

int tnmf_hip_convolve_multi_1d(tnmf_hip_ctx *ctx, int dtype, int ndim, size_t rows, const int *shape,
                               const void *in, void *out, void *tmp, const double *kernel0, int len0,
                               const double *kernel1, int len1, void *stream) {
    if (!ctx || !shape || !kernel0) return TNMF_E_NULL;
    if (dtype != 0 && dtype != 1) return TNMF_E_DTYPE;
    if (ndim != 1 && ndim != 2) return TNMF_E_GEOM;
    hipStream_t s = static_cast<hipStream_t>(stream);
    TNMF_HIP_TRY(hipSetDevice(ctx->device));
    if (rows == 0) return TNMF_OK;
    if (!in || !out) return TNMF_E_NULL;
    if (ndim == 1) return launch_convolve_axis(ctx, dtype, in, out, rows, shape[0], 1, kernel0, len0, s);
    if (!tmp || !kernel1) return TNMF_E_NULL;
    // reference order: axis -2 first, then axis -1 (TransformInvariantNMF.py:254; _NumPyBackend.py:60-62)
    CHECK(launch_convolve_axis(ctx, dtype, in, tmp, rows, shape[0], shape[1], kernel0, len0, s));
    return launch_convolve_axis(ctx, dtype, tmp, out, rows * (size_t)shape[0], shape[1], 1, kernel1, len1, s);
}

int tnmf_hip_convolve_axis(tnmf_hip_ctx *ctx, int dtype, size_t rows, int len, size_t inner, const void *in, void *out,
                           const double *kernel, int klen, void *stream) {
    if (!ctx || !kernel) return TNMF_E_NULL;
    if (dtype != 0 && dtype != 1) return TNMF_E_DTYPE;
    if (len < 0 || inner > 0x7fffffffu) return TNMF_E_GEOM;
    TNMF_HIP_TRY(hipSetDevice(ctx->device));
    if (rows == 0 || len == 0 || inner == 0) return TNMF_OK;
    if (!in || !out || in == out) return TNMF_E_NULL;
    return launch_convolve_axis(ctx, dtype, in, out, rows, len, (int)inner, kernel, klen, static_cast<hipStream_t>(stream));
}

int tnmf_hip_pad_H(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom, int mode, const void *H, void *Hpad, void *stream) {
    if (is_vol(geom)) return vol_api_pad_fold(ctx, geom, mode, false, H, Hpad, stream);
    ENTER(ctx, geom);
    if (mode < TNMF_MODE_VALID || mode > TNMF_MODE_REFLECT) return TNMF_E_UNSUPPORTED;
    if (g.N > 0 && (!H || !Hpad)) return TNMF_E_NULL;
    if (g.Hs != g.Hx) return TNMF_E_STRIDE;
    return launch_pad_fold(ctx, g, dtype, mode, false, H, Hpad, s);
}

int tnmf_hip_fold_H(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom, int mode, const void *Gpad, void *G, void *stream) {
    if (is_vol(geom)) return vol_api_pad_fold(ctx, geom, mode, true, Gpad, G, stream);
    ENTER(ctx, geom);
    if (mode < TNMF_MODE_VALID || mode > TNMF_MODE_REFLECT) return TNMF_E_UNSUPPORTED;
    if (g.N > 0 && (!Gpad || !G)) return TNMF_E_NULL;
    return launch_pad_fold(ctx, g, dtype, mode, true, Gpad, G, s);
}

int tnmf_hip_update_H(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom, const void *V, const void *W, void *H_inout,
                      void *R_scratch, int r_is_valid, double eps, double sparsity, void *stream) {
    if (is_vol(geom)) return vol_api_update_H(ctx, geom, V, W, H_inout, R_scratch, r_is_valid, eps, sparsity, stream);
    ENTER(ctx, geom);
    if (g.N == 0) return TNMF_OK;
    if (!V || !W || !H_inout) return TNMF_E_NULL;
    void *Rs = R_scratch;
    if (!Rs) {
        if (r_is_valid) return TNMF_E_NULL;
        const Scratch sc = plan_scratch(ctx, g, dtype);
        CHECK(ensure_scratch(ctx, sc.total));
        Rs = ws_at(ctx, sc.r_off);
    }
    if (!r_is_valid) CHECK(do_reconstruct(ctx, g, dtype, W, H_inout, Rs, s));
    double reg = eps;
    if (sparsity > 0) reg += sparsity;  // TransformInvariantNMF.py:227-230
    return do_corr_W(ctx, g, dtype, V, Rs, W, H_inout, nullptr, nullptr, true, reg, s);
}

int tnmf_hip_update_H_ex(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom, int mode, const void *V, const void *W,
                         void *H_inout, void *R_scratch, double eps, double sparsity, double inhibition,
                         double cross_inhibition, const double *kernel0, int len0, const double *kernel1, int len1,
                         const double *kernel2, int len2, void *stream) {
    if (is_vol(geom)) {
        const double *const kern[3] = {kernel0, kernel1, kernel2};
        const int klen[3] = {len0, len1, len2};
        return vol_api_update_H_ex(ctx, geom, mode, V, W, H_inout, R_scratch, eps, sparsity, inhibition, cross_inhibition,
                                   kern, klen, stream);
    }
    (void)kernel2;
    (void)len2;
    ENTER(ctx, geom);
    if (mode < TNMF_MODE_VALID || mode > TNMF_MODE_REFLECT) return TNMF_E_UNSUPPORTED;
    if (g.N == 0) return TNMF_OK;
    if (!V || !W || !H_inout) return TNMF_E_NULL;
    const bool lateral = inhibition > 0 || cross_inhibition > 0;
    if (lateral && (!kernel0 || (geom->ndim == 2 && !kernel1))) return TNMF_E_NULL;
    if (inhibition < 0 || cross_inhibition < 0) return TNMF_E_GEOM;
    double reg = eps;
    if (sparsity > 0) reg += sparsity;  // TransformInvariantNMF.py:227-230
    const size_t es = esize(dtype);
    void *Rs = R_scratch;
    if (!Rs) {
        const Scratch sc = plan_scratch(ctx, g, dtype);
        CHECK(ensure_scratch(ctx, sc.total));
        Rs = ws_at(ctx, sc.r_off);
    }
    // the reference's kernels come one per shift axis (TransformInvariantNMF.py:163): 1-D problems have the x kernel only
    const double one = 1.0;
    const double *ky = geom->ndim == 2 ? kernel0 : &one, *kx = geom->ndim == 2 ? kernel1 : kernel0;
    const int ly = geom->ndim == 2 ? len0 : 1, lx = geom->ndim == 2 ? len1 : len0;
    const double xc = cross_inhibition > 0 && g.M > 1 ? cross_inhibition / (g.M - 1) : 0.0;   // (:266-268)

    if (mode == TNMF_MODE_VALID) {
        const size_t nE = align_up((size_t)g.N * g.M * g.Hy * g.Hs * es, 256);
        const size_t nG = align_up((size_t)g.N * g.M * g.Hy * g.Hx * es, 256);
        void *E = nullptr;
        if (lateral) {
            CHECK(ensure_hwork(ctx, nE));
            E = ctx->hw;
            CHECK(launch_inhibition(ctx, dtype, g.N, g.M, g.Hy, g.Hs, H_inout, E, ky, ly, kx, lx, inhibition, xc, s));
        }
        CHECK(do_reconstruct(ctx, g, dtype, W, H_inout, Rs, s));
        int rc = do_corr_W(ctx, g, dtype, V, Rs, W, H_inout, nullptr, nullptr, true, reg, s, E);
        if (!E || (rc != TNMF_E_UNSUPPORTED && rc != TNMF_E_STRIDE)) return rc;
        // this kernel family has no epilogue for the extra term (nothing has been written): unfused gradient into the
        // work arrays, then one update kernel
        if (ctx->hw_bytes < nE + 2 * nG) {
            // (growing the buffer would lose E: take the larger buffer first, then compute E again)
            CHECK(ensure_hwork(ctx, nE + 2 * nG));
            E = ctx->hw;
            CHECK(launch_inhibition(ctx, dtype, g.N, g.M, g.Hy, g.Hs, H_inout, E, ky, ly, kx, lx, inhibition, xc, s));
        }
        char *neg = static_cast<char *>(ctx->hw) + nE, *pos = neg + nG;
        CHECK(do_corr_W(ctx, g, dtype, V, Rs, W, nullptr, neg, pos, false, 0.0, s));
        fft_invalidate_H(ctx, g, dtype, H_inout);
        return launch_mu_update_extra(ctx, dtype, H_inout, neg, pos, E, (size_t)g.N * g.M * g.Hy, g.Hx, g.Hs, reg, s);
    }

    // reconstruction modes: a 'valid' half step on the padded activations, folded back (adjoint of the pad) inside the
    // update kernel.  H is C-contiguous with the shift shape of the mode.
    if (g.Hs != g.Hx) return TNMF_E_STRIDE;
    const int Sy = geom->ndim == 1 ? 1 : (mode == TNMF_MODE_FULL ? g.Dy - g.Ay + 1 : g.Dy);
    const int Sx = mode == TNMF_MODE_FULL ? g.Dx - g.Ax + 1 : g.Dx;
    if (Sy < 1 || Sx < 1) return TNMF_E_GEOM;
    const size_t nP = align_up((size_t)g.N * g.M * g.Hy * g.Hx * es, 256);
    const size_t nE = lateral ? align_up((size_t)g.N * g.M * Sy * Sx * es, 256) : 0;
    CHECK(ensure_hwork(ctx, 3 * nP + nE));
    char *Hp = static_cast<char *>(ctx->hw), *negp = Hp + nP, *posp = negp + nP;
    void *E = lateral ? posp + nP : nullptr;
    fft_invalidate(ctx);   // the padded copy lives at the same address every call, with new contents
    CHECK(launch_pad_fold(ctx, g, dtype, mode, false, H_inout, Hp, s));
    CHECK(do_reconstruct(ctx, g, dtype, W, Hp, Rs, s));
    CHECK(do_corr_W(ctx, g, dtype, V, Rs, W, nullptr, negp, posp, false, 0.0, s));
    if (lateral) CHECK(launch_inhibition(ctx, dtype, g.N, g.M, Sy, Sx, H_inout, E, ky, ly, kx, lx, inhibition, xc, s));
    fft_invalidate(ctx);
    return launch_fold_update(ctx, g, dtype, mode, Sy, Sx, H_inout, negp, posp, E, reg, s);
}

int tnmf_hip_grad_W_fused(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom, const void *V, const void *W, const void *H,
                          void *R_scratch, int r_is_valid, void *negpos, void *stream) {
    if (is_vol(geom)) {
        if (!negpos || (r_is_valid && !R_scratch)) return TNMF_E_NULL;
        Vol vv;
        CHECK(to_vol(geom, &vv));
        char *np3 = static_cast<char *>(negpos);
        return vol_api_grad_W(ctx, geom, V, R_scratch, r_is_valid, W, H, np3,
                              np3 + (size_t)vv.M * vv.C * vol_avox(vv) * esize(geom->dtype), stream);
    }
    ENTER(ctx, geom);
    if (!negpos || !W || (g.N > 0 && (!V || !H))) return TNMF_E_NULL;
    const Scratch sc = plan_scratch(ctx, g, dtype);
    CHECK(ensure_scratch(ctx, sc.total));
    if (r_is_valid && !R_scratch) return TNMF_E_NULL;
    void *Rs = R_scratch ? R_scratch : ws_at(ctx, sc.r_off);
    if (!r_is_valid) CHECK(do_reconstruct(ctx, g, dtype, W, H, Rs, s));
    char *np = static_cast<char *>(negpos);
    const size_t wbytes = (size_t)g.M * g.C * g.Ay * g.Ax * esize(dtype);
    return do_corr_H(ctx, g, dtype, sc, V, Rs, H, np, np + wbytes, s);
}

int tnmf_hip_run_schedule(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom, const void *V, void *W_inout, void *H_inout,
                          void *R_scratch, void *acc, const tnmf_hip_op *ops, int n_ops, double eps, double sparsity,
                          void *stream) {
    if (is_vol(geom))
        return vol_api_run_schedule(ctx, geom, V, W_inout, H_inout, R_scratch, acc, ops, n_ops, eps, sparsity, stream);
    ENTER(ctx, geom);
    if (n_ops < 0 || (n_ops > 0 && !ops)) return TNMF_E_NULL;
    if (!V || !W_inout || !H_inout || !acc) return TNMF_E_NULL;
    const size_t es = esize(dtype);
    const size_t vs = (size_t)g.C * g.Dy * g.Dx * es, hs = (size_t)g.M * g.Hy * g.Hs * es;
    const size_t wn = (size_t)g.M * g.C * g.Ay * g.Ax;
    // scratch for the largest slice of the list: R of the slice (unless the caller brought one), split-K partials, and the
    // gradient of one batch behind them
    for (int i = 0; i < n_ops; ++i) {
        // (checked for the whole list before anything runs: the persistent kernel walks it on the device)
        if (ops[i].kind != TNMF_OP_UPDATE_H && ops[i].kind != TNMF_OP_GRAD_W && ops[i].kind != TNMF_OP_APPLY_W)
            return TNMF_E_UNSUPPORTED;
        if (ops[i].kind == TNMF_OP_APPLY_W) continue;
        if (ops[i].n0 < 0 || ops[i].n1 < ops[i].n0 || ops[i].n1 > g.N) return TNMF_E_GEOM;
    }
    // Consecutive H half steps commute when their slices are disjoint -- the H update of a sample reads that sample and W
    // only (NumPy.py:93-120), and W does not change between them -- so a RUN of them is executed as the H half step of the
    // union: the slices sorted and joined where they touch.  GSG-MU / GSAG-MU (TransformInvariantNMF.py:474-479,493-504: H for
    // every shuffled batch, W from the last one only) become ONE H half step over all samples + one W step per epoch instead
    // of 256 x 2 launches of three samples each.  Runs with overlapping slices are left as they are (an H step is not
    // idempotent).  Same arithmetic per sample; the kernel family follows the size of the joined slice.
    std::vector<tnmf_hip_op> joined;
    joined.reserve(n_ops);
    for (int i = 0; i < n_ops;) {
        if (ops[i].kind != TNMF_OP_UPDATE_H) {
            joined.push_back(ops[i++]);
            continue;
        }
        int j = i;
        while (j < n_ops && ops[j].kind == TNMF_OP_UPDATE_H) ++j;
        std::vector<tnmf_hip_op> run(ops + i, ops + j);
        run.erase(std::remove_if(run.begin(), run.end(), [](const tnmf_hip_op &o) { return o.n1 <= o.n0; }), run.end());
        std::vector<tnmf_hip_op> sorted_run = run;
        std::sort(sorted_run.begin(), sorted_run.end(), [](const tnmf_hip_op &a, const tnmf_hip_op &b) { return a.n0 < b.n0; });
        bool disjoint = true;
        for (size_t k = 1; k < sorted_run.size(); ++k) disjoint = disjoint && sorted_run[k].n0 >= sorted_run[k - 1].n1;
        if (!disjoint) {
            joined.insert(joined.end(), run.begin(), run.end());
        } else {
            for (size_t k = 0; k < sorted_run.size(); ++k) {
                if (!joined.empty() && joined.back().kind == TNMF_OP_UPDATE_H && k > 0 && joined.back().n1 == sorted_run[k].n0)
                    joined.back().n1 = sorted_run[k].n1;
                else
                    joined.push_back(sorted_run[k]);
            }
        }
        i = j;
    }
    ops = joined.data();
    n_ops = (int)joined.size();
    int nmax = 1;
    for (int i = 0; i < n_ops; ++i)
        if (ops[i].kind != TNMF_OP_APPLY_W && ops[i].n1 - ops[i].n0 > nmax) nmax = ops[i].n1 - ops[i].n0;
    double reg = eps;
    if (sparsity > 0) reg += sparsity;  // TransformInvariantNMF.py:227-230
    // A whole problem that is tiny (BASELINE config 1): every kernel would run for a few microseconds and the list would be
    // launch latency -- the persistent schedule kernel walks it in ONE launch.  (Small batches of a LARGE problem stay on
    // the per-operation path below: measured on the reference's mini-batch geometry, 768 x 1 x 32 x 32 with batch_size 3,
    // the grid barriers of the persistent kernel -- agent-scope release / acquire across eight L2s -- cost as much as
    // the launches they replace: 28.5 ms per ASG epoch against 26.7 ms.)
    // (Round 4 built an XCD-local flavour of that kernel for exactly that case -- the workgroups of ONE XCD, barriers without
    // the L2 write-back: 22.3 ms per ASG epoch against 12.0 on the per-operation path below, profiles/r04_xcd_local_schedule_kernel.txt.)
    (void)nmax;
    const bool tiny = (size_t)g.N * g.M * g.Hy * g.Hx <= ((size_t)1 << 18);
    if (n_ops > 0 && tiny && ctx->persistent != 0 && (ctx->path == TNMF_PATH_AUTO || ctx->path == TNMF_PATH_GENERIC) &&
        generic_schedule_fits(ctx, g, dtype)) {
        const int P = generic_schedule_chunks(ctx, g);
        const size_t r_bytes = R_scratch ? 0 : align_up((size_t)g.N * vs, 256);
        const size_t p_bytes = align_up((size_t)P * wn * 2 * sizeof(double), 256);
        const size_t o_bytes = align_up((size_t)n_ops * sizeof(tnmf_hip_op), 256);
        CHECK(ensure_scratch(ctx, r_bytes + p_bytes + o_bytes + 1024));
        void *Rs = R_scratch ? R_scratch : static_cast<void *>(ws_at(ctx, 0));
        double *partials = reinterpret_cast<double *>(ws_at(ctx, r_bytes));
        tnmf_hip_op *ops_dev = reinterpret_cast<tnmf_hip_op *>(ws_at(ctx, r_bytes + p_bytes));
        unsigned *counter = reinterpret_cast<unsigned *>(ws_at(ctx, r_bytes + p_bytes + o_bytes));
        {
            const int slot = ctx->ops_next;
            ctx->ops_next = (slot + 1) % tnmf_hip_ctx::kOpSlots;
            const size_t need = (size_t)n_ops * sizeof(tnmf_hip_op);
            if (ctx->ops_done[slot]) TNMF_HIP_TRY(hipEventSynchronize(ctx->ops_done[slot]));   // (4 calls ago: long done)
            else TNMF_HIP_TRY(hipEventCreateWithFlags(&ctx->ops_done[slot], hipEventDisableTiming));
            if (ctx->ops_cap[slot] < need) {
                if (ctx->ops_pinned[slot]) TNMF_HIP_TRY(hipHostFree(ctx->ops_pinned[slot]));
                ctx->ops_pinned[slot] = nullptr;
                ctx->ops_cap[slot] = 0;
                if (hipHostMalloc(&ctx->ops_pinned[slot], align_up(need, 4096), hipHostMallocDefault) != hipSuccess) {
                    (void)hipGetLastError();
                    return TNMF_E_WORKSPACE;
                }
                ctx->ops_cap[slot] = align_up(need, 4096);
            }
            memcpy(ctx->ops_pinned[slot], ops, need);
            TNMF_HIP_TRY(hipMemcpyAsync(ops_dev, ctx->ops_pinned[slot], need, hipMemcpyHostToDevice, s));
            TNMF_HIP_TRY(hipEventRecord(ctx->ops_done[slot], s));
        }
        fft_invalidate(ctx);   // H and W change under the family's caches
        ctx->last_path = "generic";
        const int rc = generic_run_schedule(ctx, g, dtype, V, W_inout, H_inout, Rs, acc, partials, P, ops_dev, n_ops, reg,
                                            eps, counter, s);
        // TNMF_E_UNSUPPORTED: the grid cannot be co-resident on this device right now (occupancy query / cooperative
        // launch refused) -- nothing was launched; the list is walked operation by operation below
        if (rc != TNMF_E_UNSUPPORTED) {
            ctx->last_schedule_persistent = rc == TNMF_OK;
            return rc;
        }
    }
    ctx->last_schedule_persistent = false;
    Geo gmax = g;
    gmax.N = nmax;
    const Scratch scm = plan_scratch(ctx, gmax, dtype);
    const size_t grad_off = scm.total;
    CHECK(ensure_scratch(ctx, scm.total + align_up(2 * wn * es, 256)));
    char *grad = ws_at(ctx, grad_off);
    for (int i = 0; i < n_ops; ++i) {
        const tnmf_hip_op &op = ops[i];
        Geo gs = g;
        gs.N = op.n1 - op.n0;
        const char *Vb = static_cast<const char *>(V) + (size_t)op.n0 * vs;
        char *Hb = static_cast<char *>(H_inout) + (size_t)op.n0 * hs;
        void *Rb = R_scratch ? static_cast<void *>(static_cast<char *>(R_scratch) + (size_t)op.n0 * vs)
                             : static_cast<void *>(ws_at(ctx, scm.r_off));
        switch (op.kind) {
            case TNMF_OP_UPDATE_H:
                if (gs.N == 0) break;
                CHECK(do_reconstruct(ctx, gs, dtype, W_inout, Hb, Rb, s));
                CHECK(do_corr_W(ctx, gs, dtype, Vb, Rb, W_inout, Hb, nullptr, nullptr, true, reg, s));
                break;
            case TNMF_OP_GRAD_W: {
                // (the partials of the slice live where plan_scratch(gmax) put them: sized for the largest slice)
                Scratch sc = plan_scratch(ctx, gs, dtype);
                sc.part_off = scm.part_off;
                if (gs.N > 0) CHECK(do_reconstruct(ctx, gs, dtype, W_inout, Hb, Rb, s));
                if (gs.N > 0 && !corr_H_on_fft(ctx, gs, dtype)) {
                    // direct kernels: split-K partials, then ONE launch for their fixed-order sum, the blend into the
                    // accumulator and -- when the W update is the next operation -- that update as well
                    int P = 1;
                    double *partials = reinterpret_cast<double *>(ws_at(ctx, scm.part_off));
                    CHECK(do_corr_H_partials(ctx, gs, dtype, Vb, Rb, Hb, partials, &P, s));
                    const bool apply_now = i + 1 < n_ops && ops[i + 1].kind == TNMF_OP_APPLY_W;
                    if (apply_now) fft_invalidate_W(ctx);
                    CHECK(launch_finalize_blend_apply(g, dtype, partials, P, acc, op.a, op.b, apply_now, W_inout, eps, s));
                    if (apply_now) ++i;
                    break;
                }
                CHECK(do_corr_H(ctx, gs, dtype, sc, Vb, Rb, Hb, grad, grad + wn * es, s));
                CHECK(launch_axpby(ctx, dtype, acc, grad, op.a, op.b, 2 * wn, s));
                break;
            }
            case TNMF_OP_APPLY_W: {
                fft_invalidate_W(ctx);
                char *np = static_cast<char *>(acc);
                CHECK(launch_apply_normalize_W(g, dtype, W_inout, np, np + wn * es, eps, true, s));
                break;
            }
            default: return TNMF_E_UNSUPPORTED;
        }
    }
    return TNMF_OK;
}

int tnmf_hip_axpby(tnmf_hip_ctx *ctx, int dtype, void *acc, const void *g, double a, double b, size_t n_elems,
                   void *stream) {
    if (!ctx) return TNMF_E_NULL;
    if (dtype != 0 && dtype != 1) return TNMF_E_DTYPE;
    if (n_elems > 0 && (!acc || !g)) return TNMF_E_NULL;
    TNMF_HIP_TRY(hipSetDevice(ctx->device));
    return launch_axpby(ctx, dtype, acc, g, a, b, n_elems, static_cast<hipStream_t>(stream));
}

int tnmf_hip_sum_parts(tnmf_hip_ctx *ctx, int dtype, const void *parts, int n_parts, size_t n_elems, void *out,
                       void *stream) {
    if (!ctx) return TNMF_E_NULL;
    if (dtype != 0 && dtype != 1) return TNMF_E_DTYPE;
    if (n_parts < 1) return TNMF_E_GEOM;
    if (n_elems > 0 && (!parts || !out)) return TNMF_E_NULL;
    TNMF_HIP_TRY(hipSetDevice(ctx->device));
    return launch_sum_parts(ctx, dtype, parts, n_parts, n_elems, out, static_cast<hipStream_t>(stream));
}

int tnmf_hip_apply_W(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom, void *W_inout, void *negpos, double eps,
                     void *stream) {
    if (is_vol(geom)) {
        VOL_ENTER(ctx, geom);
        if (!W_inout || !negpos) return TNMF_E_NULL;
        char *np3 = static_cast<char *>(negpos);
        return launch_apply_normalize_W(vol_dict_geo(v), dtype, W_inout, np3,
                                        np3 + (size_t)v.M * v.C * vol_avox(v) * esize(dtype), eps, true, s);
    }
    ENTER(ctx, geom);
    if (!W_inout || !negpos) return TNMF_E_NULL;
    fft_invalidate_W(ctx);
    char *np = static_cast<char *>(negpos);
    const size_t wbytes = (size_t)g.M * g.C * g.Ay * g.Ax * esize(dtype);
    return launch_apply_normalize_W(g, dtype, W_inout, np, np + wbytes, eps, true, s);
}

}  // extern "C"
