// split.hip -- host dispatch of the split (3 x bf16) H-gradient kernels; the kernels live in split_kernels.h and are
// instantiated one shape per object file (split_shape.hip).
#include "split.h"

// instantiated (atom rows, runs per atom row = ceil(Ax / 4)) pairs -- keep in step with SPLIT_SHAPES of the Makefile
// (1, n): the 1-D instantiations -- signals with atoms of up to 4 n taps, eight samples per tile
#define TNMF_SPLIT_SHAPES(X) X(12, 3) X(9, 3) X(16, 4) X(7, 2) X(8, 2) X(5, 2) X(1, 4) X(1, 8) X(1, 16)

#define DECL(AY_, NR4_)                                                                                              \
    int split_launch_##AY_##_##NR4_(tnmf_hip_ctx *ctx, const Geo &g, const float *V, const float *R, const float *W, \
                                    float *H_inout, float *neg, float *pos, bool fused, float reg, hipStream_t s,    \
                                    const float *extra);                                                             \
    int split_prepare_##AY_##_##NR4_();
TNMF_SPLIT_SHAPES(DECL)
#undef DECL

// The instantiation that runs a shape: template atom rows AY_i >= g.Ay and runs NR4_i >= ceil(g.Ax / 4) -- the kernel takes
// the true atom shape at run time (window offsets, tile counts) and the operand image holds zeros for the rows and taps
// the atom does not have -- so EVERY atom up to 16 x 16 runs on the bf16 matrix cores; among the instantiations that
// cover a shape the one with the fewest k blocks (least zero work) is taken.  Returns AY_i * 100 + NR4_i, 0 for none.
static int split_pick(const Geo &g) {
    if (g.Dy == 1 && g.Ay == 1) return 100 + (g.Ax <= 16 ? 4 : (g.Ax <= 32 ? 8 : 16));   // 1-D: whole k blocks of 16 taps
    const int nr4 = (g.Ax + 3) / 4;
    int best = 0, best_kb = 1 << 30;
#define CONSIDER(AY_, NR4_)                                             \
    if (AY_ > 1 && AY_ >= g.Ay && NR4_ >= nr4) {                        \
        const int kb = (((AY_ + 1) / 2) * NR4_ + 1) / 2;                \
        if (kb < best_kb) {                                             \
            best_kb = kb;                                               \
            best = AY_ * 100 + NR4_;                                    \
        }                                                               \
    }
    TNMF_SPLIT_SHAPES(CONSIDER)
#undef CONSIDER
    return best;
}

bool split_has_corr_W(const Geo &g, int dtype, bool only_if_worth) {
    if (dtype != 0) return false;
    if (g.Dy == 1 && g.Ay == 1) {
        // 1-D signals: the rows of a tile are samples; atoms of up to 64 taps
        return g.Ax <= 64 && g.Dx >= 4 && (size_t)align_up((size_t)g.M, 32) * (size_t)(g.Hs > g.Hx ? g.Hs : g.Hx) * 4 < ((size_t)1 << 31);
    }
    if (g.Dy == 1 || g.Ay == 1) return false;   // (a 2-D problem with one-row samples or atoms: other kernels)
    if (g.Ax > 16 || g.Ay > 16 || g.Dx < 4) return false;   // (16-byte window loads need 4 columns)
    // H / neg / pos of one sample are addressed through 32-bit buffer offsets (atom * plane bytes + ...): all the planes
    // of a sample, rounded up to whole atom tiles, must stay below 2^31 bytes or the offsets wrap
    if ((size_t)align_up((size_t)g.M, 32) * g.Hy * (size_t)(g.Hs > g.Hx ? g.Hs : g.Hx) * 4 >= ((size_t)1 << 31)) return false;
    const int pick = split_pick(g);
    if (!pick) return false;
    if (only_if_worth) {
        // the instantiation's k blocks (16 taps each, zeros included) at 16/6 of the f32 matrix rate against the atom's own
        // taps on the exact f32 kernels: a small atom on a much larger instantiation is not worth it
        const int AYi = pick / 100, NR4i = pick % 100, kb = (((AYi + 1) / 2) * NR4i + 1) / 2;
        if (6 * kb > g.Ay * g.Ax) return false;
    }
    return true;
}

int split_prepare_device() {
    int rc;
#define PREP(AY_, NR4_) \
    if ((rc = split_prepare_##AY_##_##NR4_()) != TNMF_OK) return rc;
    TNMF_SPLIT_SHAPES(PREP)
#undef PREP
    return TNMF_OK;
}

int split_corr_W(tnmf_hip_ctx *ctx, const Geo &g, const float *V, const float *R, const float *W, float *H_inout,
                 float *neg, float *pos, bool fused, float reg, hipStream_t s, const float *extra) {
    const int pick = split_pick(g);
#define DISPATCH(AY_, NR4_) \
    if (pick == AY_ * 100 + NR4_) return split_launch_##AY_##_##NR4_(ctx, g, V, R, W, H_inout, neg, pos, fused, reg, s, extra);
    TNMF_SPLIT_SHAPES(DISPATCH)
#undef DISPATCH
    return TNMF_E_UNSUPPORTED;
}

void split_release(tnmf_hip_ctx *ctx) {
    if (ctx->wimg) (void)hipFree(ctx->wimg);
    ctx->wimg = nullptr;
    ctx->wimg_bytes = 0;
}
