// Host-side entry points of generic.hip (see there for the kernels).
#pragma once
#include "common.h"

constexpr int kMaxTaps = 127;          // longest 1-D inhibition kernel (2 * range + 1)
constexpr int kEnergyPartials = 2048;  // per-block partial sums of the energy reduction

int generic_reconstruct(tnmf_hip_ctx *ctx, const Geo &g, int dtype, const void *W, const void *H, void *R,
                        hipStream_t s);
// fused == false: writes neg/pos.  fused == true: H = (H * neg) / (pos + reg) in place (neg/pos unused).
// extra (fused only, may be NULL): a further term of the denominator, laid out like H
int generic_corr_W(tnmf_hip_ctx *ctx, const Geo &g, int dtype, const void *V, const void *R, const void *W,
                   void *H_inout, void *neg, void *pos, bool fused, double reg, hipStream_t s,
                   const void *extra = nullptr);
// split-K partial sums: doubles, [P][M*C][Ay*Ax][2] in unflipped shift order
int generic_corr_H_chunks(const tnmf_hip_ctx *ctx, const Geo &g);
int generic_corr_H(tnmf_hip_ctx *ctx, const Geo &g, int dtype, const void *V, const void *R, const void *H,
                   double *partials, int P, hipStream_t s);
// fixed-order sum of the partials in double; writes neg/pos[M,C,*A] in the reference's (flipped) orientation
int finalize_corr_H(const Geo &g, int dtype, const double *partials, int P, void *neg, void *pos, hipStream_t s);

int launch_mu_update(const tnmf_hip_ctx *ctx, int dtype, void *arr, const void *neg, void *pos, double reg, size_t n,
                     hipStream_t s);
int launch_axpby(const tnmf_hip_ctx *ctx, int dtype, void *acc, const void *g, double a, double b, size_t n,
                 hipStream_t s);
int launch_sum_parts(const tnmf_hip_ctx *ctx, int dtype, const void *parts, int n_parts, size_t n, void *out,
                     hipStream_t s);
int launch_apply_normalize_W(const Geo &g, int dtype, void *W, const void *neg, void *pos, double eps, bool apply,
                             hipStream_t s);
int launch_half_sqdiff(const tnmf_hip_ctx *ctx, int dtype, const void *V, const void *R, size_t n, double *partials,
                       double *out_dev, hipStream_t s);
int launch_convolve_axis(const tnmf_hip_ctx *ctx, int dtype, const void *in, void *out, size_t rows, int len,
                         int inner, const double *kernel_host, int ntaps, hipStream_t s);
// reconstruction modes: pad activations (fold == false) / fold the gradient back (fold == true); mode = TNMF_MODE_*
int launch_pad_fold(const tnmf_hip_ctx *ctx, const Geo &g, int dtype, int mode, bool fold, const void *in, void *out,
                    hipStream_t s);

// inhibit.hip: lateral terms of the H half step and the fold + update of the reconstruction modes
// E[N][M][Hy][ld] = inh * (G - H) + xc * (sum over atoms of G - G), G = separable zero-padded convolution of H (same layout)
int launch_inhibition(const tnmf_hip_ctx *ctx, int dtype, int N, int M, int Hy, int ld, const void *H, void *E,
                      const double *ky_host, int ly, const double *kx_host, int lx, double inh, double xc,
                      hipStream_t s);
// H[rows][Hx] (row stride ld) = H * neg / (pos + E + reg); neg / pos C-contiguous, E laid out like H or NULL
int launch_mu_update_extra(const tnmf_hip_ctx *ctx, int dtype, void *H, const void *neg, const void *pos, const void *E,
                           size_t rows, int Hx, int ld, double reg, hipStream_t s);
// modes: H[N][M][Sy][Sx] = H * fold(negp) / (fold(posp) + E + reg), negp / posp on the padded shape (g.Hy, g.Hx)
int launch_fold_update(const tnmf_hip_ctx *ctx, const Geo &g, int dtype, int mode, int Sy, int Sx, void *H,
                       const void *negp, const void *posp, const void *E, double reg, hipStream_t s);

// the persistent schedule kernel (k_schedule): a whole operation list of tnmf_hip_run_schedule in one launch, on the generic
// kernels' device functions.  ops_dev: the operations in DEVICE memory; partials: P * M * C * Ay * Ax * 2 doubles;
// counter: one zero-initialisable word; R: [g.N, C, *D] (every sample's own slot)
bool generic_schedule_fits(const tnmf_hip_ctx *ctx, const Geo &g, int dtype);
int generic_schedule_chunks(const tnmf_hip_ctx *ctx, const Geo &g);
int generic_run_schedule(tnmf_hip_ctx *ctx, const Geo &g, int dtype, const void *V, void *W, void *H, void *R, void *acc,
                         double *partials, int P, const tnmf_hip_op *ops_dev, int n_ops, double reg, double eps,
                         unsigned *counter, hipStream_t s);
// mini-batch step behind the split-K kernel: acc = a * acc + b * (fixed-order sum of the partials), then optionally the W
// update from acc (W = W * acc_neg / (acc_pos + eps), normalised) -- one launch
int launch_finalize_blend_apply(const Geo &g, int dtype, const double *partials, int P, void *acc, double a, double b,
                                bool apply, void *W, double eps, hipStream_t s);
