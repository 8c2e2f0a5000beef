/*
 * tnmf_hip.h -- C ABI of libtnmf_hip.so: the MI355X (gfx950) kernels behind the 'hip' backend of the
 * shift-invariant multiplicative-update loop of emdgroup/tnmf.
 *
 * The reference has no FFI on this path (it is pure Python); the boundary it does have is the Python class
 * tnmf/backends/_Backend.py:13-130.  Each entry point below names the reference method it stands under.  They are
 * bound from Python with ctypes (tnmf_amd/_lib.py); INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to a C-contiguous buffer owned by the caller (e.g. torch.Tensor.data_ptr());
 *     the library never frees or keeps it beyond the call.  `stream` is a hipStream_t passed as void* (NULL = default).
 *   - every function returns int: 0 ok, <0 argument error (TNMF_E_*), >0 a hipError_t.  Nothing throws.
 *   - element type: geom.dtype 0 = float32, 1 = float64.  'valid' reconstruction mode only:
 *       V[N,C,*D]  W[M,C,*A]  H[N,M,*(D+A-1)]  R[N,C,*D];   ndim = 1 or 2 shift axes.
 *   - mini-batch slices are contiguous along the sample axis: the caller offsets V/H/R pointers and passes the slice's N.
 *   - one context = one device.  SURVEY.md 8b sketched a multi-device context (tnmf_hip_ctx_create(device_ids[], n)) and
 *     an in-library tnmf_hip_allreduce_negpos; this library deliberately has neither: the product runs one process per
 *     GPU and the single collective of the path -- the sum of the [neg | pos] buffer of tnmf_hip_grad_W_fused -- is done
 *     by the caller between tnmf_hip_grad_W_fused and tnmf_hip_apply_W (torch.distributed / RCCL in tnmf_amd/backends/HIP.py).
 *   - calls on one ctx are not thread-safe; different ctxs are independent.  All launches are asynchronous on
 *     `stream` except tnmf_hip_energy, which synchronises the stream to return its scalar.
 */
#ifndef TNMF_HIP_H
#define TNMF_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TNMF_HIP_ABI_VERSION 7

enum {
    TNMF_OK = 0,
    TNMF_E_NULL = -1,      /* required pointer is NULL */
    TNMF_E_GEOM = -2,      /* bad geometry (ndim, sizes <= 0, atom larger than supported) */
    TNMF_E_DTYPE = -3,     /* dtype not 0/1 */
    TNMF_E_WORKSPACE = -4, /* workspace allocation failed */
    TNMF_E_UNSUPPORTED = -5,
    TNMF_E_STRIDE = -6     /* h_row_stride > shift width, and the kernel family forced for this call wants C-contiguous H */
};

typedef struct tnmf_hip_ctx tnmf_hip_ctx;

typedef struct {
    int ndim;  /* 1, 2 or 3 shift axes (3: volumes, axes z, y, x -- see "Volumes" below) */
    int N;     /* samples in this call (mini-batch slice length) */
    int M;     /* atoms */
    int C;     /* channels */
    int D[3];  /* sample shape; the first ndim entries are used */
    int A[3];  /* atom shape;   the first ndim entries are used */
    int dtype; /* 0 = f32, 1 = f64 */
    /* Row stride of H in elements; 0 (or the shift width D[last] + A[last] - 1) = C-contiguous, as the reference's arrays
     * are.  A larger value describes activations whose rows are padded to whole cache lines: H[n,m,y,x] at
     * ((n*M + m)*Hy + y)*h_row_stride + x, the plane and sample strides following from it.  Supported where H is
     * streamed by the FFT family, the split kernel and the generic kernels -- everything TNMF_PATH_AUTO dispatches to for
     * such activations; only the f32 MFMA kernels want C-contiguous rows (AUTO skips them for padded ones; with
     * TNMF_PATH_MFMA forced the call answers TNMF_E_STRIDE and touches nothing -- the caller then passes a contiguous copy).
     * Outputs shaped like H (neg / pos of tnmf_hip_grad_H) are always C-contiguous. */
    int h_row_stride;
} tnmf_hip_geom;

/* Kernel family selection (tnmf_hip_ctx_set_path): AUTO picks the MFMA kernels where the shape allows and, for float32
 * problems with at least 2^19 activation entries, the HYBRID dispatch described below.  FFT is the
 * frequency-domain formulation (the algorithm of the reference's default backend, tnmf/backends/NumPy_FFT.py:16-40):
 * float32 2-D problems with shift shapes up to 576, float64 up to 288.  In float32 FFT is a W-ONLY path: the dictionary
 * and the energy of a fit stay within 1e-5 of a float64 reference, the activations do not (every gradient entry carries
 * ~1e-7 of the LARGEST entry as absolute transform error, and the H update divides two gradients); callers who need H at
 * float32 grade use AUTO / HYBRID, whose H update runs on the direct kernels. */
enum { TNMF_PATH_AUTO = 0, TNMF_PATH_GENERIC = 1, TNMF_PATH_MFMA = 2, TNMF_PATH_FFT = 3, TNMF_PATH_HYBRID = 4,
       TNMF_PATH_SPLIT = 5 };
/* HYBRID: reconstruct and the W gradient on the FFT family (their float32 transform error is benign: R has no small
 * entries, the W gradient is a sum over all samples), the H gradient / fused H update on the direct kernels (exact
 * summation of the few-tap border entries).  Falls back to AUTO where the FFT family does not cover the shape.
 * The FFT family (under FFT and HYBRID alike) assumes non-negative factors: R and the W gradient are clamped at zero
 * from below, which only removes transform rounding noise (V, W, H >= 0 imply both >= 0) and keeps the denominators
 * of the multiplicative updates non-negative. */

/* SPLIT: the direct kernels, with the H gradient / fused H update on the bf16 matrix cores: every float32 operand is
 * split exactly into three bf16 terms and a product is the sum of the six term products of weight >= 2^-16 -- float32-grade
 * results (error against a float64 reference no larger than the f32 MFMA chain's) at 16/6 of the f32 matrix rate.
 * MFMA keeps every kernel on the exact f32-input MFMA (a k-ordered fmaf chain).  AUTO and HYBRID use the split H update
 * where it covers the shape (float32, 2-D, atoms up to 16 x 16) unless tnmf_hip_ctx_set_split(ctx, 0) turned it off. */
int tnmf_hip_ctx_set_split(tnmf_hip_ctx *ctx, int enable);

/* tnmf_hip_run_schedule walks the operation list of a TINY resident problem inside ONE persistent kernel whose workgroups
 * meet at grid-wide barriers (generic.hip: k_schedule) -- which is only sound while every workgroup of the grid is
 * resident.  The library sizes that grid with an occupancy query of the kernel's real footprint on this device; when not
 * even that fits, or under mode 2 the runtime refuses the cooperative launch, the list is walked operation by operation
 * instead (same arithmetic, more launches).  mode 0: never use the persistent kernel (a caller that shares the GPU with
 * other processes -- CU masks, co-tenants -- should say so); 1 (default): plain launch of the occupancy-sized grid;
 * 2: the same grid through hipLaunchCooperativeKernel.  (The reference has no counterpart: its schedules are Python
 * loops, tnmf/TransformInvariantNMF.py:457-504.) */
int tnmf_hip_ctx_set_persistent(tnmf_hip_ctx *ctx, int mode);
/* 1 when the last tnmf_hip_run_schedule on this ctx ran as one persistent launch, 0 when it walked the list per operation. */
int tnmf_hip_ctx_last_schedule_persistent(const tnmf_hip_ctx *ctx);

int tnmf_hip_abi_version(void);
/* Row stride (elements) this context would like H of `geom` to have: the shift width itself, or -- when the H update runs
 * on the split kernel next to the FFT family -- the shift width rounded up to 32 floats, so that every 32-pixel tile of a
 * row is exactly one 128-byte line (267-float rows make each tile straddle two lines: 1.7x the H traffic, measured). */
int tnmf_hip_ctx_h_row_stride(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom, int *stride_out);
const char *tnmf_hip_strerror(int code);

/* One context per device: caches device properties and owns the scratch (R, split-K partials). */
int tnmf_hip_ctx_create(int device_id, tnmf_hip_ctx **out);
int tnmf_hip_ctx_destroy(tnmf_hip_ctx *ctx);
/* Pre-size the scratch for `geom` so that later calls allocate nothing (graph-capture safe).  Also forgets a workspace
 * size the FFT family was refused earlier (such a size is otherwise not attempted again: under TNMF_PATH_AUTO the
 * direct kernels take over silently), so call it again after freeing device memory. */
int tnmf_hip_ctx_reserve(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom);
int tnmf_hip_ctx_set_path(tnmf_hip_ctx *ctx, int path);
/* Name of the kernel family the last primitive call on this ctx dispatched to ("generic", "mfma", "split", "fft"). */
const char *tnmf_hip_ctx_last_path(const tnmf_hip_ctx *ctx);
/* FFT family only: the library may keep the row spectra of the activations H it transformed or updated last, and the
 * spectra of the samples V it transformed last, and reuse them while the same pointers and geometry come back (the
 * reference's NumPy_CachingFFT.py:22-140 caches spectra the same way).  Off by default.  A caller that enables it
 * vouches that H changes only through this library and V not at all, and calls tnmf_hip_ctx_invalidate() after
 * writing either by any other means. */
int tnmf_hip_ctx_set_cache(tnmf_hip_ctx *ctx, int enable);
int tnmf_hip_ctx_invalidate(tnmf_hip_ctx *ctx);
/* The resident problem of a fit: geom->N samples of activations at H (row stride geom->h_row_stride) and of samples at V
 * (may be NULL).  With the cache enabled, a later call whose H pointer is a whole number of samples into this H -- a
 * mini-batch slice, the way the reference's backends receive `H[s]` (tnmf/backends/NumPy.py:77-80,101) -- works on the
 * matching sample range of ONE cache with per-sample validity: a Cyclic-MU epoch transforms every batch once, like a
 * full-batch iteration (the reference's counterpart: per-slice caches, tnmf/backends/NumPy_CachingFFT.py:143-158).
 * (TNMF_PATH_AUTO still picks the kernel family by the size of the slice itself: small batches take the direct kernels,
 * which read row-padded activations through the row stride.)  Without a binding the cache
 * follows the operands of the last call.  H == NULL or geom == NULL drops the binding.  The binding holds no reference:
 * the caller re-binds (or invalidates) when the buffers are replaced. */
int tnmf_hip_ctx_bind(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom, const void *H, const void *V);
/* Observability of the cache: out[0] / out[1] = row-transform passes over activations that ran / were skipped because
 * the cache held every sample of the call, out[2] / out[3] the same for the samples V (counted since ctx creation). */
int tnmf_hip_ctx_cache_counters(const tnmf_hip_ctx *ctx, unsigned long long out[4]);

/* ---- primitives: API-parity path --------------------------------------------------------------------------- */

/* Backend.reconstruct (tnmf/backends/_Backend.py:120-122; NumPy.py:122-132):
 *   R[n,c,d] = sum_m sum_a H[n,m,d+a] * W[m,c,A-1-a] */
int tnmf_hip_reconstruct(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom, const void *W, const void *H, void *R,
                         void *stream);

/* Backend.reconstruction_gradient_H (_Backend.py:110-118; NumPy.py:93-120):
 *   neg[n,m,u] = sum_c sum_a W[m,c,a] * Vpad[n,c,u+a],  pos = same with R.  R == NULL: R is computed internally. */
int tnmf_hip_grad_H(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom, const void *V, const void *R_or_null,
                    const void *W, const void *H, void *neg, void *pos, void *stream);

/* Backend.reconstruction_gradient_W (_Backend.py:100-108; NumPy.py:69-91):
 *   neg[m,c,a] = sum_n sum_d H[n,m,d+A-1-a] * V[n,c,d],  pos = same with R.  Deterministic two-stage reduction. */
int tnmf_hip_grad_W(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom, const void *V, const void *R_or_null,
                    const void *W, const void *H, void *neg, void *pos, void *stream);

/* TransformInvariantNMF._multiplicative_update (tnmf/TransformInvariantNMF.py:217-235), elementwise part:
 *   pos += reg (IN PLACE, as the reference does);  arr = (arr * neg) / pos.   dtype 0/1, n_elems elements. */
int tnmf_hip_mu_update(tnmf_hip_ctx *ctx, int dtype, void *arr, const void *neg, void *pos, double reg,
                       size_t n_elems, void *stream);

/* Backend.normalize over the atom axes (_Backend.py:75-77 as called from TransformInvariantNMF.py:237-238):
 *   W[m,c,:] /= sum_a W[m,c,a] */
int tnmf_hip_normalize_W(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom, void *W, void *stream);

/* Backend.reconstruction_energy (_Backend.py:127-130): *out_host = 1/2 sum (V - R)^2 in double.  Synchronises. */
int tnmf_hip_energy(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom, const void *V, const void *W, const void *H,
                    double *out_host, void *stream);

/* Backend.convolve_multi_1d (_Backend.py:79-81; _NumPyBackend.py:56-64): zero-padded 'same' convolution of
 * arr[rows, *shape] along each shift axis with an odd-length kernel (host pointers kernel0/kernel1, doubles). */
int tnmf_hip_convolve_multi_1d(tnmf_hip_ctx *ctx, int dtype, int ndim, size_t rows, const int *shape,
                               const void *in, void *out, void *tmp, const double *kernel0, int len0,
                               const double *kernel1, int len1, void *stream);

/* One axis of the same convolution, for any number of shift axes: arr viewed as [rows][len][inner], convolved along
 * `len` (zero-padded 'same', odd-length kernel of at most 127 taps, host doubles), out != in.  convolve_multi_1d over k
 * axes is k calls in the reference's order, first shift axis first (_NumPyBackend.py:60-62), ping-ponging two buffers --
 * how the three axes of a volume are done. */
int tnmf_hip_convolve_axis(tnmf_hip_ctx *ctx, int dtype, size_t rows, int len, size_t inner, const void *in, void *out,
                           const double *kernel, int klen, void *stream);

/* ---- Volumes (ndim == 3) -----------------------------------------------------------------------------------------
 * The reference takes any number of shift axes in NumPy (tnmf/backends/NumPy.py:69-132 contracts over all of them) and
 * one to three in PyTorch (tnmf/backends/PyTorch.py:13-17: conv1d / conv2d / conv3d).  With ndim == 3 the entry points
 * tnmf_hip_reconstruct, _grad_H, _grad_W, _grad_W_fused, _update_H, _apply_W, _normalize_W, _energy, _pad_H, _fold_H and
 * _ctx_reserve run direct kernels of their own (float32 and float64, C-contiguous activations: h_row_stride 0 or the
 * shift width), and so do tnmf_hip_update_H_ex and tnmf_hip_run_schedule (one launch chain per list); _mu_update, _axpby,
 * _sum_parts and _convolve_axis do not look at the geometry.  tnmf_hip_ctx_bind is accepted and has nothing to do.  tnmf_hip_ctx_last_path reads "volume". */

/* ---- reconstruction modes other than 'valid' --------------------------------------------------------------------
 * Every mode of the reference is a 'valid' reconstruction of padded activations (padding table:
 * tnmf/backends/_PyTorchBackend.py:42-52, applied in tnmf/backends/PyTorch.py:36-43): 'full' pads A-1 zeros on both
 * sides, 'circular' / 'reflect' pad A-1 wrapped / mirrored elements on the left.  The activation tensor of mode `mode`
 * has shift shape S = D-A+1 ('full') or D ('circular', 'reflect'); the padded one always has D+A-1, which is what all
 * the primitives above take.  The H gradient of a mode is the 'valid' gradient folded back by the adjoint of the pad. */
enum { TNMF_MODE_VALID = 0, TNMF_MODE_FULL = 1, TNMF_MODE_CIRCULAR = 2, TNMF_MODE_REFLECT = 3 };

/* Hpad[N,M,*(D+A-1)] = pad(H[N,M,*S]) */
int tnmf_hip_pad_H(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom, int mode, const void *H, void *Hpad, void *stream);
/* G[N,M,*S] = pad^T(Gpad[N,M,*(D+A-1)]): every padded position's gradient is summed into the activation it copies */
int tnmf_hip_fold_H(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom, int mode, const void *Gpad, void *G, void *stream);

/* ---- fused half steps: performance path (same math as the primitives + mu_update) --------------------------- */

/* TransformInvariantNMF._update_H without inhibition (TransformInvariantNMF.py:246-250,271):
 *   R = reconstruct(W,H);  H *= corr(W,V) / (corr(W,R) + eps + sparsity), in place.
 *   R_scratch: device buffer [N,C,*D] or NULL (ctx scratch is used).  r_is_valid != 0: R_scratch already holds
 *   reconstruct(W, H) (the caller ran tnmf_hip_reconstruct itself, e.g. to time the two kernels separately). */
int tnmf_hip_update_H(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom, const void *V, const void *W, void *H_inout,
                      void *R_scratch, int r_is_valid, double eps, double sparsity, void *stream);

/* TransformInvariantNMF._update_H in full (TransformInvariantNMF.py:246-271), for every reconstruction mode:
 *   G   = kernel0 (*) kernel1 (*) H  along the shift axes, zeros outside (Backend.convolve_multi_1d, _NumPyBackend.py:56-64)
 *   pos += inhibition * (G - H) + cross_inhibition / (M - 1) * (sum over atoms of G - G)            (:256-269)
 *   H  *= neg / (pos + eps + sparsity)
 * kernel0 / kernel1 / kernel2: HOST pointers to the odd-length 1-D kernels of the shift axes, first axis first (ndim == 1:
 * kernel0 only, ndim == 2: kernel0 and kernel1), ignored when both strengths are 0.  mode == TNMF_MODE_VALID: H as for tnmf_hip_update_H (row stride honoured); the lateral terms are
 * computed by one kernel and enter the epilogue of the fused update (split kernel on row-padded H, generic kernels) or,
 * for the other families, one update kernel behind the unfused gradient.  Other modes: H is C-contiguous with the mode's
 * shift shape; the library pads it (work arrays of its own), runs the 'valid' kernels and applies fold + update in one
 * kernel.  R_scratch: device buffer [N,C,*D] or NULL.  Volumes (ndim == 3): the same step on the volume kernels -- three
 * passes of the 1-D convolution and one kernel for the lateral terms, pad / fold for the padded modes, one update kernel. */
int tnmf_hip_update_H_ex(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom, int mode, const void *V, const void *W,
                         void *H_inout, void *R_scratch, double eps, double sparsity, double inhibition,
                         double cross_inhibition, const double *kernel0, int len0, const double *kernel1, int len1,
                         const double *kernel2, int len2, void *stream);

/* Local part of TransformInvariantNMF._update_W (TransformInvariantNMF.py:240-241 / :444-448):
 *   negpos[0] = neg_W, negpos[1] = pos_W as one contiguous [2,M,C,*A] buffer (what the all-reduce carries).
 *   R_scratch / r_is_valid as for tnmf_hip_update_H. */
int tnmf_hip_grad_W_fused(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom, const void *V, const void *W, const void *H,
                          void *R_scratch, int r_is_valid, void *negpos, void *stream);

/* ---- mini-batch schedules in one call ---------------------------------------------------------------------------------
 * The stochastic schedules of the reference (ASG / GSG / ASAG / GSAG, TransformInvariantNMF.py:467-504, and Cyclic-MU,
 * :457-465) are chains of small dependent steps -- with batch_size 3, an H half step on 3 samples, a W gradient on the
 * same 3, a W update, 256 times per epoch -- whose cost driven batch by batch from the host is launch latency and
 * interpreter time.  tnmf_hip_run_schedule takes a whole epoch as a list of operations on sample ranges of the resident
 * problem and issues every kernel from one host call:
 *   TNMF_OP_UPDATE_H  H[n0:n1] *= corr(W, V) / (corr(W, R) + eps + sparsity)            (= tnmf_hip_update_H on the slice)
 *   TNMF_OP_GRAD_W    acc = a * acc + b * [neg | pos](V, H)[n0:n1]    (a == 0: acc = b * g; _accumulate_gradient_W,
 *                     :444-455, with (a, b) = (1, 1), (1 - lambda, lambda) or, from the integer start (0, 0), (0, lambda))
 *   TNMF_OP_APPLY_W   W = W * acc_neg / (acc_pos + eps), normalised; acc_pos is left incremented by eps (:232)
 * geom->N = samples of the resident problem (V, H_inout point at sample 0); acc: device buffer [2,M,C,*A] that persists
 * between calls where the schedule says so (ASAG / GSAG).  Single device: with several ranks the gradient must be summed
 * across them between GRAD_W and APPLY_W, which is the caller's collective.
 * A RUN of consecutive TNMF_OP_UPDATE_H operations on pairwise disjoint sample ranges is executed as the H half step of
 * their union (ranges sorted and joined where they touch): such steps commute -- the H update of a sample reads that sample
 * and W only, and W does not change inside the run -- so GSG-MU / GSAG-MU (:474-479, :493-504: H for every shuffled batch, W
 * from the last batch) cost one H half step over all samples per epoch, not one launch chain per batch. */
enum { TNMF_OP_UPDATE_H = 0, TNMF_OP_GRAD_W = 1, TNMF_OP_APPLY_W = 2 };
typedef struct {
    int kind;    /* TNMF_OP_* */
    int n0, n1;  /* sample range [n0, n1) of the resident problem (may be empty) */
    double a, b; /* TNMF_OP_GRAD_W */
} tnmf_hip_op;
int tnmf_hip_run_schedule(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom, const void *V, void *W_inout, void *H_inout,
                          void *R_scratch, void *acc, const tnmf_hip_op *ops, int n_ops, double eps, double sparsity,
                          void *stream);

/* acc = a * acc + b * g over n_elems elements (a == 0: acc = b * g, whatever acc held): the blend of
 * _accumulate_gradient_W (TransformInvariantNMF.py:444-455) for callers that drive the schedules step by step (several
 * ranks: the collective sits between the gradient and this). */
int tnmf_hip_axpby(tnmf_hip_ctx *ctx, int dtype, void *acc, const void *g, double a, double b, size_t n_elems,
                   void *stream);

/* Deterministic cross-rank reduction of the [neg | pos] buffer ("all-gather ... then sum in rank order"):
 *   out[i] = ((parts[0][i] + parts[1][i]) + parts[2][i]) + ...   for the n_parts buffers of n_elems elements that the
 * caller gathered one behind the other (rank order), in the element type.  Every rank that runs it on the same gathered
 * buffer gets the same bits, whatever protocol the collective library would have picked for an all-reduce.  The gather
 * itself is the caller's (torch.distributed.all_gather_into_tensor over RCCL in tnmf_amd/backends/HIP.py). */
int tnmf_hip_sum_parts(tnmf_hip_ctx *ctx, int dtype, const void *parts, int n_parts, size_t n_elems, void *out,
                       void *stream);

/* Rest of _update_W (TransformInvariantNMF.py:232-238,244): W = W * neg / (pos + eps); W /= sum over atom axes.
 * `pos` is left incremented by eps, like the reference's in-place `pos += eps`. */
int tnmf_hip_apply_W(tnmf_hip_ctx *ctx, const tnmf_hip_geom *geom, void *W_inout, void *negpos, double eps,
                     void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TNMF_HIP_H */
