// fft.hip -- host side of the FFT kernel family: transform-length choice, workspace layout and the kernel sequences of
// the three primitives (see fft.h).  The kernels live in fft_kernels.h / fft_len.hip.
//
// 'valid' mode in the frequency domain (circular transforms of length L >= H = D + A - 1 per shift axis, so nothing
// that is kept ever wraps):
//   reconstruct  R   = crop_[A-1, A-1+D) ( ifft( sum_m H^ W^ ) )                       NumPy.py:122-132
//   H gradient   neg = crop_[0, H)       ( ifft( sum_c V^ Wf^ ) ),  Wf = W flipped     NumPy.py:93-120
//   W gradient   neg[a] = c[A-1-a],  c = crop_[0, A) ( ifft( sum_n H^ conj(V^) ) )     NumPy.py:69-91
// pos is the same with R in place of V.  1/(Ly*Lx) is folded into the W spectra (and into the partial-sum kernel of
// the W gradient), so no separate scaling pass exists.
#include <algorithm>
#include <cstdlib>

#include "fft.h"
#include "fft_engine.h"

namespace {

// Transform lengths.  Along x (row transforms) two more lengths exist: 270 and 540 store 136 / 271 frequencies = 8.5 / 17
// segments of 16 where 288 / 576 store 145 / 289 = 9.06 / 18.06 -- one whole 128-byte segment per spectrum row less for
// the shift widths of the BASELINE geometries (267, 527): a tenth less of every stream of spectra and of every
// contraction's work.  Along y (column transforms) 270 shortens the full spectra by 6 %; its column kernels run 480
// threads (4320 tile elements: 9 per thread).  540 stays out of the y list: its 16-column H-gradient tile would be 18
// elements per thread at 480 threads -- measured 41.8 ms against 28.7 ms for the pure-FFT H update of the config-5 shard.
// That kernel only runs under TNMF_PATH_FFT: under every other dispatch (AUTO / HYBRID: the H update is a direct kernel)
// the column direction takes 540 as well (`tall`): the full spectra of H at the config-5 shard (527 shifts) are 6 %
// shorter -- they are written once and read three times per iteration -- and so are the contractions that stream them.
const int kLensY[] = {32, 48, 64, 96, 144, 192, 270, 288, 384, 576};
const int kLensX[] = {32, 48, 64, 96, 144, 192, 270, 288, 384, 540, 576};
constexpr int kMixMaxGroups = 128;   // partial-sum slots of the mixed W-gradient kernel (Gn / Gp)

int pick_len(int h, int dtype, bool along_x, bool tall = false) {
    if (along_x || tall) {
        for (int L : kLensX)
            if (L >= h && (dtype == 0 || L <= 288)) return L;
        return 0;
    }
    for (int L : kLensY)
        if (L >= h && (dtype == 0 || L <= 288)) return L;
    return 0;
}

// the column direction may take the x-only lengths whenever the FFT family's own H-gradient kernel cannot be asked for
bool tall_columns(const tnmf_hip_ctx *ctx) { return ctx->path != TNMF_PATH_FFT; }

fft_run_fn lookup(int L) {
    switch (L) {
        case 32: return fft_run_32;
        case 48: return fft_run_48;
        case 64: return fft_run_64;
        case 96: return fft_run_96;
        case 144: return fft_run_144;
        case 192: return fft_run_192;
        case 270: return fft_run_270;
        case 288: return fft_run_288;
        case 384: return fft_run_384;
        case 540: return fft_run_540;
        case 576: return fft_run_576;
        default: return nullptr;
    }
}

bool use_mixed(const Geo &g, int dtype, bool grad_W);

// workspace layout, in bytes.  Sizes and offsets follow the RESIDENT problem (`gfull`: the binding, or the call itself);
// the splits of the sample / atom loops follow the samples of the call (`g`, a mini-batch slice of it or the same).
struct Lay {
    int Ly, Lx, KX, KXP, ngroups, nper, chunk, mgroups, mper;
    fft_run_fn rowf, colf;
    size_t csz;  // bytes of one complex element
    size_t T, SH, Tn, Tp, SV, SR, Ts, VT, RT, SW, SWf, TW, TWr, Wt, Gn, Gp, Gs, Wo, total, total_no_window;
    size_t sT, sSH, sS, sD;   // bytes per sample of T, SH, SV | SR, Ts | VT | RT
    bool resident;   // full spectra of H are kept (SH): the contractions stream them instead of transforming tiles
};

// split of the sample sum of the column-transform W-gradient kernel into groups
void sample_groups(const Geo &g, int tiles, int *ngroups, int *nper) {
    int ng = cdiv(2048, g.M * tiles);
    if (ng > 16) ng = 16;
    if (ng > g.N) ng = g.N;
    if (ng < 1) ng = 1;
    *nper = cdiv(g.N > 0 ? g.N : 1, ng);
    *ngroups = cdiv(g.N > 0 ? g.N : 1, *nper);
}

bool make_layout(const tnmf_hip_ctx *ctx, const Geo &gfull, const Geo &g, int dtype, Lay *l) {
    l->Ly = pick_len(g.Hy, dtype, false, tall_columns(ctx));
    l->Lx = pick_len(g.Hx, dtype, true);
    if (!l->Ly || !l->Lx) return false;
    l->rowf = lookup(l->Lx);
    l->colf = lookup(l->Ly);
    l->KX = l->Lx / 2 + 1;
    l->KXP = (int)align_up((size_t)l->KX, 16);
    l->csz = dtype == 0 ? 8 : 16;
    const int tiles = cdiv(l->KX, l->Ly > 384 ? 8 : 16);   // LenCfg<Ly>::col_tile of fft_kernels.h
    int ngroups_full, nper_full;
    sample_groups(gfull, tiles, &ngroups_full, &nper_full);
    sample_groups(g, tiles, &l->ngroups, &l->nper);
    const size_t c = l->csz, kxp = (size_t)l->KXP;
    const size_t Nf = gfull.N > 0 ? gfull.N : 1;
    l->sT = (size_t)g.M * g.Hy * kxp * c;
    l->sSH = (size_t)g.M * l->Ly * kxp * c;
    l->sS = (size_t)g.C * l->Ly * kxp * c;
    l->sD = (size_t)g.C * g.Dy * kxp * c;
    // H half step in windows of `chunk` samples, which bounds the neg/pos row spectra (written by the column kernel, read
    // back by the row kernel right after) to 8 GB.  Measured at config 3: windows small enough for the 256 MiB Infinity
    // Cache are slower (launch tails: 24 MB 11.3 ms, 96 MB 7.3 ms, unwindowed 6.1 ms), so the window is as large as
    // the budget allows.
    size_t budget = (size_t)8 << 30;
    if (const char *e = tnmf_diag_env("TNMF_FFT_WINDOW_MB")) budget = (size_t)atol(e) << 20;
    long chunk_full = (long)(budget / (2 * l->sT));
    if (chunk_full < 1) chunk_full = 1;
    if (chunk_full > (long)Nf) chunk_full = (long)Nf;
    l->chunk = (int)(chunk_full > g.N && g.N > 0 ? g.N : chunk_full);
    int mg = cdiv(1024, l->chunk * cdiv(l->KX, 16) * 2);   // the H-gradient kernel always takes 16-column tiles
    if (mg > g.M) mg = g.M;
    if (mg < 1) mg = 1;
    l->mper = cdiv(g.M, mg);
    l->mgroups = cdiv(g.M, l->mper);
    const size_t nSW = (size_t)g.M * g.C * l->Ly * kxp * c;
    size_t o = 0;
    auto take = [&o](size_t bytes) {
        const size_t at = o;
        o += align_up(bytes, 256);
        return at;
    };
    // (+ 64 slack rows: k_mix_reconstruct reads up to S + Ay - 1 rows past the last plane under outputs it drops)
    l->T = take(Nf * l->sT + (size_t)64 * kxp * c);
    // Problems the mixed kernels do not cover (several channels, tall atoms) keep the full spectra of H resident: one
    // column-transform pass per H, then every contraction is a plain streaming kernel (fft_spectral.hip)
    l->resident = !(use_mixed(g, dtype, false) && use_mixed(g, dtype, true));
    l->SH = take(l->resident ? Nf * l->sSH : 0);
    l->SV = take(Nf * l->sS);
    l->SR = take(Nf * l->sS);
    l->Ts = take(Nf * l->sD);
    l->VT = take(Nf * l->sD);   // row spectra of V (kept) and of R
    l->RT = take(Nf * l->sD);
    l->SW = take(nSW);
    l->SWf = take(nSW);
    l->TW = take((size_t)2 * g.M * g.C * g.Ay * kxp * c);
    l->TWr = take((size_t)g.M * g.C * g.Ay * kxp * c);   // row spectra of W for the mixed reconstruct (kept: FftState::W_ok)
    l->Wt = take((size_t)2 * g.M * g.C * g.Ay * g.Ax * (c / 2));
    // partial W-gradient spectra: [groups][M*C][Ly][KXP] for the column-transform kernel, [<= kMixMaxGroups groups]
    // [M*C][Ay][KXP] for the mixed kernel
    const size_t nG = std::max(nSW * ngroups_full, (size_t)kMixMaxGroups * g.M * g.C * g.Ay * kxp * c);
    l->Gn = take(nG);
    l->Gp = take(nG);
    l->Gs = take(nSW * 2);
    l->Wo = take((size_t)2 * g.M * g.C * g.Ay * g.Ax * (c / 2));
    l->total_no_window = o;
    const size_t nTc = (size_t)chunk_full * l->sT;
    l->Tn = take(nTc);   // the window buffers of the H half step come last: callers that never run it (the hybrid
    l->Tp = take(nTc);   // dispatch) do not pay for them
    l->total = o;
    return true;
}

void clear_flags(FftState &f) {
    f.W_ok = false;
    std::fill(f.T_ok.begin(), f.T_ok.end(), 0);
    std::fill(f.SH_ok.begin(), f.SH_ok.end(), 0);
    std::fill(f.V_ok.begin(), f.V_ok.end(), 0);
    std::fill(f.SV_ok.begin(), f.SV_ok.end(), 0);
}

int ensure_ws(tnmf_hip_ctx *ctx, size_t bytes) {
    FftState &f = ctx->fft;
    if (bytes <= f.ws_bytes) return TNMF_OK;
    // A request at least as large as one that already failed is refused at once: under TNMF_PATH_AUTO the callers fall
    // back to the direct kernels, and must not pay a multi-GB hipMalloc attempt on every call.
    if (f.failed_bytes && bytes >= f.failed_bytes) return TNMF_E_WORKSPACE;
    // the larger buffer is allocated BEFORE the current one is released, so a failure leaves the working one in place;
    // when old + new do not fit side by side the old order (release, then allocate) is tried once, and only if that
    // fails too is the size remembered as refused (until tnmf_hip_ctx_reserve: the caller may have freed memory since)
    const size_t want = align_up(bytes, 1 << 20);
    void *bigger = nullptr;
    if (hipMalloc(&bigger, want) != hipSuccess) {
        (void)hipGetLastError();
        bigger = nullptr;
        if (!f.ws) {
            f.failed_bytes = bytes;
            return TNMF_E_WORKSPACE;
        }
    }
    if (f.ws) {
        TNMF_HIP_TRY(hipDeviceSynchronize());
        TNMF_HIP_TRY(hipFree(f.ws));
        f.ws = nullptr;
        f.ws_bytes = 0;
        clear_flags(f);
    }
    if (!bigger && hipMalloc(&bigger, want) != hipSuccess) {
        (void)hipGetLastError();
        f.failed_bytes = bytes;
        return TNMF_E_WORKSPACE;
    }
    f.ws = bigger;
    f.ws_bytes = want;
    clear_flags(f);
    return TNMF_OK;
}

inline char *at(tnmf_hip_ctx *ctx, size_t off) { return static_cast<char *>(ctx->fft.ws) + off; }

#define CHECK(rc_expr)                  \
    do {                                \
        const int _rc = (rc_expr);      \
        if (_rc != TNMF_OK) return _rc; \
    } while (0)

// out0 = scale * W, out1 = scale * W flipped along both atom axes
template <typename T>
__global__ void k_fft_prep_W(const T *W, T *out0, T *out1, int planes, int Ay, int Ax, double scale) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, per = Ay * Ax;
    if (i >= planes * per) return;
    const int p = i / per, r = i - p * per, ay = r / Ax, ax = r - ay * Ax;
    const T v = (T)(scale * (double)W[i]);
    out0[i] = v;
    out1[(long)p * per + (Ay - 1 - ay) * Ax + (Ax - 1 - ax)] = v;
}

// out = scale * sum over groups, in group order (deterministic); blockIdx.y picks the gradient (neg / pos).  The loads of
// four groups are issued together, the additions keep the group order.
template <typename T>
__global__ __launch_bounds__(64) void k_fft_sum_groups(const cplx<T> *parts0, const cplx<T> *parts1, cplx<T> *out0,
                                                      cplx<T> *out1, long count, int ngroups, double scale) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const cplx<T> *parts = blockIdx.y ? parts1 : parts0;
    double re = 0, im = 0;
    int gidx = 0;
    for (; gidx + 4 <= ngroups; gidx += 4) {
        cplx<T> v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = parts[(gidx + k) * count + i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            re += (double)v[k].x;
            im += (double)v[k].y;
        }
    }
    for (; gidx < ngroups; ++gidx) {
        const cplx<T> v = parts[gidx * count + i];
        re += (double)v.x;
        im += (double)v.y;
    }
    (blockIdx.y ? out1 : out0)[i] = {(T)(re * scale), (T)(im * scale)};
}

// neg/pos[m,c,a] = corr[m,c,A-1-a]  (the flip of NumPy.py:85,90)
template <typename T>
__global__ void k_fft_flip_out(const T *in, T *out, int planes, int Ay, int Ax, int clamp0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, per = Ay * Ax;
    if (i >= planes * per) return;
    const int p = i / per, r = i - p * per, ay = r / Ax, ax = r - ay * Ax;
    T v = in[(long)p * per + (Ay - 1 - ay) * Ax + (Ax - 1 - ax)];
    if (clamp0) v = v < (T)0 ? (T)0 : v;
    out[i] = v;
}

FftArgs base_args(const Geo &g, const Lay &l) {
    FftArgs a = {};
    a.KX = l.KX;
    a.KXP = l.KXP;
    a.N = g.N;
    a.M = g.M;
    a.C = g.C;
    a.Hy = g.Hy;
    return a;
}

// real planes [planes][rows][cols] (contiguous) -> full spectra
int forward_planes(const Geo &g, const Lay &l, int dtype, const void *src, int planes, int rows, int cols, void *Ttmp,
                   void *S, hipStream_t s) {
    FftArgs a = base_args(g, l);
    a.src0 = src;
    a.dst0 = Ttmp;
    a.planes = planes;
    a.rows = rows;
    a.cols = cols;
    a.ld_src = cols;
    a.ps_src = (long)rows * cols;
    a.ps_dst = (long)rows * l.KXP;
    CHECK(l.rowf(kFftRowsFwd, dtype, &a, s));
    FftArgs b = base_args(g, l);
    b.src0 = Ttmp;
    b.dst0 = S;
    b.planes = planes;
    b.rows = rows;
    return l.colf(kFftColsFwd, dtype, &b, s);
}

bool use_mixed(const Geo &g, int dtype, bool grad_W) {
    static const bool off = tnmf_diag_env("TNMF_FFT_NO_MIXED") != nullptr;   // diagnostic: force the column-transform kernels
    return !off && (grad_W ? mixed_has_grad_W(g, dtype) : mixed_has_reconstruct(g, dtype));
}

bool use_resident(const Lay &l) {
    static const bool off = tnmf_diag_env("TNMF_FFT_NO_RESIDENT") != nullptr;   // diagnostic: contract inside the column kernels
    return l.resident && !off;
}

// Wt <- scale * W (and its flipped copy behind it)
int scaled_W(tnmf_hip_ctx *ctx, const Geo &g, const Lay &l, int dtype, const void *W, double scale, hipStream_t s) {
    const int planes = g.M * g.C, n = planes * g.Ay * g.Ax;
    char *wt = at(ctx, l.Wt);
    const size_t half = (size_t)n * (l.csz / 2);
    if (dtype == 0)
        hipLaunchKernelGGL(k_fft_prep_W<float>, dim3(cdiv(n, 256)), dim3(256), 0, s, (const float *)W, (float *)wt,
                           (float *)(wt + half), planes, g.Ay, g.Ax, scale);
    else
        hipLaunchKernelGGL(k_fft_prep_W<double>, dim3(cdiv(n, 256)), dim3(256), 0, s, (const double *)W, (double *)wt,
                           (double *)(wt + half), planes, g.Ay, g.Ax, scale);
    TNMF_LAUNCH_CHECK();
    return TNMF_OK;
}

// W spectra (plain into SW, flipped into SWf), both scaled by 1/(Ly*Lx)
int spectra_W(tnmf_hip_ctx *ctx, const Geo &g, const Lay &l, int dtype, const void *W, bool plain, bool flipped,
              hipStream_t s) {
    const int planes = g.M * g.C, per = g.Ay * g.Ax, n = planes * per;
    const double scale = 1.0 / ((double)l.Ly * l.Lx);
    char *wt = at(ctx, l.Wt);
    const size_t half = (size_t)n * (l.csz / 2);
    if (dtype == 0)
        hipLaunchKernelGGL(k_fft_prep_W<float>, dim3(cdiv(n, 256)), dim3(256), 0, s, (const float *)W, (float *)wt,
                           (float *)(wt + half), planes, g.Ay, g.Ax, scale);
    else
        hipLaunchKernelGGL(k_fft_prep_W<double>, dim3(cdiv(n, 256)), dim3(256), 0, s, (const double *)W, (double *)wt,
                           (double *)(wt + half), planes, g.Ay, g.Ax, scale);
    TNMF_LAUNCH_CHECK();
    if (plain) CHECK(forward_planes(g, l, dtype, wt, planes, g.Ay, g.Ax, at(ctx, l.TW), at(ctx, l.SW), s));
    if (flipped) CHECK(forward_planes(g, l, dtype, wt + half, planes, g.Ay, g.Ax, at(ctx, l.TW), at(ctx, l.SWf), s));
    return TNMF_OK;
}

// ---- binding and per-sample validity ---------------------------------------------------------------------------------

inline size_t esz_of(int dtype) { return dtype == 0 ? 4 : 8; }

bool same_shape(const Geo &a, const Geo &b) {
    return a.M == b.M && a.C == b.C && a.Dy == b.Dy && a.Dx == b.Dx && a.Ay == b.Ay && a.Ax == b.Ax && a.Hs == b.Hs;
}

void bind(FftState &f, const Geo &g, int dtype, const void *H, const void *V, bool explicit_bind) {
    f.bound = true;
    f.explicit_bind = explicit_bind;
    f.H_base = H;
    f.V_base = V;
    f.geo = g;
    f.dtype = dtype;
    const size_t n = g.N > 0 ? (size_t)g.N : 0;
    f.T_ok.assign(n, 0);
    f.SH_ok.assign(n, 0);
    f.V_ok.assign(n, 0);
    f.SV_ok.assign(n, 0);
}

// where the samples of a call sit in the per-sample arrays of the workspace
struct Slot {
    int n0;          // first sample of the call within the binding (0: the call is not a slice of the bound problem)
    bool cached;     // H of the call is a slice of the bound activations: validity flags apply and results are recorded
    bool v_cached;   // likewise V
};

Slot locate(FftState &f, const Geo &g, int dtype, const void *H, const void *V) {
    Slot sl = {0, false, false};
    if (!f.cache_enabled || !f.bound || dtype != f.dtype || !same_shape(f.geo, g) || g.N <= 0) return sl;
    const size_t es = esz_of(dtype);
    const size_t hb = (size_t)g.M * g.Hy * g.Hs * es, vb = (size_t)g.C * g.Dy * g.Dx * es;
    long n0 = -1;
    if (H) {
        const ptrdiff_t d = static_cast<const char *>(H) - static_cast<const char *>(f.H_base);
        if (!f.H_base || d < 0 || (size_t)d % hb) return sl;
        n0 = (long)((size_t)d / hb);
    } else if (V && f.V_base) {   // (the H gradient takes no H: the samples locate the call)
        const ptrdiff_t d = static_cast<const char *>(V) - static_cast<const char *>(f.V_base);
        if (d < 0 || (size_t)d % vb) return sl;
        n0 = (long)((size_t)d / vb);
    }
    if (n0 < 0 || n0 + g.N > f.geo.N) return sl;
    sl.n0 = (int)n0;
    sl.cached = true;
    if (V) {
        if (!f.V_base) {   // binding taken from a call that had no V: adopt the samples now
            f.V_base = static_cast<const char *>(V) - (size_t)n0 * vb;
            std::fill(f.V_ok.begin(), f.V_ok.end(), 0);
            std::fill(f.SV_ok.begin(), f.SV_ok.end(), 0);
        }
        sl.v_cached = static_cast<const char *>(V) == static_cast<const char *>(f.V_base) + (size_t)n0 * vb;
    }
    return sl;
}

bool all_ok(const std::vector<unsigned char> &v, int n0, int n) {
    for (int i = n0; i < n0 + n; ++i)
        if (!v[i]) return false;
    return true;
}
void set_ok(std::vector<unsigned char> &v, int n0, int n, unsigned char val) {
    for (int i = n0; i < n0 + n && i < (int)v.size(); ++i) v[i] = val;
}

// per-call view of the workspace: layout + the pointers of the call's sample range
struct Call {
    Lay l;
    Slot sl;
    char *T, *SH, *SV, *SR, *Ts, *VT, *RT;
};

int prepare(tnmf_hip_ctx *ctx, const Geo &g, int dtype, const void *H, const void *V, Call *c, bool window = false) {
    if (!fft_has(g, dtype)) return TNMF_E_UNSUPPORTED;
    FftState &f = ctx->fft;
    c->sl = locate(f, g, dtype, H, V);
    if (!c->sl.cached) {
        if (f.cache_enabled && !f.explicit_bind && (H || V)) {
            // no binding told by the caller: the operands of this call become the resident problem (the spectra of the
            // activations transformed or updated last are reused while the same pointers and geometry come back)
            bind(f, g, dtype, H, V, false);
            c->sl = {0, H != nullptr, V != nullptr};
        } else {
            clear_flags(f);   // a foreign problem passes through the workspace: nothing of the binding survives it
        }
    }
    const bool in_binding = c->sl.cached || c->sl.v_cached;
    if (!make_layout(ctx, in_binding ? f.geo : g, g, dtype, &c->l)) return TNMF_E_UNSUPPORTED;
    CHECK(ensure_ws(ctx, window ? c->l.total : c->l.total_no_window));
    const Lay &l = c->l;
    const size_t n0 = (size_t)c->sl.n0;
    c->T = at(ctx, l.T) + n0 * l.sT;
    c->SH = at(ctx, l.SH) + n0 * l.sSH;
    c->SV = at(ctx, l.SV) + n0 * l.sS;
    c->SR = at(ctx, l.SR) + n0 * l.sS;
    c->Ts = at(ctx, l.Ts) + n0 * l.sD;
    c->VT = at(ctx, l.VT) + n0 * l.sD;
    c->RT = at(ctx, l.RT) + n0 * l.sD;
    ctx->last_path = "fft";
    return TNMF_OK;
}

// rows only: real planes [planes][rows][cols] -> row spectra
int forward_rows(const Geo &g, const Lay &l, int dtype, const void *src, int planes, int rows, int cols, void *Tdst,
                 hipStream_t s) {
    FftArgs a = base_args(g, l);
    a.src0 = src;
    a.dst0 = Tdst;
    a.planes = planes;
    a.rows = rows;
    a.cols = cols;
    a.ld_src = cols;
    a.ps_src = (long)rows * cols;
    a.ps_dst = (long)rows * l.KXP;
    return l.rowf(kFftRowsFwd, dtype, &a, s);
}

int columns_of(const Geo &g, const Lay &l, int dtype, const void *Tsrc, int planes, int rows, void *S, hipStream_t s) {
    FftArgs b = base_args(g, l);
    b.src0 = Tsrc;
    b.dst0 = S;
    b.planes = planes;
    b.rows = rows;
    return l.colf(kFftColsFwd, dtype, &b, s);
}

// spectra of the samples: row spectra into VT always, full spectra into SV on demand (both skipped for the samples whose
// spectra the cache holds: V never changes during a fit)
int spectra_V(tnmf_hip_ctx *ctx, const Geo &g, const Call &c, int dtype, const void *V, bool full, hipStream_t s) {
    FftState &f = ctx->fft;
    const Lay &l = c.l;
    const bool track = c.sl.v_cached && f.cache_enabled;
    const bool hit = track && all_ok(f.V_ok, c.sl.n0, g.N);
    ++(hit ? f.v_hits : f.v_runs);
    if (!hit) {
        if (track) {
            set_ok(f.V_ok, c.sl.n0, g.N, 0);
            set_ok(f.SV_ok, c.sl.n0, g.N, 0);
        }
        CHECK(forward_rows(g, l, dtype, V, g.N * g.C, g.Dy, g.Dx, c.VT, s));
        if (track) set_ok(f.V_ok, c.sl.n0, g.N, 1);
    }
    if (full && !(hit && all_ok(f.SV_ok, c.sl.n0, g.N))) {
        CHECK(columns_of(g, l, dtype, c.VT, g.N * g.C, g.Dy, c.SV, s));
        if (track) set_ok(f.SV_ok, c.sl.n0, g.N, 1);
    }
    return TNMF_OK;
}

// row spectra of H into the workspace (skipped when the cache holds them for every sample of the call)
int rows_of_H(tnmf_hip_ctx *ctx, const Geo &g, const Call &c, int dtype, const void *H, hipStream_t s) {
    FftState &f = ctx->fft;
    const bool track = c.sl.cached && f.cache_enabled;
    if (track && all_ok(f.T_ok, c.sl.n0, g.N)) {
        ++f.h_hits;
        return TNMF_OK;
    }
    ++f.h_runs;
    if (track) {
        set_ok(f.T_ok, c.sl.n0, g.N, 0);
        set_ok(f.SH_ok, c.sl.n0, g.N, 0);
    }
    FftArgs a = base_args(g, c.l);
    a.src0 = H;
    a.dst0 = c.T;
    a.planes = g.N * g.M;
    a.rows = g.Hy;
    a.cols = g.Hx;
    a.ld_src = g.Hs;   // rows of H may be padded (tnmf_hip_geom.h_row_stride)
    a.ps_src = (long)g.Hy * g.Hs;
    a.ps_dst = (long)g.Hy * c.l.KXP;
    CHECK(c.l.rowf(kFftRowsFwd, dtype, &a, s));
    if (track) set_ok(f.T_ok, c.sl.n0, g.N, 1);
    return TNMF_OK;
}

// full spectra of H (column transforms of its row spectra) into SH, once per H
int spectra_of_H(tnmf_hip_ctx *ctx, const Geo &g, const Call &c, int dtype, const void *H, hipStream_t s) {
    FftState &f = ctx->fft;
    CHECK(rows_of_H(ctx, g, c, dtype, H, s));
    const bool track = c.sl.cached && f.cache_enabled;
    if (track && all_ok(f.SH_ok, c.sl.n0, g.N)) return TNMF_OK;
    CHECK(columns_of(g, c.l, dtype, c.T, g.N * g.M, g.Hy, c.SH, s));
    if (track) set_ok(f.SH_ok, c.sl.n0, g.N, 1);   // (only as good as the row spectra they came from: cleared with them)
    return TNMF_OK;
}

}  // namespace

bool fft_has(const Geo &g, int dtype) {
    if (g.Dy == 1 && g.Ay == 1) {
        // 1-D signals: the row-transform half of the family alone (float32, up to three channels): reconstruct and the W
        // gradient are pointwise products of row spectra; the H update stays on the direct kernels (fft_grad_H /
        // fft_update_H refuse 1-D problems)
        return pick_len(g.Hx, dtype, true) != 0 && mixed_has_reconstruct(g, dtype) && mixed_has_grad_W(g, dtype);
    }
    return pick_len(g.Hy, dtype, false) != 0 && pick_len(g.Hx, dtype, true) != 0;
}

void fft_bind(tnmf_hip_ctx *ctx, const Geo &g, int dtype, const void *H, const void *V) {
    bind(ctx->fft, g, dtype, H, V, true);
}

void fft_unbind(tnmf_hip_ctx *ctx) {
    FftState &f = ctx->fft;
    f.bound = f.explicit_bind = false;
    f.H_base = f.V_base = nullptr;
    f.T_ok.clear();
    f.SH_ok.clear();
    f.V_ok.clear();
    f.SV_ok.clear();
}

void fft_invalidate_H(tnmf_hip_ctx *ctx, const Geo &g, int dtype, const void *H) {
    FftState &f = ctx->fft;
    const Slot sl = locate(f, g, dtype, H, nullptr);
    if (sl.cached) {
        set_ok(f.T_ok, sl.n0, g.N, 0);
        set_ok(f.SH_ok, sl.n0, g.N, 0);
    } else {
        std::fill(f.T_ok.begin(), f.T_ok.end(), 0);
        std::fill(f.SH_ok.begin(), f.SH_ok.end(), 0);
    }
}

void fft_invalidate(tnmf_hip_ctx *ctx) { clear_flags(ctx->fft); }

void fft_invalidate_W(tnmf_hip_ctx *ctx) { ctx->fft.W_ok = false; }

int fft_reserve(tnmf_hip_ctx *ctx, const Geo &g, int dtype, bool with_window) {
    Lay l;
    ctx->fft.failed_bytes = 0;   // an explicit request: try again even if this size was refused before
    const FftState &f = ctx->fft;
    const bool in_binding = f.bound && f.dtype == dtype && same_shape(f.geo, g) && f.geo.N >= g.N;
    if (!fft_has(g, dtype) || !make_layout(ctx, in_binding ? f.geo : g, g, dtype, &l)) return TNMF_E_UNSUPPORTED;
    return ensure_ws(ctx, with_window ? l.total : l.total_no_window);
}

void fft_release(tnmf_hip_ctx *ctx) {
    if (ctx->fft.ws) (void)hipFree(ctx->fft.ws);
    ctx->fft.ws = nullptr;
    ctx->fft.ws_bytes = 0;
    ctx->fft.failed_bytes = 0;
    clear_flags(ctx->fft);
}

int fft_reconstruct(tnmf_hip_ctx *ctx, const Geo &g, int dtype, const void *W, const void *H, void *R, bool nonneg,
                    hipStream_t s) {
    Call c;
    CHECK(prepare(ctx, g, dtype, H, nullptr, &c));
    const Lay &l = c.l;
    CHECK(rows_of_H(ctx, g, c, dtype, H, s));
    FftState &f = ctx->fft;
    // the spectra of W this primitive needs are kept while W stays the same (mini-batch schedules: one W per epoch)
    const bool w_hit = f.cache_enabled && f.W_ok && f.W_owner == W && f.W_dtype == dtype && same_shape(f.W_geo, g) &&
                       (c.sl.cached || f.W_geo.N == g.N);   // (same layout: a slice of the binding, or the same call shape)
    auto w_mark = [&]() {
        f.W_ok = f.cache_enabled;
        f.W_owner = W;
        f.W_geo = g;
        f.W_dtype = dtype;
    };
    if (use_mixed(g, dtype, false)) {
        // transform along x only; the atom rows are contracted directly (fft_mixed.hip)
        if (!w_hit) {
            f.W_ok = false;
            CHECK(scaled_W(ctx, g, l, dtype, W, 1.0 / l.Lx, s));
            CHECK(forward_rows(g, l, dtype, at(ctx, l.Wt), g.M * g.C, g.Ay, g.Ax, at(ctx, l.TWr), s));
            w_mark();
        }
        CHECK(mixed_reconstruct(g, c.T, at(ctx, l.TWr), c.Ts, l.KX, l.KXP, s));
    } else {
        if (!w_hit) {
            f.W_ok = false;
            CHECK(spectra_W(ctx, g, l, dtype, W, true, false, s));
            w_mark();
        }
        if (use_resident(l)) {
            CHECK(spectra_of_H(ctx, g, c, dtype, H, s));
            CHECK(spectral_contract_R(g, dtype, c.SH, at(ctx, l.SW), c.SR, l.Ly, l.KX, l.KXP, s));
        } else {
            FftArgs a = base_args(g, l);
            a.src0 = c.T;
            a.src1 = at(ctx, l.SW);
            a.dst0 = c.SR;
            CHECK(l.colf(kFftContractR, dtype, &a, s));
        }
        FftArgs b = base_args(g, l);
        b.src0 = c.SR;
        b.dst0 = c.Ts;
        b.planes = g.N * g.C;
        b.rows = g.Dy;
        b.yoff = g.Ay - 1;
        CHECK(l.colf(kFftColsInv, dtype, &b, s));
    }
    FftArgs d = base_args(g, l);
    d.src0 = c.Ts;
    d.dst0 = R;
    d.planes = g.N * g.C;
    d.rows = g.Dy;
    d.cols = g.Dx;
    d.xoff = g.Ax - 1;
    d.ld_dst = g.Dx;
    d.ps_dst = (long)g.Dy * g.Dx;
    d.clamp0 = nonneg ? 1 : 0;
    return l.rowf(kFftRowsInv, dtype, &d, s);
}

namespace {

// spectra of V and R and of the flipped W: what the H-gradient column kernel contracts
int grad_H_spectra(tnmf_hip_ctx *ctx, const Geo &g, const Call &c, int dtype, const void *V, const void *R,
                   const void *W, hipStream_t s) {
    CHECK(spectra_W(ctx, g, c.l, dtype, W, false, true, s));
    CHECK(spectra_V(ctx, g, c, dtype, V, true, s));
    return forward_planes(g, c.l, dtype, R, g.N * g.C, g.Dy, g.Dx, c.Ts, c.SR, s);
}

// neg/pos row spectra of the call's samples [n0, n0+cnt) into the window buffers
int grad_H_window(tnmf_hip_ctx *ctx, const Geo &g, const Call &c, int dtype, int n0, int cnt, hipStream_t s) {
    const Lay &l = c.l;
    FftArgs a = base_args(g, l);
    a.src0 = c.SV;
    a.src1 = c.SR;
    a.src2 = at(ctx, l.SWf);
    a.dst0 = at(ctx, l.Tn);
    a.dst1 = at(ctx, l.Tp);
    a.n0 = n0;
    a.planes = cnt;
    a.mgroups = l.mgroups;
    a.mper = l.mper;
    return l.colf(kFftGradH, dtype, &a, s);
}

}  // namespace

int fft_grad_H(tnmf_hip_ctx *ctx, const Geo &g, int dtype, const void *V, const void *R, const void *W, void *neg,
               void *pos, hipStream_t s) {
    if (g.Dy == 1 && g.Ay == 1) return TNMF_E_UNSUPPORTED;   // 1-D: row-transform half only (fft_has)
    Call c;
    CHECK(prepare(ctx, g, dtype, nullptr, V, &c, true));
    const Lay &l = c.l;
    CHECK(grad_H_spectra(ctx, g, c, dtype, V, R, W, s));
    const size_t esz = l.csz / 2, hplane = (size_t)g.M * g.Hy * g.Hx;
    for (int n0 = 0; n0 < g.N; n0 += l.chunk) {
        const int cnt = g.N - n0 < l.chunk ? g.N - n0 : l.chunk;
        CHECK(grad_H_window(ctx, g, c, dtype, n0, cnt, s));
        FftArgs a = base_args(g, l);
        a.src0 = at(ctx, l.Tn);
        a.src1 = at(ctx, l.Tp);
        a.dst0 = static_cast<char *>(neg) + (size_t)n0 * hplane * esz;
        a.dst1 = static_cast<char *>(pos) + (size_t)n0 * hplane * esz;
        a.planes = cnt * g.M;
        a.rows = g.Hy;
        a.cols = g.Hx;
        a.xoff = 0;
        a.ld_dst = g.Hx;
        a.ps_dst = (long)g.Hy * g.Hx;
        CHECK(l.rowf(kFftRowsInv2, dtype, &a, s));
    }
    return TNMF_OK;
}

int fft_update_H(tnmf_hip_ctx *ctx, const Geo &g, int dtype, const void *V, const void *R, const void *W, void *H,
                 double reg, hipStream_t s) {
    if (g.Dy == 1 && g.Ay == 1) return TNMF_E_UNSUPPORTED;   // 1-D: row-transform half only (fft_has)
    Call c;
    CHECK(prepare(ctx, g, dtype, H, V, &c, true));
    const Lay &l = c.l;
    FftState &f = ctx->fft;
    const bool track = c.sl.cached && f.cache_enabled;
    CHECK(grad_H_spectra(ctx, g, c, dtype, V, R, W, s));
    if (track) {
        set_ok(f.T_ok, c.sl.n0, g.N, 0);
        set_ok(f.SH_ok, c.sl.n0, g.N, 0);
    }
    const size_t esz = l.csz / 2, hplane = (size_t)g.M * g.Hy * g.Hs;
    for (int n0 = 0; n0 < g.N; n0 += l.chunk) {
        const int cnt = g.N - n0 < l.chunk ? g.N - n0 : l.chunk;
        CHECK(grad_H_window(ctx, g, c, dtype, n0, cnt, s));
        FftArgs a = base_args(g, l);
        a.src0 = at(ctx, l.Tn);
        a.src1 = at(ctx, l.Tp);
        a.dst0 = static_cast<char *>(H) + (size_t)n0 * hplane * esz;
        a.dst1 = c.T + (size_t)n0 * l.sT;
        a.planes = cnt * g.M;
        a.rows = g.Hy;
        a.cols = g.Hx;
        a.ld_dst = g.Hs;
        a.ps_src = (long)g.Hy * g.Hs;
        a.ps_dst = (long)g.Hy * l.KXP;
        a.reg = reg;
        CHECK(l.rowf(kFftRowsMu, dtype, &a, s));
    }
    if (track) set_ok(f.T_ok, c.sl.n0, g.N, 1);   // the kernel left the row spectra of the new H behind
    return TNMF_OK;
}

// Sample groups (= partial sums) of the mixed W gradient.  Its grid is (atom blocks) x (kx tiles) x (groups / 4) workgroups
// of ONE wave, and the register budget lets two waves share a SIMD: the grid should fill the chip's 2 * 4 * CUs wave
// slots a whole number of times -- at config 3, 64 groups gave 2560 waves for 2048 slots, i.e. a second round on a
// quarter of the chip (0.89 ms; 128 groups: 2.5 rounds of half-size waves).  More groups cost partial-sum traffic.
int mix_groups(const tnmf_hip_ctx *ctx, const Geo &g, const Lay &l) {
    const long slots = 2L * 4 * (ctx->num_cu > 0 ? ctx->num_cu : 256);
    const long per_group_block = (long)(g.C == 1 && g.Ay <= 12 ? cdiv(g.M, 2) : g.M) * cdiv(l.KX, 16);
    int best = 32;
    double best_cost = 1e30;
    for (int cand = 32; cand <= kMixMaxGroups; cand *= 2) {
        int ng = g.N < cand ? g.N : cand;
        const int nper = cdiv(g.N, ng);
        ng = cdiv(g.N, nper);
        const double rounds = (double)(per_group_block * cdiv(ng, 4)) / (double)slots;
        const double whole = rounds <= 1.0 ? 1.0 : (double)(long)(rounds + 0.999999);
        // time ~ whole rounds of (work / rounds) each; partial sums: 3 % of the kernel per 32 groups (measured at 64)
        const double cost = whole / rounds + 0.03 * (cand / 32);
        if (cost < best_cost - 1e-9) {
            best_cost = cost;
            best = cand;
        }
    }
    return best;
}

int fft_grad_W(tnmf_hip_ctx *ctx, const Geo &g, int dtype, const void *V, const void *R, const void *H, void *neg,
               void *pos, bool nonneg, hipStream_t s) {
    Call c;
    CHECK(prepare(ctx, g, dtype, H, V, &c));
    const Lay &l = c.l;
    CHECK(rows_of_H(ctx, g, c, dtype, H, s));
    const bool mixed = use_mixed(g, dtype, true);
    CHECK(spectra_V(ctx, g, c, dtype, V, !mixed, s));
    const int planes = 2 * g.M * g.C;
    if (mixed) {
        // transform along x only; the Ay lags along y are accumulated directly (fft_mixed.hip)
        CHECK(forward_rows(g, l, dtype, R, g.N * g.C, g.Dy, g.Dx, c.RT, s));
        // sample groups = partial sums (mix_groups below)
        int ng_want = mix_groups(ctx, g, l);
        if (const char *e = tnmf_diag_env("TNMF_MIX_GROUPS")) ng_want = atoi(e);   // diagnostic builds only
        ng_want = ng_want < 1 ? 1 : (ng_want > kMixMaxGroups ? kMixMaxGroups : ng_want);
        int ng = g.N < ng_want ? g.N : ng_want;
        const int nper = cdiv(g.N, ng);
        ng = cdiv(g.N, nper);
        const int ngpad = (int)align_up((size_t)ng, 4);   // whole blocks of 4 groups; the extra groups write zeros
        CHECK(mixed_grad_W(g, c.T, c.VT, c.RT, at(ctx, l.Gn), at(ctx, l.Gp), l.KX, l.KXP, ngpad, nper, s));
        const long count = (long)g.M * g.C * g.Ay * l.KXP;
        char *out = at(ctx, l.TW);   // [2*M*C][Ay][KXP]
        hipLaunchKernelGGL(k_fft_sum_groups<float>, dim3((unsigned)((count + 63) / 64), 2), dim3(64), 0, s,
                           (const cplx<float> *)at(ctx, l.Gn), (const cplx<float> *)at(ctx, l.Gp), (cplx<float> *)out,
                           (cplx<float> *)(out + (size_t)count * l.csz), count, ngpad, 1.0 / l.Lx);
        TNMF_LAUNCH_CHECK();
    } else {
        CHECK(forward_planes(g, l, dtype, R, g.N * g.C, g.Dy, g.Dx, c.Ts, c.SR, s));
        int ngroups = l.ngroups;
        if (use_resident(l)) {
            CHECK(spectra_of_H(ctx, g, c, dtype, H, s));
            // Sample groups of the streaming kernel: every group writes (and the sum reads back) a full set of partial
            // gradient spectra -- 2 x 35 MB per group at config 4, as much as the activations of 6 samples -- so only
            // as many as it takes to fill the chip about twice with (atom quads) x (f blocks) x (channel groups)
            // workgroups of four waves
            const long blocks = (long)cdiv(g.M, 4) * (((long)l.Ly * l.KXP + 255) / 256) * cdiv(g.C, 4);
            int ng = (int)((8L * ctx->num_cu + blocks - 1) / blocks);
            ng = ng < 1 ? 1 : (ng > l.ngroups ? l.ngroups : ng);
            const int nper = cdiv(g.N, ng);
            ngroups = cdiv(g.N, nper);
            CHECK(spectral_grad_W(g, dtype, c.SH, c.SV, c.SR, at(ctx, l.Gn), at(ctx, l.Gp), l.Ly, l.KX, l.KXP, ngroups,
                                  nper, s));
        } else {
            FftArgs a = base_args(g, l);
            a.src0 = c.T;
            a.src1 = c.SV;
            a.src2 = c.SR;
            a.dst0 = at(ctx, l.Gn);
            a.dst1 = at(ctx, l.Gp);
            a.ngroups = l.ngroups;
            a.nper = l.nper;
            CHECK(l.colf(kFftGradW, dtype, &a, s));
        }
        // fixed-order sum of the groups, scaled; both gradients side by side: [2][M*C][Ly][KXP]
        const long count = (long)g.M * g.C * l.Ly * l.KXP;
        const size_t sbytes = (size_t)count * l.csz;
        const double scale = 1.0 / ((double)l.Ly * l.Lx);
        {
            char *out = at(ctx, l.Gs);
            const dim3 grid((unsigned)((count + 63) / 64), 2);
            if (dtype == 0)
                hipLaunchKernelGGL(k_fft_sum_groups<float>, grid, dim3(64), 0, s, (const cplx<float> *)at(ctx, l.Gn),
                                   (const cplx<float> *)at(ctx, l.Gp), (cplx<float> *)out, (cplx<float> *)(out + sbytes),
                                   count, ngroups, scale);
            else
                hipLaunchKernelGGL(k_fft_sum_groups<double>, grid, dim3(64), 0, s, (const cplx<double> *)at(ctx, l.Gn),
                                   (const cplx<double> *)at(ctx, l.Gp), (cplx<double> *)out,
                                   (cplx<double> *)(out + sbytes), count, ngroups, scale);
            TNMF_LAUNCH_CHECK();
        }
        FftArgs b = base_args(g, l);
        b.src0 = at(ctx, l.Gs);
        b.dst0 = at(ctx, l.TW);   // [2*M*C][Ay][KXP]
        b.planes = planes;
        b.rows = g.Ay;
        b.yoff = 0;
        CHECK(l.colf(kFftColsInv, dtype, &b, s));
    }
    FftArgs d = base_args(g, l);
    d.src0 = at(ctx, l.TW);
    d.dst0 = at(ctx, l.Wo);
    d.planes = planes;
    d.rows = g.Ay;
    d.cols = g.Ax;
    d.xoff = 0;
    d.ld_dst = g.Ax;
    d.ps_dst = (long)g.Ay * g.Ax;
    CHECK(l.rowf(kFftRowsInv, dtype, &d, s));
    const int per = g.M * g.C * g.Ay * g.Ax;
    const char *wo = at(ctx, l.Wo);
    if (dtype == 0) {
        hipLaunchKernelGGL(k_fft_flip_out<float>, dim3(cdiv(per, 256)), dim3(256), 0, s, (const float *)wo,
                           (float *)neg, g.M * g.C, g.Ay, g.Ax, nonneg ? 1 : 0);
        hipLaunchKernelGGL(k_fft_flip_out<float>, dim3(cdiv(per, 256)), dim3(256), 0, s, (const float *)wo + per,
                           (float *)pos, g.M * g.C, g.Ay, g.Ax, nonneg ? 1 : 0);
    } else {
        hipLaunchKernelGGL(k_fft_flip_out<double>, dim3(cdiv(per, 256)), dim3(256), 0, s, (const double *)wo,
                           (double *)neg, g.M * g.C, g.Ay, g.Ax, nonneg ? 1 : 0);
        hipLaunchKernelGGL(k_fft_flip_out<double>, dim3(cdiv(per, 256)), dim3(256), 0, s, (const double *)wo + per,
                           (double *)pos, g.M * g.C, g.Ay, g.Ax, nonneg ? 1 : 0);
    }
    TNMF_LAUNCH_CHECK();
    return TNMF_OK;
}
