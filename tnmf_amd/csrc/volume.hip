// volume.hip -- the three primitives and the mode padding for THREE shift axes (volumes), float32 and float64.
//
//   R[n,c,d]        = sum_m sum_a W[m,c,a] * H[n,m,d + A-1 - a]                   (NumPy.py:122-132, k-generic)
//   gradH[n,m,h]    = sum_c sum_a W[m,c,a] * X[n,c,h - (A-1) + a]   X = V | R     (NumPy.py:93-120; zero outside X)
//   gradW[m,c,a]    = sum_n sum_d X[n,c,d] * H[n,m,d + A-1 - a]     X = V | R     (NumPy.py:69-91)
//
// with d, a, h three-component indices.  Direct kernels, one thread per output voxel (reconstruct, H gradient) or one
// workgroup per dictionary entry (W gradient): a workgroup stays inside one (sample, channel) or (sample, atom) volume,
// so every dictionary address is wave-uniform (scalar loads) and the lanes run along x (coalesced rows).  No BASELINE
// configuration has three shift axes; this family exists so that what the reference accepts runs here, not for speed.
// The W gradient accumulates in double and reduces in a fixed order: bit-reproducible.
#include <hip/hip_runtime.h>

#include "volume.h"

namespace {

constexpr int kVolBlock = 256;

template <typename T>
__global__ __launch_bounds__(kVolBlock) void k_vol_reconstruct(Vol v, int tiles, const T *__restrict__ W,
                                                               const T *__restrict__ H, T *__restrict__ R) {
    const int nc = blockIdx.x / tiles, tile = blockIdx.x - nc * tiles;
    const int n = nc / v.C, c = nc - n * v.C;
    const int vox = v.D[0] * v.D[1] * v.D[2], hvox = v.H[0] * v.H[1] * v.H[2], avox = v.A[0] * v.A[1] * v.A[2];
    const int i = tile * kVolBlock + threadIdx.x;
    if (i >= vox) return;
    const int x = i % v.D[2], y = (i / v.D[2]) % v.D[1], z = i / (v.D[2] * v.D[1]);
    T acc = 0;
    for (int m = 0; m < v.M; ++m) {
        const T *Hm = H + ((size_t)n * v.M + m) * hvox;
        const T *Wm = W + ((size_t)m * v.C + c) * avox;
        for (int az = 0; az < v.A[0]; ++az)
            for (int ay = 0; ay < v.A[1]; ++ay) {
                const T *hrow = Hm + ((size_t)(z + v.A[0] - 1 - az) * v.H[1] + (y + v.A[1] - 1 - ay)) * v.H[2] + x + v.A[2] - 1;
                const T *wrow = Wm + (az * v.A[1] + ay) * v.A[2];
#pragma unroll 4
                for (int ax = 0; ax < v.A[2]; ++ax) acc += wrow[ax] * hrow[-ax];
            }
    }
    R[(size_t)nc * vox + i] = acc;
}

template <typename T, bool FUSED>
__global__ __launch_bounds__(kVolBlock) void k_vol_corr_W(Vol v, int tiles, const T *__restrict__ V, const T *__restrict__ Rr,
                                                          const T *__restrict__ W, T *__restrict__ Hio,
                                                          T *__restrict__ neg, T *__restrict__ pos, T reg) {
    const int nm = blockIdx.x / tiles, tile = blockIdx.x - nm * tiles;
    const int n = nm / v.M, m = nm - n * v.M;
    const int vox = v.D[0] * v.D[1] * v.D[2], hvox = v.H[0] * v.H[1] * v.H[2], avox = v.A[0] * v.A[1] * v.A[2];
    const int i = tile * kVolBlock + threadIdx.x;
    if (i >= hvox) return;
    const int hx = i % v.H[2], hy = (i / v.H[2]) % v.H[1], hz = i / (v.H[2] * v.H[1]);
    T an = 0, ap = 0;
    for (int c = 0; c < v.C; ++c) {
        const T *Vc = V + ((size_t)n * v.C + c) * vox;
        const T *Rc = Rr + ((size_t)n * v.C + c) * vox;
        const T *Wc = W + ((size_t)m * v.C + c) * avox;
        for (int az = 0; az < v.A[0]; ++az) {
            const int z = hz - (v.A[0] - 1) + az;
            if (z < 0 || z >= v.D[0]) continue;
            for (int ay = 0; ay < v.A[1]; ++ay) {
                const int y = hy - (v.A[1] - 1) + ay;
                if (y < 0 || y >= v.D[1]) continue;
                const size_t row = ((size_t)z * v.D[1] + y) * v.D[2];
                const T *wrow = Wc + (az * v.A[1] + ay) * v.A[2];
                for (int ax = 0; ax < v.A[2]; ++ax) {
                    const int x = hx - (v.A[2] - 1) + ax;
                    if (x < 0 || x >= v.D[2]) continue;
                    const T w = wrow[ax];
                    an += w * Vc[row + x];
                    ap += w * Rc[row + x];
                }
            }
        }
    }
    const size_t o = (size_t)nm * hvox + i;
    if (FUSED) {
        Hio[o] = Hio[o] * an / (ap + reg);   // TransformInvariantNMF.py:232-235
    } else {
        neg[o] = an;
        pos[o] = ap;
    }
}

// W gradient.  A workgroup owns one (atom, channel, az, ay) and a chunk p of the rows (n, z, y) of the samples; its waves
// take rows in turn, the lanes run along x, and every thread keeps the sums of up to kVolTaps taps ax -- one load of V and
// R per voxel serves all of them.  Sums are doubles; lanes, waves (and, in k_vol_corr_H_finalize, chunks) are added up
// in a fixed order: bit-reproducible.  partials[p][(m, c, a)][neg | pos].
constexpr int kVolTaps = 8;

template <typename T>
__global__ __launch_bounds__(kVolBlock) void k_vol_corr_H(Vol v, int P, const T *__restrict__ V, const T *__restrict__ Rr,
                                                          const T *__restrict__ H, double *__restrict__ partials) {
    const int avox = v.A[0] * v.A[1] * v.A[2], nzy = v.A[0] * v.A[1];
    const int p = blockIdx.x % P, e = blockIdx.x / P;
    const int azy = e % nzy, mc = e / nzy;
    const int c = mc % v.C, m = mc / v.C;
    const int ay = azy % v.A[1], az = azy / v.A[1];
    const int oz = v.A[0] - 1 - az, oy = v.A[1] - 1 - ay;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long rows = (long)v.N * v.D[0] * v.D[1];
    const long r0 = rows * p / P, r1 = rows * (p + 1) / P;
    __shared__ double red[kVolBlock / 64][2 * kVolTaps];
    for (int a0 = 0; a0 < v.A[2]; a0 += kVolTaps) {
        double sn[kVolTaps], sp[kVolTaps];
#pragma unroll
        for (int t = 0; t < kVolTaps; ++t) sn[t] = sp[t] = 0;
        const int nt = v.A[2] - a0 < kVolTaps ? v.A[2] - a0 : kVolTaps;
        for (long r = r0 + wave; r < r1; r += kVolBlock / 64) {
            const int y = (int)(r % v.D[1]);
            const long rest = r / v.D[1];
            const int z = (int)(rest % v.D[0]), n = (int)(rest / v.D[0]);
            const size_t xo = ((((size_t)n * v.C + c) * v.D[0] + z) * v.D[1] + y) * v.D[2];
            // tap ax reads H at x + A[2]-1 - ax
            const T *hrow = H + ((((size_t)n * v.M + m) * v.H[0] + z + oz) * v.H[1] + y + oy) * v.H[2] + v.A[2] - 1 - a0;
            for (int x = lane; x < v.D[2]; x += 64) {
                const double vv = (double)V[xo + x], rr = (double)Rr[xo + x];
#pragma unroll
                for (int t = 0; t < kVolTaps; ++t)
                    if (t < nt) {
                        const double h = (double)hrow[x - t];
                        sn[t] += h * vv;
                        sp[t] += h * rr;
                    }
            }
        }
#pragma unroll
        for (int t = 0; t < kVolTaps; ++t) {
#pragma unroll
            for (int w = 32; w > 0; w >>= 1) {
                sn[t] += __shfl_down(sn[t], w);
                sp[t] += __shfl_down(sp[t], w);
            }
            if (lane == 0) {
                red[wave][2 * t] = sn[t];
                red[wave][2 * t + 1] = sp[t];
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < 2 * nt) {
            double acc = red[0][threadIdx.x];
            for (int w = 1; w < kVolBlock / 64; ++w) acc += red[w][threadIdx.x];
            const int t = threadIdx.x >> 1;
            const size_t o = ((size_t)mc * nzy + azy) * v.A[2] + a0 + t;
            partials[((size_t)p * v.M * v.C * avox + o) * 2 + (threadIdx.x & 1)] = acc;
        }
        __syncthreads();
    }
}

template <typename T>
__global__ void k_vol_corr_H_finalize(int n, int P, const double *__restrict__ partials, T *__restrict__ neg,
                                      T *__restrict__ pos) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a = 0, b = 0;
    for (int p = 0; p < P; ++p) {
        a += partials[((size_t)p * n + i) * 2];
        b += partials[((size_t)p * n + i) * 2 + 1];
    }
    neg[i] = (T)a;
    pos[i] = (T)b;
}

// ---- reconstruction modes (backends/_PyTorchBackend.py:42-52): the same per-axis maps as the 1-D / 2-D kernels of
// generic.hip.  S = activation length of the axis in this mode, a = atom length; the padded length is always D + a - 1.
__device__ __forceinline__ int vol_pad_src(int j, int S, int a, int mode) {   // activation copied to padded position j, or -1
    const int l = a - 1;
    if (mode == TNMF_MODE_FULL) {
        const int u = j - l;
        return (u >= 0 && u < S) ? u : -1;
    }
    if (j >= l) return j - l;
    return mode == TNMF_MODE_CIRCULAR ? S - l + j : l - j;
}
__device__ __forceinline__ int vol_pad_dup(int u, int S, int a, int mode) {   // second padded copy of activation u, or -1
    const int l = a - 1;
    if (mode == TNMF_MODE_CIRCULAR) return u >= S - l ? u - (S - l) : -1;
    if (mode == TNMF_MODE_REFLECT) return (u >= 1 && u <= l) ? l - u : -1;
    return -1;
}

struct Vol3 {
    int S[3], A[3], P[3];
};

template <typename T>
__global__ void k_vol_pad(size_t planes, Vol3 q, int mode, const T *__restrict__ H, T *__restrict__ Hp) {
    const size_t pvox = (size_t)q.P[0] * q.P[1] * q.P[2], total = planes * pvox;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const size_t r = e / pvox, w = e - r * pvox;
        const int jx = (int)(w % q.P[2]), jy = (int)((w / q.P[2]) % q.P[1]), jz = (int)(w / ((size_t)q.P[2] * q.P[1]));
        const int uz = vol_pad_src(jz, q.S[0], q.A[0], mode), uy = vol_pad_src(jy, q.S[1], q.A[1], mode),
                  ux = vol_pad_src(jx, q.S[2], q.A[2], mode);
        Hp[e] = (uz >= 0 && uy >= 0 && ux >= 0) ? H[((r * q.S[0] + uz) * q.S[1] + uy) * q.S[2] + ux] : T(0);
    }
}

template <typename T>
__global__ void k_vol_fold(size_t planes, Vol3 q, int mode, const T *__restrict__ Gp, T *__restrict__ G) {
    const size_t svox = (size_t)q.S[0] * q.S[1] * q.S[2], total = planes * svox;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const size_t r = e / svox, w = e - r * svox;
        const int ux = (int)(w % q.S[2]), uy = (int)((w / q.S[2]) % q.S[1]), uz = (int)(w / ((size_t)q.S[2] * q.S[1]));
        const int jz[2] = {uz + q.A[0] - 1, vol_pad_dup(uz, q.S[0], q.A[0], mode)};
        const int jy[2] = {uy + q.A[1] - 1, vol_pad_dup(uy, q.S[1], q.A[1], mode)};
        const int jx[2] = {ux + q.A[2] - 1, vol_pad_dup(ux, q.S[2], q.A[2], mode)};
        T acc = T(0);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    if (jz[a] >= 0 && jy[b] >= 0 && jx[c] >= 0)
                        acc += Gp[((r * q.P[0] + jz[a]) * q.P[1] + jy[b]) * q.P[2] + jx[c]];
        G[e] = acc;
    }
}

// lateral terms of the H half step (TransformInvariantNMF.py:253-269), in place:  G <- inh * (G - H) + xc * (sum_m G - G)
// one thread per (sample, voxel), the atoms walked twice (the second pass re-reads what the first just touched)
template <typename T>
__global__ void k_vol_lateral(size_t N, int M, size_t vox, T *__restrict__ G, const T *__restrict__ H, T inh, T xc) {
    const size_t total = N * vox, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const size_t n = e / vox, o0 = n * M * vox + (e - n * vox);
        T S = 0;
        if (xc != T(0))
            for (int m = 0; m < M; ++m) S += G[o0 + (size_t)m * vox];
        for (int m = 0; m < M; ++m) {
            const size_t o = o0 + (size_t)m * vox;
            const T gv = G[o];
            G[o] = inh * (gv - H[o]) + xc * (S - gv);
        }
    }
}

// every index inside one volume is an int; volumes times samples go through size_t
bool vol_fits(const Vol &v) {
    const long long lim = 0x7fffffffLL;
    const long long vox = (long long)v.D[0] * v.D[1] * v.D[2], hvox = (long long)v.H[0] * v.H[1] * v.H[2];
    const long long avox = (long long)v.A[0] * v.A[1] * v.A[2];
    const long long tiles = (hvox + kVolBlock - 1) / kVolBlock;
    return vox < lim && hvox < lim && avox * v.M * v.C < lim && tiles * v.N * (v.M > v.C ? v.M : v.C) < lim;
}

}  // namespace

int vol_reconstruct(const Vol &v, int dtype, const void *W, const void *H, void *R, hipStream_t s) {
    if (!vol_fits(v)) return TNMF_E_GEOM;
    if (v.N == 0) return TNMF_OK;
    const int tiles = cdiv(v.D[0] * v.D[1] * v.D[2], kVolBlock);
    const dim3 grid((unsigned)(tiles * v.N * v.C));
    if (dtype == 0)
        hipLaunchKernelGGL(k_vol_reconstruct<float>, grid, dim3(kVolBlock), 0, s, v, tiles, (const float *)W,
                           (const float *)H, (float *)R);
    else
        hipLaunchKernelGGL(k_vol_reconstruct<double>, grid, dim3(kVolBlock), 0, s, v, tiles, (const double *)W,
                           (const double *)H, (double *)R);
    TNMF_LAUNCH_CHECK();
    return TNMF_OK;
}

int vol_corr_W(const Vol &v, int dtype, const void *V, const void *R, const void *W, void *Hio, void *neg, void *pos,
               bool fused, double reg, hipStream_t s) {
    if (!vol_fits(v)) return TNMF_E_GEOM;
    if (v.N == 0) return TNMF_OK;
    const int tiles = cdiv(v.H[0] * v.H[1] * v.H[2], kVolBlock);
    const dim3 grid((unsigned)(tiles * v.N * v.M));
#define VOL_CW(T_, F_)                                                                                              \
    hipLaunchKernelGGL((k_vol_corr_W<T_, F_>), grid, dim3(kVolBlock), 0, s, v, tiles, (const T_ *)V, (const T_ *)R,   \
                       (const T_ *)W, (T_ *)Hio, (T_ *)neg, (T_ *)pos, (T_)reg)
    if (dtype == 0) {
        if (fused) VOL_CW(float, true);
        else VOL_CW(float, false);
    } else {
        if (fused) VOL_CW(double, true);
        else VOL_CW(double, false);
    }
#undef VOL_CW
    TNMF_LAUNCH_CHECK();
    return TNMF_OK;
}

int vol_corr_H_chunks(const tnmf_hip_ctx *ctx, const Vol &v) {
    // enough workgroups for four per CU, no chunk below eight rows
    const long entries = (long)v.M * v.C * v.A[0] * v.A[1], rows = (long)v.N * v.D[0] * v.D[1];
    long P = (4L * ctx->num_cu + entries - 1) / entries;
    if (P > rows / 8) P = rows / 8;
    if (P > 1024) P = 1024;
    return P < 1 ? 1 : (int)P;
}

int vol_corr_H(const Vol &v, int dtype, const void *V, const void *R, const void *H, void *neg, void *pos,
               double *partials, int P, hipStream_t s) {
    if (!vol_fits(v)) return TNMF_E_GEOM;
    const long entries = (long)v.M * v.C * v.A[0] * v.A[1];
    if (entries * P > 0x7fffffffL) return TNMF_E_GEOM;
    const dim3 grid((unsigned)(entries * P));   // (an empty slice: every sum is zero)
    const int n = v.M * v.C * v.A[0] * v.A[1] * v.A[2];
    if (dtype == 0) {
        hipLaunchKernelGGL(k_vol_corr_H<float>, grid, dim3(kVolBlock), 0, s, v, P, (const float *)V, (const float *)R,
                           (const float *)H, partials);
        hipLaunchKernelGGL(k_vol_corr_H_finalize<float>, dim3(cdiv(n, kVolBlock)), dim3(kVolBlock), 0, s, n, P, partials,
                           (float *)neg, (float *)pos);
    } else {
        hipLaunchKernelGGL(k_vol_corr_H<double>, grid, dim3(kVolBlock), 0, s, v, P, (const double *)V, (const double *)R,
                           (const double *)H, partials);
        hipLaunchKernelGGL(k_vol_corr_H_finalize<double>, dim3(cdiv(n, kVolBlock)), dim3(kVolBlock), 0, s, n, P, partials,
                           (double *)neg, (double *)pos);
    }
    TNMF_LAUNCH_CHECK();
    return TNMF_OK;
}

int vol_pad_fold(const tnmf_hip_ctx *ctx, const Vol &v, int dtype, int mode, bool fold, const void *in, void *out,
                 hipStream_t s) {
    Vol3 q;
    for (int i = 0; i < 3; ++i) {
        q.A[i] = v.A[i];
        q.P[i] = v.H[i];
        q.S[i] = mode == TNMF_MODE_VALID ? v.H[i] : (mode == TNMF_MODE_FULL ? v.D[i] - v.A[i] + 1 : v.D[i]);
        if (q.S[i] < 1) return TNMF_E_GEOM;
        // (the same limits as launch_pad_fold of generic.hip: at most one wrap; 'reflect' mirrors without the edge)
        if (mode != TNMF_MODE_VALID && v.A[i] - 1 > q.S[i]) return TNMF_E_GEOM;
        if (mode == TNMF_MODE_REFLECT && v.A[i] - 1 >= q.S[i]) return TNMF_E_GEOM;
    }
    const size_t planes = (size_t)v.N * v.M;
    const size_t total = planes * (fold ? (size_t)q.S[0] * q.S[1] * q.S[2] : (size_t)q.P[0] * q.P[1] * q.P[2]);
    if (total == 0) return TNMF_OK;
    if (mode == TNMF_MODE_VALID) {   // identity
        TNMF_HIP_TRY(hipMemcpyAsync(out, in, total * (dtype == 0 ? 4 : 8), hipMemcpyDeviceToDevice, s));
        return TNMF_OK;
    }
    size_t blocks = (total + kVolBlock - 1) / kVolBlock;
    const size_t cap = (size_t)ctx->num_cu * 32;
    if (blocks > cap) blocks = cap;
    const dim3 grid((unsigned)blocks);
    if (dtype == 0) {
        if (fold) hipLaunchKernelGGL(k_vol_fold<float>, grid, dim3(kVolBlock), 0, s, planes, q, mode, (const float *)in, (float *)out);
        else hipLaunchKernelGGL(k_vol_pad<float>, grid, dim3(kVolBlock), 0, s, planes, q, mode, (const float *)in, (float *)out);
    } else {
        if (fold) hipLaunchKernelGGL(k_vol_fold<double>, grid, dim3(kVolBlock), 0, s, planes, q, mode, (const double *)in, (double *)out);
        else hipLaunchKernelGGL(k_vol_pad<double>, grid, dim3(kVolBlock), 0, s, planes, q, mode, (const double *)in, (double *)out);
    }
    TNMF_LAUNCH_CHECK();
    return TNMF_OK;
}

int vol_lateral(const tnmf_hip_ctx *ctx, int dtype, size_t N, int M, size_t vox, void *G, const void *H, double inh,
                double xc, hipStream_t s) {
    const size_t total = N * vox;
    if (total == 0) return TNMF_OK;
    size_t blocks = (total + kVolBlock - 1) / kVolBlock;
    const size_t cap = (size_t)ctx->num_cu * 32;
    if (blocks > cap) blocks = cap;
    if (dtype == 0)
        hipLaunchKernelGGL(k_vol_lateral<float>, dim3((unsigned)blocks), dim3(kVolBlock), 0, s, N, M, vox, (float *)G,
                           (const float *)H, (float)inh, (float)xc);
    else
        hipLaunchKernelGGL(k_vol_lateral<double>, dim3((unsigned)blocks), dim3(kVolBlock), 0, s, N, M, vox, (double *)G,
                           (const double *)H, inh, xc);
    TNMF_LAUNCH_CHECK();
    return TNMF_OK;
}
