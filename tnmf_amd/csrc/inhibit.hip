// inhibit.hip -- the lateral terms of the H half step and the fold + update of the reconstruction modes, as kernels of
// the library (the reference computes them with array arithmetic in its front end:
// tnmf/TransformInvariantNMF.py:253-269 on top of tnmf/backends/_NumPyBackend.py:56-64).
//
//   G[n,m]  = ky (*)_y kx (*)_x H[n,m]            separable 'same' convolution, zeros outside (scipy convolve1d, cval 0)
//   E[n,m]  = inh * (G[n,m] - H[n,m])  +  xc * (sum_m' G[n,m'] - G[n,m]),      xc = cross_inhibition / (M - 1)
//
// E is the extra term of the denominator of the multiplicative update: H <- H * neg / (pos + E + eps + sparsity).  One
// kernel computes it: a workgroup owns a 32 x 32 pixel tile of one sample and walks the atoms; per atom the tile with its
// halo goes through LDS once (x pass into a second LDS array, y pass into registers; every thread slides a register
// window along the convolution axis: one LDS read per 8 / 4 multiply-adds).  With cross inhibition the G of every atom
// is parked in E on the way and a second walk over the atoms (its reads hit in L2) turns it into E.
#include "generic.h"

namespace {

constexpr int kTY = 32, kTX = 32, kThreads = 256;   // 32 x 32 pixel tiles: one round of items in either pass, least halo

// taps in WINDOW order (weight of window offset d = kernel[l - 1 - d]: scipy's convolve1d is a true convolution), already in
// the element type: uniform reads of a kernel argument, i.e. scalar loads -- not LDS reads, not conversions
template <typename T>
struct Taps2 {
    T ky[kMaxTaps];
    T kx[kMaxTaps];
};

template <typename T>
__device__ __forceinline__ T buffer_load_elem(const __amdgpu_buffer_rsrc_t &rs, int byte_off);
template <>
__device__ __forceinline__ float buffer_load_elem<float>(const __amdgpu_buffer_rsrc_t &rs, int byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, byte_off, 0, 0));
}
template <>
__device__ __forceinline__ double buffer_load_elem<double>(const __amdgpu_buffer_rsrc_t &rs, int byte_off) {
    typedef unsigned u32x2i __attribute__((ext_vector_type(2)));
    const u32x2i w = __builtin_amdgcn_raw_buffer_load_b64(rs, byte_off, 0, 0);
    const unsigned long long b = (unsigned long long)w[0] | ((unsigned long long)w[1] << 32);
    return __builtin_bit_cast(double, b);
}

// H, E: [N][M][Hy][ld] (the same row stride; pad columns beyond the shift width are pixels that hold zeros)
//
// Instruction budget (the kernel is VALU bound: 23 + 23 taps on 5.8e8 activations at config 3): per atom and thread about
// 13 staged elements (row / column by increments, no division), 23 x (1 LDS read + 8 multiply-adds) in the x pass over
// one 8-column item, 23 x (1 LDS read + 4 multiply-adds) in the y pass over 4 rows of one column.
// LYC / LXC: kernel lengths known at compile time (0: run-time lengths) -- the tap loops of the default inhibition ranges
// (atom size - 1 per axis: 23 taps for 12 x 12 atoms) unroll completely: no loop control, no guards, immediate offsets.
template <typename T, int LYC, int LXC>
__global__ __launch_bounds__(kThreads) void k_inhibition(const T *__restrict__ H, T *__restrict__ E, int M, int Hy, int ld,
                                                       Taps2<T> taps, int ly_rt, int lx_rt, T inh, T xc, int tiles_y,
                                                       int tiles_x) {
    const int ly = LYC ? LYC : ly_rt, lx = LXC ? LXC : lx_rt;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int ry = (ly - 1) / 2, rx = (lx - 1) / 2;
    // (+8: the register window of the x pass reads 7 past its last tap; both LDS row strides are ODD: in the x pass the
    // lanes of a wave are consecutive ROWS of one 8-column segment, so a stride coprime to the bank count keeps every read
    // of A0 and every write of A1 free of bank conflicts)
    const int SH = kTY + 2 * ry, SW = (kTX + 2 * rx + 8) | 1;
    constexpr int S1 = kTX + 1;
    T *A0 = reinterpret_cast<T *>(smem_raw);             // [SH][SW]   tile + halo
    T *A1 = A0 + (size_t)SH * SW;                        // [SH][S1]   after the x pass
    // XCD-aware block order: consecutive workgroups are dealt round-robin to the eight XCDs, each with its own L2 -- in
    // logical order the neighbours of a tile, which share its halo (54 x 54 values for 32 x 32 outputs), sat on other
    // XCDs and every halo came from HBM (PMC: 11.7 GB moved per launch for 5.0 GB of activations in and out).  Remapped,
    // each XCD walks a contiguous run of (sample, tile) pairs and the halos meet in its L2.
    unsigned bid = blockIdx.x;
    {
        const unsigned whole = gridDim.x / 8 * 8;
        if (bid < whole) bid = (bid & 7) * (whole / 8) + (bid >> 3);
    }
    const int txi = bid % tiles_x;
    bid /= tiles_x;
    const int tyi = bid % tiles_y;
    const int n = bid / tiles_y;
    const int u0 = tyi * kTY, v0 = txi * kTX;
    const int col = threadIdx.x & (kTX - 1), rg = threadIdx.x / kTX;   // y pass / output: column, group of 4 rows
    const size_t plane = (size_t)Hy * ld;
    // compile-time lengths: the taps live in VECTOR registers for the whole kernel (read once through LDS -- 46 uniform
    // values are more than the scalar file holds beside everything else: the compiler spilled them lane by lane)
    T kxr[LXC > 0 ? LXC : 1], kyr[LYC > 0 ? LYC : 1];
    if constexpr (LXC > 0 && LYC > 0) {
        T *kt = A1;   // (scratch use before the first atom; the barrier at the top of the atom loop follows)
        for (int i = threadIdx.x; i < LXC + LYC; i += kThreads) kt[i] = i < LXC ? taps.kx[i] : taps.ky[i - LXC];
        __syncthreads();
#pragma unroll
        for (int d = 0; d < LXC; ++d) kxr[d] = kt[d];
#pragma unroll
        for (int d = 0; d < LYC; ++d) kyr[d] = kt[LXC + d];
    }
    T S[4] = {T(0), T(0), T(0), T(0)};
    const bool cross = xc != T(0);
    // Staging: the tile + halo of the NEXT atom is fetched into registers (kPre values per thread, unconditional loads on
    // clamped addresses, all in flight at once) while the current atom is convolved, and parked in LDS behind it.  Element
    // i = tid + k * 256 of the [SH][SW] tile: its row / column follow from those of element tid by constant increments.
    // Tiles too large for the register stage (very long kernels) are staged in batches of eight loads instead.
    constexpr int kPre = 16;
    const int nel = SH * SW;
    const bool pre_ok = nel <= kPre * kThreads;
    const int dr = kThreads / SW, dq = kThreads - dr * SW;   // (r, q) of element i + 256 = (r + dr, q + dq) or (r + dr + 1, q + dq - SW)
    const int r_first = (int)threadIdx.x / SW, q_first = (int)threadIdx.x - r_first * SW;
    T pre[kPre];
    auto fetch = [&](int m_, int k0, T *dst, int cnt) {   // elements tid + (k0 + k) * 256, k < cnt
        // one buffer descriptor per plane, 32-bit element offsets: rows above / below the plane fail the range check and
        // read as zero by themselves, columns left / right of it are masked (no 64-bit address arithmetic per element)
        const __amdgpu_buffer_rsrc_t hr = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(H + ((size_t)n * M + m_) * plane), 0, (int)(plane * sizeof(T)), 0x00020000);
        int r = r_first, q = q_first;
        for (int k = 0; k < k0; ++k) {
            r += dr;
            q += dq;
            if (q >= SW) {
                q -= SW;
                ++r;
            }
        }
#pragma unroll
        for (int k = 0; k < kPre; ++k) {
            if (k < cnt) {
                const int y = u0 + r - ry, x = v0 + q - rx;
                const bool xin = x >= 0 && x < ld;
                // (y < 0: the offset is negative = huge unsigned: out of range like y >= Hy)
                const int off = (y * ld + (xin ? x : 0)) * (int)sizeof(T);
                dst[k] = xin ? buffer_load_elem<T>(hr, off) : T(0);
                r += dr;
                q += dq;
                if (q >= SW) {
                    q -= SW;
                    ++r;
                }
            }
        }
    };
    auto park = [&](int k0, const T *src, int cnt) {
#pragma unroll
        for (int k = 0; k < kPre; ++k) {
            const int i = (k0 + k) * kThreads + threadIdx.x;
            if (k < cnt && i < nel) A0[i] = src[k];
        }
    };
    const int npre = (nel + kThreads - 1) / kThreads;   // (<= kPre when pre_ok)
    if (pre_ok) fetch(0, 0, pre, npre);
    for (int m = 0; m < M; ++m) {
        __syncthreads();   // the previous atom's passes are done with A0 / A1
        if (pre_ok) {
            park(0, pre, npre);
        } else {
            for (int k0 = 0; k0 < npre; k0 += 8) {
                T tmp[kPre];
                fetch(m, k0, tmp, 8);
                park(k0, tmp, 8);
            }
        }
        __syncthreads();
        if (pre_ok && m + 1 < M) fetch(m + 1, 0, pre, npre);   // in flight under the two passes below
        // x pass: item = (row, segment of 8 columns); out[c] = sum_d kx[d] * A0[row][c + d]
        // (wave w takes segment w = columns 8 w .. 8 w + 7, its lanes the rows)
        static_assert(kTX / 8 == kThreads / 64, "one 8-column segment per wave");
        for (int r = threadIdx.x & 63; r < SH; r += 64) {
            const int c0 = (threadIdx.x >> 6) * 8;
            const T *row = A0 + (size_t)r * SW + c0;
            T win[8], acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc[j] = T(0);
                win[j] = j < 7 ? row[j] : T(0);
            }
            if constexpr (LXC > 0) {
#pragma unroll
                for (int d = 0; d < LXC; ++d) {
                    win[(d + 7) & 7] = row[d + 7];
                    const T k = kxr[d];
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += k * win[(d + j) & 7];
                }
            } else {
                for (int d0 = 0; d0 < lx; d0 += 8) {
#pragma unroll
                    for (int dd = 0; dd < 8; ++dd) {
                        const int d = d0 + dd;
                        if (d < lx) {
                            win[(dd + 7) & 7] = row[d + 7];
                            const T k = taps.kx[d];
#pragma unroll
                            for (int j = 0; j < 8; ++j) acc[j] += k * win[(dd + j) & 7];
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) A1[(size_t)r * S1 + c0 + j] = acc[j];
        }
        __syncthreads();
        // y pass: rows 4 rg .. 4 rg + 3 of column col; out[r] = sum_d ky[d] * A1[r + d][col]
        T g4[4], win[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            g4[j] = T(0);
            win[j] = j < 3 ? A1[(size_t)(4 * rg + j) * S1 + col] : T(0);
        }
        if constexpr (LYC > 0) {
#pragma unroll
            for (int d = 0; d < LYC; ++d) {
                win[(d + 3) & 3] = A1[(size_t)(4 * rg + d + 3) * S1 + col];   // (row <= SH - 1)
                const T k = kyr[d];
#pragma unroll
                for (int j = 0; j < 4; ++j) g4[j] += k * win[(d + j) & 3];
            }
        } else {
            for (int d0 = 0; d0 < ly; d0 += 4) {
#pragma unroll
                for (int dd = 0; dd < 4; ++dd) {
                    const int d = d0 + dd;
                    if (d < ly) {
                        const int rr = 4 * rg + d + 3;   // (< SH whenever it matters: the last window row of the last output)
                        win[(dd + 3) & 3] = A1[(size_t)(rr < SH ? rr : SH - 1) * S1 + col];
                        const T k = taps.ky[d];
#pragma unroll
                        for (int j = 0; j < 4; ++j) g4[j] += k * win[(dd + j) & 3];
                    }
                }
            }
        }
        const int x = v0 + col;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int u = u0 + 4 * rg + j;
            if (u < Hy && x < ld) {
                const size_t o = ((size_t)n * M + m) * plane + (size_t)u * ld + x;
                if (cross) {
                    S[j] += g4[j];
                    E[o] = g4[j];   // parked; the second walk below turns it into E
                } else {
                    E[o] = inh * (g4[j] - A0[(size_t)(4 * rg + j + ry) * SW + col + rx]);
                }
            }
        }
    }
    if (cross) {
        const int x = v0 + col;
        for (int m = 0; m < M; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int u = u0 + 4 * rg + j;
                if (u < Hy && x < ld) {
                    const size_t o = ((size_t)n * M + m) * plane + (size_t)u * ld + x;
                    const T gv = E[o];   // (this thread's own store)
                    E[o] = inh * (gv - H[o]) + xc * (S[j] - gv);
                }
            }
    }
}

// valid mode, kernel families without an extra-term epilogue: H = H * neg / (pos + E + reg); H and E with row stride ld,
// neg / pos C-contiguous
template <typename T>
__global__ void k_mu_update_extra(T *__restrict__ H, const T *__restrict__ neg, const T *__restrict__ pos,
                                  const T *__restrict__ E, size_t rows, int Hx, int ld, T reg) {
    const size_t total = rows * (size_t)Hx;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const size_t r = e / Hx;
        const size_t o = r * ld + (e - r * Hx);
        const T p = pos[e] + (E ? E[o] : T(0)) + reg;
        H[o] = (H[o] * neg[e]) / p;
    }
}

// activation index copied to padded position j of one axis (-1: a zero) and the second padded copy of activation u
// (generic.hip has the same tables for tnmf_hip_pad_H / tnmf_hip_fold_H; reference: backends/_PyTorchBackend.py:42-52)
__device__ __forceinline__ int dup_of(int u, int S, int a, int mode) {
    const int l = a - 1;
    if (mode == TNMF_MODE_CIRCULAR) return u >= S - l ? u - (S - l) : -1;
    if (mode == TNMF_MODE_REFLECT) return (u >= 1 && u <= l) ? l - u : -1;
    return -1;
}

// reconstruction modes: the gradient with respect to an activation is the sum of the gradients at the padded positions
// that copy it (fold = adjoint of the pad), then the multiplicative update -- in one pass:
//   H[u] = H[u] * sum_copies negp / (sum_copies posp + E[u] + reg)
template <typename T>
__global__ void k_fold_update(size_t rows, int Sy, int Sx, int Ay, int Ax, int Py, int Px, int mode, T *__restrict__ H,
                              const T *__restrict__ negp, const T *__restrict__ posp, const T *__restrict__ E, T reg) {
    const size_t total = rows * (size_t)Sy * Sx;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int ux = (int)(e % Sx);
        const size_t rest = e / Sx;
        const int uy = (int)(rest % Sy);
        const size_t r = rest / Sy;
        const int jy[2] = {Py == 1 ? 0 : uy + Ay - 1, Py == 1 ? -1 : dup_of(uy, Sy, Ay, mode)};
        const int jx[2] = {ux + Ax - 1, dup_of(ux, Sx, Ax, mode)};
        T an = T(0), ap = T(0);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                if (jy[a] >= 0 && jx[b] >= 0) {
                    const size_t o = (r * Py + jy[a]) * Px + jx[b];
                    an += negp[o];
                    ap += posp[o];
                }
        H[e] = (H[e] * an) / (ap + (E ? E[e] : T(0)) + reg);
    }
}

inline int grid_cap(size_t n, const tnmf_hip_ctx *ctx) {
    const size_t want = (n + kThreads - 1) / kThreads, cap = (size_t)ctx->num_cu * 8;
    return (int)(want < cap ? (want ? want : 1) : cap);
}

template <typename T>
int launch_inhibition_t(int N, int M, int Hy, int ld, const void *H, void *E, const double *ky_host, int ly,
                        const double *kx_host, int lx, double inh, double xc, hipStream_t s) {
    const int ry = (ly - 1) / 2, rx = (lx - 1) / 2;
    const size_t SH = kTY + 2 * ry, SW = (kTX + 2 * rx + 8) | 1;
    const size_t lds = (SH * SW + SH * (kTX + 1)) * sizeof(T);
    if (lds > 160 * 1024) return TNMF_E_UNSUPPORTED;
    // (a plane is addressed through one buffer descriptor with 32-bit byte offsets)
    if ((size_t)Hy * ld * sizeof(T) >= ((size_t)1 << 31)) return TNMF_E_UNSUPPORTED;
    Taps2<T> taps;
    for (int i = 0; i < kMaxTaps; ++i) {
        taps.ky[i] = i < ly ? (T)ky_host[ly - 1 - i] : T(0);
        taps.kx[i] = i < lx ? (T)kx_host[lx - 1 - i] : T(0);
    }
    const int tiles_y = cdiv(Hy, kTY), tiles_x = cdiv(ld, kTX);
    const size_t blocks = (size_t)N * tiles_y * tiles_x;
    if (blocks > 0x7fffffffull) return TNMF_E_GEOM;
#define INH_LAUNCH(LY_, LX_)                                                                                           \
    do {                                                                                                             \
        if (lds > 64 * 1024)                                                                                         \
            TNMF_HIP_TRY(hipFuncSetAttribute((const void *)k_inhibition<T, LY_, LX_>,                                \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));               \
        hipLaunchKernelGGL((k_inhibition<T, LY_, LX_>), dim3((unsigned)blocks), dim3(kThreads), lds, s, (const T *)H, \
                           (T *)E, M, Hy, ld, taps, ly, lx, (T)inh, (T)xc, tiles_y, tiles_x);                        \
    } while (0)
    // the default inhibition ranges (atom size - 1) of the BASELINE atom shapes get unrolled tap loops
    if (ly == 23 && lx == 23) INH_LAUNCH(23, 23);
    else if (ly == 17 && lx == 17) INH_LAUNCH(17, 17);
    else if (ly == 31 && lx == 31) INH_LAUNCH(31, 31);
    else INH_LAUNCH(0, 0);
#undef INH_LAUNCH
    TNMF_LAUNCH_CHECK();
    return TNMF_OK;
}

}  // namespace

int launch_inhibition(const tnmf_hip_ctx *, int dtype, int N, int M, int Hy, int ld, const void *H, void *E,
                      const double *ky_host, int ly, const double *kx_host, int lx, double inh, double xc,
                      hipStream_t s) {
    if (ly < 1 || lx < 1 || ly > kMaxTaps || lx > kMaxTaps || !(ly & 1) || !(lx & 1)) return TNMF_E_UNSUPPORTED;
    if (N <= 0) return TNMF_OK;
    return dtype == 0 ? launch_inhibition_t<float>(N, M, Hy, ld, H, E, ky_host, ly, kx_host, lx, inh, xc, s)
                      : launch_inhibition_t<double>(N, M, Hy, ld, H, E, ky_host, ly, kx_host, lx, inh, xc, s);
}

int launch_mu_update_extra(const tnmf_hip_ctx *ctx, int dtype, void *H, const void *neg, const void *pos, const void *E,
                           size_t rows, int Hx, int ld, double reg, hipStream_t s) {
    const size_t total = rows * (size_t)Hx;
    if (total == 0) return TNMF_OK;
    const int grid = grid_cap(total, ctx);
    if (dtype == 0)
        hipLaunchKernelGGL(k_mu_update_extra<float>, dim3(grid), dim3(kThreads), 0, s, (float *)H, (const float *)neg,
                           (const float *)pos, (const float *)E, rows, Hx, ld, (float)reg);
    else
        hipLaunchKernelGGL(k_mu_update_extra<double>, dim3(grid), dim3(kThreads), 0, s, (double *)H, (const double *)neg,
                           (const double *)pos, (const double *)E, rows, Hx, ld, reg);
    TNMF_LAUNCH_CHECK();
    return TNMF_OK;
}

int launch_fold_update(const tnmf_hip_ctx *ctx, const Geo &g, int dtype, int mode, int Sy, int Sx, void *H,
                       const void *negp, const void *posp, const void *E, double reg, hipStream_t s) {
    const size_t rows = (size_t)g.N * g.M, total = rows * (size_t)Sy * Sx;
    if (total == 0) return TNMF_OK;
    const int grid = grid_cap(total, ctx);
    if (dtype == 0)
        hipLaunchKernelGGL(k_fold_update<float>, dim3(grid), dim3(kThreads), 0, s, rows, Sy, Sx, g.Ay, g.Ax, g.Hy, g.Hx, mode,
                           (float *)H, (const float *)negp, (const float *)posp, (const float *)E, (float)reg);
    else
        hipLaunchKernelGGL(k_fold_update<double>, dim3(grid), dim3(kThreads), 0, s, rows, Sy, Sx, g.Ay, g.Ax, g.Hy, g.Hx,
                           mode, (double *)H, (const double *)negp, (const double *)posp, (const double *)E, reg);
    TNMF_LAUNCH_CHECK();
    return TNMF_OK;
}
