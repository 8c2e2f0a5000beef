// fft_kernels.h -- the kernels of the FFT family, templated on the transform length; fft_len.hip instantiates them
// for one length per object file.  See fft.h for the formulation and fft_engine.h for the tile transform.
//
// Array kinds (T = float | double, c = cplx<T>):
//   real planes   [plane][rows][ld]                         V, R, H, W (C-contiguous as handed over by the caller)
//   row spectra   [plane][rows][KXP]  c, kx natural order   "T": transform along x only, one line per data row
//   full spectra  [plane][L][KXP]     c, ky digit-reversed  "S": what the pointwise contractions work on
// A row kernel works on NB row pairs (two real rows = one complex sequence), a column kernel on a tile of 16 kx.
#pragma once
#include "fft.h"
#include "fft_engine.h"

namespace {

constexpr int kColTile = 16;   // kx columns per column-kernel tile
constexpr int kColBS = 17;     // position stride of a column tile in LDS (odd: spreads the stage-3 blocks over banks)

template <int L>
struct LenCfg {
    static constexpr int col_threads = L > 192 ? 512 : 256;
    static constexpr int col_elems = L * kColTile / col_threads;   // tile elements per thread
    static constexpr int row_pairs = L > 288 ? 8 : 16;
    static constexpr int row_threads = 256;
    // channels per pass of the contraction kernels (accumulators live in registers)
    static constexpr int cg_R = 4;
    static constexpr int cg_W = col_elems <= 9 ? 4 : 2;
};

template <typename T, int L>
__device__ __forceinline__ void make_twiddles(cplx<T> *tw, int tid, int nt) {
    for (int t = tid; t < L; t += nt) {
        double sn, cs;
        sincospi(2.0 * t / L, &sn, &cs);
        tw[t] = {(T)cs, (T)(-sn)};
    }
}

// all NB sequences of the tile, forward; the caller has synchronised the tile, the function ends synchronised
template <typename T, typename P, int NB, int BS, int NT>
__device__ __forceinline__ void tile_fwd(cplx<T> *x, const cplx<T> *tw, int tid) {
    for (int t = tid; t < NB * P::tasks1; t += NT) P::template fwd1<BS>(x + (t % NB), tw, t / NB);
    __syncthreads();
    for (int t = tid; t < NB * P::tasks2; t += NT) P::template fwd2<BS>(x + (t % NB), tw, t / NB);
    __syncthreads();
    for (int t = tid; t < NB * P::tasks3; t += NT) P::template fwd3<BS>(x + (t % NB), t / NB);
    __syncthreads();
}

template <typename T, typename P, int NB, int BS, int NT>
__device__ __forceinline__ void tile_inv(cplx<T> *x, const cplx<T> *tw, int tid) {
    for (int t = tid; t < NB * P::tasks3; t += NT) P::template inv3<BS>(x + (t % NB), t / NB);
    __syncthreads();
    for (int t = tid; t < NB * P::tasks2; t += NT) P::template inv2<BS>(x + (t % NB), tw, t / NB);
    __syncthreads();
    for (int t = tid; t < NB * P::tasks1; t += NT) P::template inv1<BS>(x + (t % NB), tw, t / NB);
    __syncthreads();
}

// ---- row kernels ------------------------------------------------------------------------------------------------

// row spectra of 2*NB rows out of a transformed pair tile (kx natural order)
template <typename T, typename P, int NB, int BS, int NT>
__device__ __forceinline__ void store_rows_split(const cplx<T> *x, cplx<T> *dst, int y0, int rows, int KXP, int tid) {
    constexpr int L = P::L, KX = L / 2 + 1;
    for (int idx = tid; idx < 2 * NB * KX; idx += NT) {
        const int r = idx / KX, k = idx - r * KX, y = y0 + r;
        if (y >= rows) continue;
        const int pr = r >> 1;
        const cplx<T> z1 = x[P::pos_of_k(k) * BS + pr];
        const cplx<T> z2 = x[P::pos_of_k(k == 0 ? 0 : L - k) * BS + pr];
        cplx<T> A, B;
        split_pair(z1, z2, A, B);
        dst[(long)y * KXP + k] = (r & 1) ? B : A;
    }
}

// pair tile from the row spectra of 2*NB rows, ready for the inverse transform
template <typename T, typename P, int NB, int BS, int NT>
__device__ __forceinline__ void load_rows_merge(cplx<T> *x, const cplx<T> *src, int y0, int rows, int KXP, int tid) {
    constexpr int L = P::L, KX = L / 2 + 1;
    for (int idx = tid; idx < NB * KX; idx += NT) {
        const int pr = idx / KX, k = idx - pr * KX, ya = y0 + 2 * pr;
        cplx<T> A = {0, 0}, B = {0, 0};
        if (ya < rows) A = src[(long)ya * KXP + k];
        if (ya + 1 < rows) B = src[(long)(ya + 1) * KXP + k];
        if (k == 0 || 2 * k == L) {
            x[P::pos_of_k(k) * BS + pr] = {A.x, B.x};   // real by symmetry: the imaginary parts are dropped
        } else {
            cplx<T> zk, zlk;
            merge_pair(A, B, zk, zlk);
            x[P::pos_of_k(k) * BS + pr] = zk;
            x[P::pos_of_k(L - k) * BS + pr] = zlk;
        }
    }
}

// kFftRowsFwd: src0 real [planes][rows][ld_src] (cols valid) -> dst0 row spectra [planes][rows][KXP]
template <typename T, int L, int NB, int NT>
__global__ __launch_bounds__(NT) void k_fft_rows_fwd(FftArgs a) {
    using P = FftPlanFor<T, L>;
    constexpr int BS = NB + 1;
    extern __shared__ __align__(16) unsigned char smem[];
    cplx<T> *x = reinterpret_cast<cplx<T> *>(smem);
    cplx<T> *tw = x + L * BS;
    const int tid = threadIdx.x, y0 = blockIdx.x * 2 * NB;
    const long plane = blockIdx.y;
    make_twiddles<T, L>(tw, tid, NT);
    const T *src = static_cast<const T *>(a.src0) + plane * a.ps_src;
    T *xr = reinterpret_cast<T *>(x);
    for (int idx = tid; idx < 2 * NB * L; idx += NT) {
        const int r = idx / L, xx = idx - r * L, y = y0 + r;
        T v = 0;
        if (y < a.rows && xx < a.cols) v = src[(long)y * a.ld_src + xx];
        xr[(xx * BS + (r >> 1)) * 2 + (r & 1)] = v;
    }
    __syncthreads();
    tile_fwd<T, P, NB, BS, NT>(x, tw, tid);
    store_rows_split<T, P, NB, BS, NT>(x, static_cast<cplx<T> *>(a.dst0) + plane * a.ps_dst, y0, a.rows, a.KXP, tid);
}

// kFftRowsInv  (MODE 0): src0 row spectra -> dst0 real [planes][rows][ld_dst], columns [xoff, xoff+cols)
// kFftRowsInv2 (MODE 1): src0, src1 -> dst0, dst1 likewise
// kFftRowsMu   (MODE 2): src0 = neg, src1 = pos row spectra; dst0 = H real (in/out, ps_src/ld_dst/cols describe it);
//                        H = (H*neg)/(pos+reg)  (TransformInvariantNMF.py:232-235), then dst1 = row spectra of the new H
template <typename T, int L, int NB, int NT, int MODE>
__global__ __launch_bounds__(NT) void k_fft_rows_inv(FftArgs a) {
    using P = FftPlanFor<T, L>;
    constexpr int BS = NB + 1;
    extern __shared__ __align__(16) unsigned char smem[];
    cplx<T> *x = reinterpret_cast<cplx<T> *>(smem);
    cplx<T> *xb = x + L * BS;
    cplx<T> *tw = MODE == 0 ? xb : xb + L * BS;
    const int tid = threadIdx.x, y0 = blockIdx.x * 2 * NB;
    const long plane = blockIdx.y;
    make_twiddles<T, L>(tw, tid, NT);
    const long tin = plane * ((long)a.rows * a.KXP);
    load_rows_merge<T, P, NB, BS, NT>(x, static_cast<const cplx<T> *>(a.src0) + tin, y0, a.rows, a.KXP, tid);
    if (MODE != 0)
        load_rows_merge<T, P, NB, BS, NT>(xb, static_cast<const cplx<T> *>(a.src1) + tin, y0, a.rows, a.KXP, tid);
    __syncthreads();
    tile_inv<T, P, NB, BS, NT>(x, tw, tid);
    if (MODE != 0) tile_inv<T, P, NB, BS, NT>(xb, tw, tid);
    T *xr = reinterpret_cast<T *>(x);
    T *xbr = reinterpret_cast<T *>(xb);
    if (MODE == 0 || MODE == 1) {
        T *d0 = static_cast<T *>(a.dst0) + plane * a.ps_dst;
        T *d1 = MODE == 1 ? static_cast<T *>(a.dst1) + plane * a.ps_dst : nullptr;
        for (int idx = tid; idx < 2 * NB * a.cols; idx += NT) {
            const int r = idx / a.cols, xc = idx - r * a.cols, y = y0 + r;
            if (y >= a.rows) continue;
            const int off = ((xc + a.xoff) * BS + (r >> 1)) * 2 + (r & 1);
            d0[(long)y * a.ld_dst + xc] = xr[off];
            if (MODE == 1) d1[(long)y * a.ld_dst + xc] = xbr[off];
        }
    } else {
        T *Hp = static_cast<T *>(a.dst0) + plane * a.ps_src;
        const T reg = (T)a.reg;
        for (int idx = tid; idx < 2 * NB * L; idx += NT) {
            const int r = idx / L, xx = idx - r * L, y = y0 + r;
            const int off = (xx * BS + (r >> 1)) * 2 + (r & 1);
            T hn = 0;
            if (y < a.rows && xx < a.cols) {
                const T h = Hp[(long)y * a.ld_dst + xx];
                hn = (h * xr[off]) / (xbr[off] + reg);
                Hp[(long)y * a.ld_dst + xx] = hn;
            }
            xr[off] = hn;
        }
        __syncthreads();
        tile_fwd<T, P, NB, BS, NT>(x, tw, tid);
        store_rows_split<T, P, NB, BS, NT>(x, static_cast<cplx<T> *>(a.dst1) + plane * a.ps_dst, y0, a.rows, a.KXP,
                                           tid);
    }
}

// ---- column kernels ---------------------------------------------------------------------------------------------

// tile <- rows [0, rows) x columns [kx0, kx0+16) of one plane of row spectra; zero elsewhere
template <typename T, int L, int NT>
__device__ __forceinline__ void load_col_tile(cplx<T> *x, const cplx<T> *src, int rows, int KXP, int KX, int kx0,
                                              int tid) {
    for (int idx = tid; idx < L * kColTile; idx += NT) {
        const int y = idx >> 4, col = idx & 15, kx = kx0 + col;
        cplx<T> v = {0, 0};
        if (y < rows && kx < KX) v = src[(long)y * KXP + kx];
        x[y * kColBS + col] = v;
    }
}

// rows [yoff, yoff+rows) of the tile -> one plane of row spectra
template <typename T, int NT>
__device__ __forceinline__ void store_col_tile_rows(const cplx<T> *x, cplx<T> *dst, int rows, int yoff, int KXP, int KX,
                                                    int kx0, int tid) {
    for (int idx = tid; idx < rows * kColTile; idx += NT) {
        const int y = idx >> 4, col = idx & 15, kx = kx0 + col;
        if (kx < KX) dst[(long)y * KXP + kx] = x[(y + yoff) * kColBS + col];
    }
}

// kFftColsFwd: src0 row spectra [planes][rows][KXP] -> dst0 full spectra [planes][L][KXP]
template <typename T, int L, int NT>
__global__ __launch_bounds__(NT) void k_fft_cols_fwd(FftArgs a) {
    using P = FftPlanFor<T, L>;
    extern __shared__ __align__(16) unsigned char smem[];
    cplx<T> *x = reinterpret_cast<cplx<T> *>(smem);
    cplx<T> *tw = x + L * kColBS;
    const int tid = threadIdx.x, kx0 = blockIdx.x * kColTile;
    const long plane = blockIdx.y;
    make_twiddles<T, L>(tw, tid, NT);
    load_col_tile<T, L, NT>(x, static_cast<const cplx<T> *>(a.src0) + plane * ((long)a.rows * a.KXP), a.rows, a.KXP,
                            a.KX, kx0, tid);
    __syncthreads();
    tile_fwd<T, P, kColTile, kColBS, NT>(x, tw, tid);
    store_col_tile_rows<T, NT>(x, static_cast<cplx<T> *>(a.dst0) + plane * ((long)L * a.KXP), L, 0, a.KXP, a.KX, kx0,
                               tid);
}

// kFftColsInv: src0 full spectra -> dst0 row spectra [planes][rows][KXP] holding transform rows [yoff, yoff+rows)
template <typename T, int L, int NT>
__global__ __launch_bounds__(NT) void k_fft_cols_inv(FftArgs a) {
    using P = FftPlanFor<T, L>;
    extern __shared__ __align__(16) unsigned char smem[];
    cplx<T> *x = reinterpret_cast<cplx<T> *>(smem);
    cplx<T> *tw = x + L * kColBS;
    const int tid = threadIdx.x, kx0 = blockIdx.x * kColTile;
    const long plane = blockIdx.y;
    make_twiddles<T, L>(tw, tid, NT);
    load_col_tile<T, L, NT>(x, static_cast<const cplx<T> *>(a.src0) + plane * ((long)L * a.KXP), L, a.KXP, a.KX, kx0,
                            tid);
    __syncthreads();
    tile_inv<T, P, kColTile, kColBS, NT>(x, tw, tid);
    store_col_tile_rows<T, NT>(x, static_cast<cplx<T> *>(a.dst0) + plane * ((long)a.rows * a.KXP), a.rows, a.yoff,
                               a.KXP, a.KX, kx0, tid);
}

// kFftContractR: R^[n,c,f] = sum_m H^[n,m,f] W^[m,c,f]   (reconstruct, NumPy.py:122-132 in the frequency domain)
//   src0 = row spectra of H [N*M][Hy][KXP], src1 = W spectra [M*C][L][KXP], dst0 = R spectra [N*C][L][KXP]
//   grid (N, tiles, channel groups of CG)
template <typename T, int L, int NT, int CG>
__global__ __launch_bounds__(NT) void k_fft_contract_R(FftArgs a) {
    using P = FftPlanFor<T, L>;
    constexpr int E = L * kColTile / NT;
    extern __shared__ __align__(16) unsigned char smem[];
    cplx<T> *x = reinterpret_cast<cplx<T> *>(smem);
    cplx<T> *tw = x + L * kColBS;
    const int tid = threadIdx.x, n = blockIdx.x, kx0 = blockIdx.y * kColTile, c0 = blockIdx.z * CG;
    make_twiddles<T, L>(tw, tid, NT);
    const cplx<T> *Tsrc = static_cast<const cplx<T> *>(a.src0);
    const cplx<T> *SW = static_cast<const cplx<T> *>(a.src1);
    const long tplane = (long)a.Hy * a.KXP, splane = (long)L * a.KXP;
    cplx<T> acc[CG][E];
#pragma unroll
    for (int cc = 0; cc < CG; ++cc)
#pragma unroll
        for (int e = 0; e < E; ++e) acc[cc][e] = {0, 0};
    for (int m = 0; m < a.M; ++m) {
        load_col_tile<T, L, NT>(x, Tsrc + ((long)n * a.M + m) * tplane, a.Hy, a.KXP, a.KX, kx0, tid);
        __syncthreads();
        tile_fwd<T, P, kColTile, kColBS, NT>(x, tw, tid);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int idx = tid + e * NT, pos = idx >> 4, col = idx & 15, kx = kx0 + col;
            const cplx<T> h = x[pos * kColBS + col];
            if (kx < a.KX) {
#pragma unroll
                for (int cc = 0; cc < CG; ++cc)
                    if (c0 + cc < a.C)
                        cfma(acc[cc][e], h, SW[((long)m * a.C + c0 + cc) * splane + (long)pos * a.KXP + kx]);
            }
        }
        __syncthreads();
    }
    cplx<T> *SR = static_cast<cplx<T> *>(a.dst0);
#pragma unroll
    for (int cc = 0; cc < CG; ++cc) {
        if (c0 + cc >= a.C) continue;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int idx = tid + e * NT, pos = idx >> 4, col = idx & 15, kx = kx0 + col;
            if (kx < a.KX) SR[((long)n * a.C + c0 + cc) * splane + (long)pos * a.KXP + kx] = acc[cc][e];
        }
    }
}

// kFftGradH: neg^[n,m,f] = sum_c V^[n,c,f] Wf^[m,c,f], pos^ likewise with R^ (NumPy.py:93-120 in the frequency domain;
//   Wf = W flipped along the shift axes), inverse transform along y, rows [0, Hy) kept.
//   src0 = V spectra [N*C][L][KXP], src1 = R spectra, src2 = Wf spectra [M*C][L][KXP];
//   dst0, dst1 = row spectra of neg, pos [(n-n0)*M+m][Hy][KXP];  grid (samples of the window, tiles)
template <typename T, int L, int NT>
__global__ __launch_bounds__(NT) void k_fft_grad_H(FftArgs a) {
    using P = FftPlanFor<T, L>;
    constexpr int E = L * kColTile / NT;
    extern __shared__ __align__(16) unsigned char smem[];
    cplx<T> *x = reinterpret_cast<cplx<T> *>(smem);
    cplx<T> *tw = x + L * kColBS;
    const int tid = threadIdx.x, nl = blockIdx.x, n = a.n0 + nl, kx0 = blockIdx.y * kColTile;
    make_twiddles<T, L>(tw, tid, NT);
    const cplx<T> *SV = static_cast<const cplx<T> *>(a.src0);
    const cplx<T> *SR = static_cast<const cplx<T> *>(a.src1);
    const cplx<T> *SWf = static_cast<const cplx<T> *>(a.src2);
    cplx<T> *Tn = static_cast<cplx<T> *>(a.dst0);
    cplx<T> *Tp = static_cast<cplx<T> *>(a.dst1);
    const long tplane = (long)a.Hy * a.KXP, splane = (long)L * a.KXP;
    __syncthreads();
    for (int m = 0; m < a.M; ++m) {
        cplx<T> ps[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int idx = tid + e * NT, pos = idx >> 4, col = idx & 15, kx = kx0 + col;
            cplx<T> ng = {0, 0};
            ps[e] = {0, 0};
            if (kx < a.KX) {
                for (int c = 0; c < a.C; ++c) {
                    const long o = (long)pos * a.KXP + kx;
                    const cplx<T> w = SWf[((long)m * a.C + c) * splane + o];
                    cfma(ng, SV[((long)n * a.C + c) * splane + o], w);
                    cfma(ps[e], SR[((long)n * a.C + c) * splane + o], w);
                }
            }
            x[pos * kColBS + col] = ng;
        }
        __syncthreads();
        tile_inv<T, P, kColTile, kColBS, NT>(x, tw, tid);
        store_col_tile_rows<T, NT>(x, Tn + ((long)nl * a.M + m) * tplane, a.Hy, 0, a.KXP, a.KX, kx0, tid);
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int idx = tid + e * NT, pos = idx >> 4, col = idx & 15;
            x[pos * kColBS + col] = ps[e];
        }
        __syncthreads();
        tile_inv<T, P, kColTile, kColBS, NT>(x, tw, tid);
        store_col_tile_rows<T, NT>(x, Tp + ((long)nl * a.M + m) * tplane, a.Hy, 0, a.KXP, a.KX, kx0, tid);
        __syncthreads();
    }
}

// kFftGradW: Gneg^[m,c,f] = sum_n H^[n,m,f] conj(V^[n,c,f]), Gpos^ with R^  (NumPy.py:69-91 in the frequency domain)
//   src0 = row spectra of H [N*M][Hy][KXP], src1 = V spectra, src2 = R spectra;
//   dst0, dst1 = partial spectra [group][M*C][L][KXP];  grid (M, tiles, groups * channel groups of CG)
template <typename T, int L, int NT, int CG>
__global__ __launch_bounds__(NT) void k_fft_grad_W(FftArgs a) {
    using P = FftPlanFor<T, L>;
    constexpr int E = L * kColTile / NT;
    extern __shared__ __align__(16) unsigned char smem[];
    cplx<T> *x = reinterpret_cast<cplx<T> *>(smem);
    cplx<T> *tw = x + L * kColBS;
    const int tid = threadIdx.x, m = blockIdx.x, kx0 = blockIdx.y * kColTile;
    const int grp = blockIdx.z % a.ngroups, c0 = (blockIdx.z / a.ngroups) * CG;
    make_twiddles<T, L>(tw, tid, NT);
    const cplx<T> *Tsrc = static_cast<const cplx<T> *>(a.src0);
    const cplx<T> *SV = static_cast<const cplx<T> *>(a.src1);
    const cplx<T> *SR = static_cast<const cplx<T> *>(a.src2);
    const long tplane = (long)a.Hy * a.KXP, splane = (long)L * a.KXP;
    cplx<T> an[CG][E], ap[CG][E];
#pragma unroll
    for (int cc = 0; cc < CG; ++cc)
#pragma unroll
        for (int e = 0; e < E; ++e) {
            an[cc][e] = {0, 0};
            ap[cc][e] = {0, 0};
        }
    const int nbeg = grp * a.nper, nend = min(a.N, nbeg + a.nper);
    for (int n = nbeg; n < nend; ++n) {
        load_col_tile<T, L, NT>(x, Tsrc + ((long)n * a.M + m) * tplane, a.Hy, a.KXP, a.KX, kx0, tid);
        __syncthreads();
        tile_fwd<T, P, kColTile, kColBS, NT>(x, tw, tid);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int idx = tid + e * NT, pos = idx >> 4, col = idx & 15, kx = kx0 + col;
            const cplx<T> h = x[pos * kColBS + col];
            if (kx < a.KX) {
#pragma unroll
                for (int cc = 0; cc < CG; ++cc)
                    if (c0 + cc < a.C) {
                        const long o = ((long)n * a.C + c0 + cc) * splane + (long)pos * a.KXP + kx;
                        cfmac(an[cc][e], h, SV[o]);
                        cfmac(ap[cc][e], h, SR[o]);
                    }
            }
        }
        __syncthreads();
    }
    cplx<T> *Gn = static_cast<cplx<T> *>(a.dst0) + (long)grp * a.M * a.C * splane;
    cplx<T> *Gp = static_cast<cplx<T> *>(a.dst1) + (long)grp * a.M * a.C * splane;
#pragma unroll
    for (int cc = 0; cc < CG; ++cc) {
        if (c0 + cc >= a.C) continue;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int idx = tid + e * NT, pos = idx >> 4, col = idx & 15, kx = kx0 + col;
            if (kx < a.KX) {
                const long o = ((long)m * a.C + c0 + cc) * splane + (long)pos * a.KXP + kx;
                Gn[o] = an[cc][e];
                Gp[o] = ap[cc][e];
            }
        }
    }
}

// ---- launcher for one (length, dtype) -----------------------------------------------------------------------------

template <typename K>
int set_lds_limit(K kernel, size_t bytes) {
    if (bytes > 64 * 1024) TNMF_HIP_TRY(hipFuncSetAttribute((const void *)kernel,
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return TNMF_OK;
}

#define TNMF_FFT_LAUNCH(kernel, grid, threads, lds)                         \
    do {                                                                    \
        static bool attr_done = false;                                      \
        if (!attr_done) {                                                   \
            const int _rc = set_lds_limit(kernel, (lds));                   \
            if (_rc != TNMF_OK) return _rc;                                 \
            attr_done = true;                                               \
        }                                                                   \
        hipLaunchKernelGGL(kernel, grid, dim3(threads), (lds), s, *a);      \
        TNMF_LAUNCH_CHECK();                                                \
        return TNMF_OK;                                                     \
    } while (0)

template <typename T, int L>
int fft_run_typed(int op, const FftArgs *a, hipStream_t s) {
    using Cfg = LenCfg<L>;
    constexpr int NB = Cfg::row_pairs, NTR = Cfg::row_threads, NTC = Cfg::col_threads;
    constexpr size_t row_tile = (size_t)L * (NB + 1) * sizeof(cplx<T>), tw_bytes = (size_t)L * sizeof(cplx<T>);
    constexpr size_t col_lds = (size_t)L * kColBS * sizeof(cplx<T>) + tw_bytes;
    if (row_tile * 2 + tw_bytes > 160 * 1024 || col_lds > 160 * 1024) return TNMF_E_UNSUPPORTED;
    const dim3 rgrid((unsigned)cdiv(a->rows, 2 * NB), (unsigned)a->planes);
    const unsigned tiles = (unsigned)cdiv(a->KX, kColTile);
    switch (op) {
        case kFftRowsFwd: TNMF_FFT_LAUNCH((k_fft_rows_fwd<T, L, NB, NTR>), rgrid, NTR, row_tile + tw_bytes);
        case kFftRowsInv: TNMF_FFT_LAUNCH((k_fft_rows_inv<T, L, NB, NTR, 0>), rgrid, NTR, row_tile + tw_bytes);
        case kFftRowsInv2: TNMF_FFT_LAUNCH((k_fft_rows_inv<T, L, NB, NTR, 1>), rgrid, NTR, 2 * row_tile + tw_bytes);
        case kFftRowsMu: TNMF_FFT_LAUNCH((k_fft_rows_inv<T, L, NB, NTR, 2>), rgrid, NTR, 2 * row_tile + tw_bytes);
        case kFftColsFwd:
            TNMF_FFT_LAUNCH((k_fft_cols_fwd<T, L, NTC>), dim3(tiles, (unsigned)a->planes), NTC, col_lds);
        case kFftColsInv:
            TNMF_FFT_LAUNCH((k_fft_cols_inv<T, L, NTC>), dim3(tiles, (unsigned)a->planes), NTC, col_lds);
        case kFftContractR:
            if (a->C == 1)
                TNMF_FFT_LAUNCH((k_fft_contract_R<T, L, NTC, 1>), dim3((unsigned)a->N, tiles, 1), NTC, col_lds);
            TNMF_FFT_LAUNCH((k_fft_contract_R<T, L, NTC, Cfg::cg_R>),
                            dim3((unsigned)a->N, tiles, (unsigned)cdiv(a->C, Cfg::cg_R)), NTC, col_lds);
        case kFftGradH:
            TNMF_FFT_LAUNCH((k_fft_grad_H<T, L, NTC>), dim3((unsigned)a->planes, tiles), NTC, col_lds);
        case kFftGradW:
            if (a->C == 1)
                TNMF_FFT_LAUNCH((k_fft_grad_W<T, L, NTC, 1>), dim3((unsigned)a->M, tiles, (unsigned)a->ngroups), NTC,
                                col_lds);
            TNMF_FFT_LAUNCH((k_fft_grad_W<T, L, NTC, Cfg::cg_W>),
                            dim3((unsigned)a->M, tiles, (unsigned)(a->ngroups * cdiv(a->C, Cfg::cg_W))), NTC, col_lds);
        default: break;
    }
    return TNMF_E_UNSUPPORTED;
}

}  // namespace
