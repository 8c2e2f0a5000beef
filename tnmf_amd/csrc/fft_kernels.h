// fft_kernels.h -- the kernels of the FFT family, templated on the transform length; fft_len.hip instantiates them
// for one length per object file.  See fft.h for the formulation and fft_engine.h for the tile transform.
//
// Array kinds (T = float | double, c = cplx<T>):
//   real planes   [plane][rows][ld]                         V, R, H, W (C-contiguous as handed over by the caller)
//   row spectra   [plane][rows][KXP]  c, kx natural order   "T": transform along x only, one line per data row
//   full spectra  [plane][L][KXP]     c, ky digit-reversed  "S": what the pointwise contractions work on
// A row kernel works on NB row pairs (two real rows = one complex sequence), a column kernel on a tile of 16 kx.
#pragma once
#include <atomic>
#include "fft.h"
#include "fft_engine.h"
#include "fft_len_twiddles.h"

namespace {

// A column kernel works on a tile of LenCfg<L>::col_tile kx columns; its LDS position stride is col_tile + 1 (odd:
// spreads the stage-3 blocks over the banks).

template <int L>
struct LenCfg {
    static constexpr int col_tile = L > 384 ? 8 : 16;   // kx columns per column-kernel tile (a power of two); the long
                                                        // transforms take narrow tiles to keep the per-thread share at 9
    // (270 and 540: 270 * 16 = 540 * 8 = 4320 tile elements = 2^5 * 3^3 * 5: 480 threads take 9 each)
    static constexpr int col_threads = (L == 270 || L == 540) ? 480 : (L > 192 ? 512 : 256);
    static_assert((L * col_tile) % col_threads == 0 && (L * 16) % col_threads == 0, "whole tile elements per thread");
    static constexpr int col_elems = L * col_tile / col_threads;   // tile elements per thread
    static constexpr int row_pairs = L > 288 ? 8 : 16;
    static constexpr int row_threads = 256;
    static constexpr int mu_pairs = L > 288 ? 4 : 8;   // the fused update kernel keeps its tile small: more blocks per CU
    static constexpr int mu_threads = L / 2 + 1 > 256 ? 512 : 256;
    // channels per pass of the contraction kernels (accumulators live in registers)
};

// tile width of a column kernel: WIDE kernels (the H-gradient kernel, which stores two activation-sized streams and
// needs whole 128-byte segments for that) always take 16 columns
template <int L, bool WIDE>
struct ColTile {
    static constexpr int v = WIDE ? 16 : LenCfg<L>::col_tile;
};

// the twiddles exp(-2 pi i t / L) of a workgroup's transforms: copied into LDS from the generated table of this length
// (fft_len_twiddles.h; the double sincospi per entry this replaces was a seventh of the vector instructions of the row
// transform, and sat in front of its loads)
template <typename T, int L>
__device__ __forceinline__ void make_twiddles(cplx<T> *tw, int tid, int nt) {
    static_assert(sizeof(tnmf_len_twiddles) == sizeof(double) * 2 * L, "the table of this translation unit's length");
    for (int t = tid; t < L; t += nt) tw[t] = {(T)tnmf_len_twiddles[t][0], (T)tnmf_len_twiddles[t][1]};
}

// all NB sequences of the tile, forward; the caller has synchronised the tile, the function ends synchronised
template <typename T, typename P, int NB, int BS, int NT>
__device__ __forceinline__ void tile_fwd(cplx<T> *x, const cplx<T> *tw, int tid) {
#pragma unroll 1
    for (int t = tid; t < NB * P::tasks1; t += NT) P::template fwd1<BS>(x + (t % NB), tw, t / NB);
    __syncthreads();
#pragma unroll 1
    for (int t = tid; t < NB * P::tasks2; t += NT) P::template fwd2<BS>(x + (t % NB), tw, t / NB);
    __syncthreads();
#pragma unroll 1
    for (int t = tid; t < NB * P::tasks3; t += NT) P::template fwd3<BS>(x + (t % NB), t / NB);
    __syncthreads();
}

template <typename T, typename P, int NB, int BS, int NT>
__device__ __forceinline__ void tile_inv(cplx<T> *x, const cplx<T> *tw, int tid) {
#pragma unroll 1
    for (int t = tid; t < NB * P::tasks3; t += NT) P::template inv3<BS>(x + (t % NB), t / NB);
    __syncthreads();
#pragma unroll 1
    for (int t = tid; t < NB * P::tasks2; t += NT) P::template inv2<BS>(x + (t % NB), tw, t / NB);
    __syncthreads();
#pragma unroll 1
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
// Thread mapping: for the real data a thread owns column x = tid (+ NT, ...) for all 2*NB rows -- those tile elements
// are consecutive floats, so all addresses are a per-thread base plus an immediate; for the spectra it owns frequency
// kx = tid (+ NT, ...) for all row pairs, with its two digit-reversed tile positions computed once.
template <typename T, int L, int NB, int NT>
__global__ __launch_bounds__(NT) void k_fft_rows_fwd(FftArgs a) {
    using P = FftPlanFor<T, L>;
    constexpr int BS = NB + 1, KX = L / 2 + 1, R2 = 2 * NB;
    constexpr int XS = (L + NT - 1) / NT, KS = (KX + NT - 1) / NT;
    extern __shared__ __align__(16) unsigned char smem[];
    cplx<T> *x = reinterpret_cast<cplx<T> *>(smem);
    cplx<T> *tw = x + L * BS;
    const int tid = threadIdx.x, y0 = blockIdx.x * R2;
    const long plane = blockIdx.y;
    make_twiddles<T, L>(tw, tid, NT);
    const T *src = static_cast<const T *>(a.src0) + plane * a.ps_src;
    T *xr = reinterpret_cast<T *>(x);
    if (tid < L) {
        const int xc = min(tid, a.cols - 1);
        T v[R2];
#pragma unroll
        for (int r = 0; r < R2; ++r) v[r] = src[(long)min(y0 + r, a.rows - 1) * a.ld_src + xc];
#pragma unroll
        for (int r = 0; r < R2; ++r) xr[tid * (2 * BS) + r] = (y0 + r < a.rows && tid < a.cols) ? v[r] : (T)0;
    }
    if constexpr (XS > 2) {   // long transforms: further column slots
#pragma unroll
        for (int sl = 1; sl < XS; ++sl) {
            const int xx = tid + sl * NT;
            if (xx < L) {
                const int xc = min(xx, a.cols - 1);
                T v[R2];
#pragma unroll
                for (int r = 0; r < R2; ++r) v[r] = src[(long)min(y0 + r, a.rows - 1) * a.ld_src + xc];
#pragma unroll
                for (int r = 0; r < R2; ++r) xr[xx * (2 * BS) + r] = (y0 + r < a.rows && xx < a.cols) ? v[r] : (T)0;
            }
        }
    } else if constexpr (XS == 2) {
        // columns beyond the block size: the few that hold data (267 - 256 at config 3) are dealt over ALL threads, row
        // by row -- as a second column slot they were 2 R2 more loads for the first wave alone -- the rest is padding
        const int rem = a.cols > NT ? a.cols - NT : 0, c0 = NT + rem;
        for (int idx = tid; idx < rem * R2; idx += NT) {
            const int r = idx / rem, c = idx - r * rem;
            xr[(NT + c) * (2 * BS) + r] = y0 + r < a.rows ? src[(long)(y0 + r) * a.ld_src + NT + c] : (T)0;
        }
        for (int idx = tid; idx < (L - c0) * R2; idx += NT) {
            const int c = idx / R2, r = idx - c * R2;
            xr[(c0 + c) * (2 * BS) + r] = (T)0;
        }
    }
    __syncthreads();
    tile_fwd<T, P, NB, BS, NT>(x, tw, tid);
    cplx<T> *dst = static_cast<cplx<T> *>(a.dst0) + plane * a.ps_dst;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int k = tid + ks * NT;
        if (k < KX) {
            const int pk = P::pos_of_k(k) * BS, plk = P::pos_of_k(k == 0 ? 0 : L - k) * BS;
#pragma unroll
            for (int p = 0; p < NB; ++p) {
                const int ya = y0 + 2 * p;
                cplx<T> A, B;
                split_pair(x[pk + p], x[plk + p], A, B);
                if (ya < a.rows) dst[(long)ya * a.KXP + k] = A;
                if (ya + 1 < a.rows) dst[(long)(ya + 1) * a.KXP + k] = B;
            }
        }
    }
}

// kFftRowsInv  (MODE 0): src0 row spectra -> dst0 real [planes][rows][ld_dst], columns [xoff, xoff+cols)
// kFftRowsInv2 (MODE 1): src0, src1 -> dst0, dst1 likewise
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
    const T *xr = reinterpret_cast<const T *>(x);
    const T *xbr = reinterpret_cast<const T *>(xb);
    T *d0 = static_cast<T *>(a.dst0) + plane * a.ps_dst;
    T *d1 = MODE == 1 ? static_cast<T *>(a.dst1) + plane * a.ps_dst : nullptr;
    for (int idx = tid; idx < 2 * NB * a.cols; idx += NT) {
        const int r = idx / a.cols, xc = idx - r * a.cols, y = y0 + r;
        if (y >= a.rows) continue;
        const int off = ((xc + a.xoff) * BS + (r >> 1)) * 2 + (r & 1);
        T v = xr[off];
        if (MODE == 0 && a.clamp0) v = v < (T)0 ? (T)0 : v;
        d0[(long)y * a.ld_dst + xc] = v;
        if (MODE == 1) d1[(long)y * a.ld_dst + xc] = xbr[off];
    }
}

// kFftRowsMu: src0 = neg, src1 = pos row spectra; dst0 = H real (in/out; ps_src / ld_dst / cols describe it);
//   H = (H*neg)/(pos+reg)  (TransformInvariantNMF.py:232-235), then dst1 = row spectra of the new H.
// One LDS tile is used three times (neg, pos, new H).  Thread mapping: in the spectrum phases thread t owns the
// frequency kx = t for all row pairs (its two digit-reversed tile positions are computed once); in the real phases it
// owns the column x = t for all 2*NB rows, whose tile elements are consecutive floats, so every LDS and global
// address is a per-thread base plus a compile-time offset -- these phases used to cost more issue slots than the
// transforms.  The pos spectra and the H values are loaded before the first transform and wait in registers; columns
// beyond the block size (x >= NT) park their neg values in a small LDS stash instead.
template <typename T, int L, int NB, int NT>
__global__ __launch_bounds__(NT) void k_fft_rows_mu(FftArgs a) {
    using P = FftPlanFor<T, L>;
    constexpr int BS = NB + 1, KX = L / 2 + 1, R2 = 2 * NB;
    constexpr int XS = (L + NT - 1) / NT;             // column slots per thread (slot 0 in registers)
    constexpr int KS = (KX + NT - 1) / NT;            // frequency slots per thread
    static_assert(KS == 1, "block size must cover Lx/2+1 frequencies");
    extern __shared__ __align__(16) unsigned char smem[];
    cplx<T> *x = reinterpret_cast<cplx<T> *>(smem);
    cplx<T> *tw = x + L * BS;
    T *stash = reinterpret_cast<T *>(tw + L);         // [(x - NT)][2*NB] neg values of the columns x >= NT
    T *xr = reinterpret_cast<T *>(x);
    const int tid = threadIdx.x, y0 = blockIdx.x * R2;
    const long plane = blockIdx.y;
    make_twiddles<T, L>(tw, tid, NT);
    const long tin = plane * ((long)a.rows * a.KXP);
    const cplx<T> *sn = static_cast<const cplx<T> *>(a.src0) + tin;
    const cplx<T> *sp = static_cast<const cplx<T> *>(a.src1) + tin;
    T *Hp = static_cast<T *>(a.dst0) + plane * a.ps_src;
    const int k = tid < KX ? tid : KX - 1;            // frequency of this thread (threads >= KX idle in those phases)
    const bool kact = tid < KX;
    const int pk = P::pos_of_k(k) * BS, plk = P::pos_of_k(k == 0 ? 0 : L - k) * BS;
    const bool kself = (k == 0 || 2 * k == L);
    // neg spectra first, then pos spectra and H: unconditional loads on clamped addresses, masked at use; the neg
    // values are consumed right away, the others stay in flight under the first transform
    cplx<T> na[NB], nb[NB], pa[NB], pb[NB];
#pragma unroll
    for (int p = 0; p < NB; ++p) {
        na[p] = sn[(long)min(y0 + 2 * p, a.rows - 1) * a.KXP + k];
        nb[p] = sn[(long)min(y0 + 2 * p + 1, a.rows - 1) * a.KXP + k];
    }
#pragma unroll
    for (int p = 0; p < NB; ++p) {
        pa[p] = sp[(long)min(y0 + 2 * p, a.rows - 1) * a.KXP + k];
        pb[p] = sp[(long)min(y0 + 2 * p + 1, a.rows - 1) * a.KXP + k];
    }
    const int xc = min(tid, a.cols - 1);
    T hv[R2];
#pragma unroll
    for (int r = 0; r < R2; ++r) hv[r] = Hp[(long)min(y0 + r, a.rows - 1) * a.ld_dst + xc];
    // neg spectra -> tile
    if (kact) {
#pragma unroll
        for (int p = 0; p < NB; ++p) {
            const int ya = y0 + 2 * p;
            cplx<T> A = na[p], B = nb[p];
            if (ya >= a.rows) A = {0, 0};
            if (ya + 1 >= a.rows) B = {0, 0};
            if (kself) {
                x[pk + p] = {A.x, B.x};
            } else {
                cplx<T> zk, zlk;
                merge_pair(A, B, zk, zlk);
                x[pk + p] = zk;
                x[plk + p] = zlk;
            }
        }
    }
    __syncthreads();
    tile_inv<T, P, NB, BS, NT>(x, tw, tid);
    // neg values of this thread's column(s): tile element (x, row r) is the float xr[x * 2*BS + r]
    T ng[R2];
    if (tid < L) {
#pragma unroll
        for (int r = 0; r < R2; ++r) ng[r] = xr[tid * (2 * BS) + r];
    }
#pragma unroll
    for (int sl = 1; sl < XS; ++sl) {
        const int xx = tid + sl * NT;
        if (xx < L) {
#pragma unroll
            for (int r = 0; r < R2; ++r) stash[(xx - NT) * R2 + r] = xr[xx * (2 * BS) + r];
        }
    }
    __syncthreads();
    // pos spectra -> tile
    if (kact) {
#pragma unroll
        for (int p = 0; p < NB; ++p) {
            const int ya = y0 + 2 * p;
            cplx<T> A = pa[p], B = pb[p];
            if (ya >= a.rows) A = {0, 0};
            if (ya + 1 >= a.rows) B = {0, 0};
            if (kself) {
                x[pk + p] = {A.x, B.x};
            } else {
                cplx<T> zk, zlk;
                merge_pair(A, B, zk, zlk);
                x[pk + p] = zk;
                x[plk + p] = zlk;
            }
        }
    }
    __syncthreads();
    tile_inv<T, P, NB, BS, NT>(x, tw, tid);
    // multiplicative update in place; the new H goes back into the tile (zero outside the data)
    const T reg = (T)a.reg;
    if (tid < L) {
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            const int y = y0 + r, off = tid * (2 * BS) + r;
            T hn = 0;
            if (y < a.rows && tid < a.cols) {
                // both gradients are sums of non-negative products: what is below zero is transform rounding noise
                hn = (hv[r] * fmax(ng[r], (T)0)) / (fmax(xr[off], (T)0) + reg);
                Hp[(long)y * a.ld_dst + tid] = hn;
            }
            xr[off] = hn;
        }
    }
#pragma unroll
    for (int sl = 1; sl < XS; ++sl) {
        const int xx = tid + sl * NT;
        if (xx < L) {
#pragma unroll
            for (int r = 0; r < R2; ++r) {
                const int y = y0 + r, off = xx * (2 * BS) + r;
                T hn = 0;
                if (y < a.rows && xx < a.cols) {
                    hn = (Hp[(long)y * a.ld_dst + xx] * fmax(stash[(xx - NT) * R2 + r], (T)0)) / (fmax(xr[off], (T)0) + reg);
                    Hp[(long)y * a.ld_dst + xx] = hn;
                }
                xr[off] = hn;
            }
        }
    }
    __syncthreads();
    tile_fwd<T, P, NB, BS, NT>(x, tw, tid);
    // row spectra of the new H
    if (kact) {
        cplx<T> *dst = static_cast<cplx<T> *>(a.dst1) + plane * a.ps_dst;
#pragma unroll
        for (int p = 0; p < NB; ++p) {
            const int ya = y0 + 2 * p;
            cplx<T> A, B;
            split_pair(x[pk + p], x[plk + p], A, B);
            if (ya < a.rows) dst[(long)ya * a.KXP + k] = A;
            if (ya + 1 < a.rows) dst[(long)(ya + 1) * a.KXP + k] = B;
        }
    }
}

// ---- column kernels ---------------------------------------------------------------------------------------------

// tile <- rows [0, rows) x columns [kx0, kx0+16) of one plane of row spectra; zero elsewhere
template <typename T, int L, int NT, bool WIDE>
__device__ __forceinline__ void load_col_tile(cplx<T> *x, const cplx<T> *src, int rows, int KXP, int KX, int kx0,
                                              int tid) {
    for (int idx = tid; idx < L * ColTile<L, WIDE>::v; idx += NT) {
        const int y = idx / ColTile<L, WIDE>::v, col = idx % ColTile<L, WIDE>::v, kx = kx0 + col;
        cplx<T> v = {0, 0};
        if (y < rows && kx < KX) v = src[(long)y * KXP + kx];
        x[y * (ColTile<L, WIDE>::v + 1) + col] = v;
    }
}

// rows [yoff, yoff+rows) of the tile -> one plane of row spectra
template <typename T, int L, int NT, bool WIDE>
__device__ __forceinline__ void store_col_tile_rows(const cplx<T> *x, cplx<T> *dst, int rows, int yoff, int KXP, int KX,
                                                    int kx0, int tid) {
    for (int idx = tid; idx < rows * ColTile<L, WIDE>::v; idx += NT) {
        const int y = idx / ColTile<L, WIDE>::v, col = idx % ColTile<L, WIDE>::v, kx = kx0 + col;
        if (kx < KX) dst[(long)y * KXP + kx] = x[(y + yoff) * (ColTile<L, WIDE>::v + 1) + col];
    }
}

// (tile, plane) of a workgroup of the plain column kernels, grid (tiles, planes).  The long transforms take 8-column
// tiles: 64-byte HALF lines of the row spectra, the other half of every line belonging to the neighbouring tile -- and the
// hardware deals consecutive workgroups round-robin to the eight XCDs, so the two halves were fetched by two different L2s:
// FETCH_SIZE of k_fft_cols_fwd<float, 576, 512> read 2.6x the stream at the config-5 shard, and the calibration on exactly
// this shape (tools/probes/fetch_calib_cols.hip, profiles/r04_fetch_calib_cols.txt) shows the doubling is real and that it
// vanishes -- 1.00x, and 1291 -> 845 us for the bare reads -- when every XCD walks a CONTIGUOUS chunk of the (tile fastest)
// order: the neighbour then runs on the same XCD at about the same time and finds the line in its L2.  16-column tiles
// are whole lines and keep the natural order.
template <int CT>
__device__ __forceinline__ void col_block(int &tile, long &plane) {
    if constexpr (CT >= 16) {
        tile = blockIdx.x;
        plane = blockIdx.y;
    } else {
        long lin = (long)blockIdx.y * gridDim.x + blockIdx.x;
        const long whole = (long)gridDim.x * gridDim.y / 8 * 8;
        if (lin < whole) lin = (lin & 7) * (whole / 8) + (lin >> 3);
        tile = (int)(lin % gridDim.x);
        plane = lin / gridDim.x;
    }
}

// kFftColsFwd: src0 row spectra [planes][rows][KXP] -> dst0 full spectra [planes][L][KXP]
template <typename T, int L, int NT>
__global__ __launch_bounds__(NT) void k_fft_cols_fwd(FftArgs a) {
    using P = FftPlanFor<T, L>;
    constexpr bool WIDE = false;
    extern __shared__ __align__(16) unsigned char smem[];
    cplx<T> *x = reinterpret_cast<cplx<T> *>(smem);
    cplx<T> *tw = x + L * (ColTile<L, WIDE>::v + 1);
    int tile;
    long plane;
    col_block<ColTile<L, WIDE>::v>(tile, plane);
    const int tid = threadIdx.x, kx0 = tile * ColTile<L, WIDE>::v;
    make_twiddles<T, L>(tw, tid, NT);
    load_col_tile<T, L, NT, WIDE>(x, static_cast<const cplx<T> *>(a.src0) + plane * ((long)a.rows * a.KXP), a.rows, a.KXP,
                            a.KX, kx0, tid);
    __syncthreads();
    tile_fwd<T, P, ColTile<L, WIDE>::v, (ColTile<L, WIDE>::v + 1), NT>(x, tw, tid);
    store_col_tile_rows<T, L, NT, WIDE>(x, static_cast<cplx<T> *>(a.dst0) + plane * ((long)L * a.KXP), L, 0, a.KXP, a.KX, kx0,
                               tid);
}

// kFftColsInv: src0 full spectra -> dst0 row spectra [planes][rows][KXP] holding transform rows [yoff, yoff+rows)
template <typename T, int L, int NT>
__global__ __launch_bounds__(NT) void k_fft_cols_inv(FftArgs a) {
    using P = FftPlanFor<T, L>;
    constexpr bool WIDE = false;
    extern __shared__ __align__(16) unsigned char smem[];
    cplx<T> *x = reinterpret_cast<cplx<T> *>(smem);
    cplx<T> *tw = x + L * (ColTile<L, WIDE>::v + 1);
    int tile;
    long plane;
    col_block<ColTile<L, WIDE>::v>(tile, plane);
    const int tid = threadIdx.x, kx0 = tile * ColTile<L, WIDE>::v;
    make_twiddles<T, L>(tw, tid, NT);
    load_col_tile<T, L, NT, WIDE>(x, static_cast<const cplx<T> *>(a.src0) + plane * ((long)L * a.KXP), L, a.KXP, a.KX, kx0,
                            tid);
    __syncthreads();
    tile_inv<T, P, ColTile<L, WIDE>::v, (ColTile<L, WIDE>::v + 1), NT>(x, tw, tid);
    store_col_tile_rows<T, L, NT, WIDE>(x, static_cast<cplx<T> *>(a.dst0) + plane * ((long)a.rows * a.KXP), a.rows, a.yoff,
                               a.KXP, a.KX, kx0, tid);
}

// Register staging of a column tile: the loads are unconditional on clamped addresses (hipcc keeps them in flight
// across the transform of the previous tile); what lies outside the data is zeroed when the tile is committed to LDS.
// Addresses are a uniform plane base plus per-thread 32-bit byte offsets that do not change from tile to tile.
// per-thread part of the element offsets of a column tile: element e of the thread sits at row (tid>>4) + e*NT/16
struct ColLane {
    int y0, kxc;   // first row of the thread, clamped kx
};
template <int L, int NT, bool WIDE>
__device__ __forceinline__ ColLane col_lane(int KX, int kx0, int tid) {
    return {tid / ColTile<L, WIDE>::v, min(kx0 + (tid % ColTile<L, WIDE>::v), KX - 1)};
}

// pre[e] <- base[min(row_e, rows-1)][kx]  (base: one plane, uniform)
template <typename T, int L, int NT, bool WIDE>
__device__ __forceinline__ void fetch_col(cplx<T> *pre, const cplx<T> *base, ColLane ln, int rows, int KXP) {
    constexpr int E = L * ColTile<L, WIDE>::v / NT;
#pragma unroll
    for (int e = 0; e < E; ++e) pre[e] = base[(unsigned)(min(ln.y0 + e * (NT / ColTile<L, WIDE>::v), rows - 1) * KXP + ln.kxc)];
}

template <typename T, int L, int NT, bool WIDE>
__device__ __forceinline__ void commit_col_tile(cplx<T> *x, const cplx<T> *pre, int rows, int KX, int kx0, int tid) {
    constexpr int E = L * ColTile<L, WIDE>::v / NT;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int idx = tid + e * NT, y = idx / ColTile<L, WIDE>::v, col = idx % ColTile<L, WIDE>::v;
        cplx<T> v = pre[e];
        if (y >= rows || kx0 + col >= KX) v = {0, 0};
        x[y * (ColTile<L, WIDE>::v + 1) + col] = v;
    }
}

// kFftContractR: R^[n,c,f] = sum_m H^[n,m,f] W^[m,c,f]   (reconstruct, NumPy.py:122-132 in the frequency domain)
//   src0 = row spectra of H [N*M][Hy][KXP], src1 = W spectra [M*C][L][KXP], dst0 = R spectra [N*C][L][KXP]
//   grid (N, tiles, channel groups of CG).  The next atom's tile is fetched while the current one is transformed.
template <typename T, int L, int NT, int CG>
__global__ __launch_bounds__(NT) void k_fft_contract_R(FftArgs a) {
    using P = FftPlanFor<T, L>;
    constexpr bool WIDE = false;
    constexpr int E = L * ColTile<L, WIDE>::v / NT;
    extern __shared__ __align__(16) unsigned char smem[];
    cplx<T> *x = reinterpret_cast<cplx<T> *>(smem);
    cplx<T> *tw = x + L * (ColTile<L, WIDE>::v + 1);
    const int tid = threadIdx.x, n = blockIdx.x, kx0 = blockIdx.y * ColTile<L, WIDE>::v, c0 = blockIdx.z * CG;
    make_twiddles<T, L>(tw, tid, NT);
    const cplx<T> *Tsrc = static_cast<const cplx<T> *>(a.src0);
    const cplx<T> *SW = static_cast<const cplx<T> *>(a.src1);
    const long tplane = (long)a.Hy * a.KXP, splane = (long)L * a.KXP;
    cplx<T> acc[CG][E], pre[E];
    const ColLane ln = col_lane<L, NT, WIDE>(a.KX, kx0, tid);
#pragma unroll
    for (int cc = 0; cc < CG; ++cc)
#pragma unroll
        for (int e = 0; e < E; ++e) acc[cc][e] = {0, 0};
    fetch_col<T, L, NT, WIDE>(pre, Tsrc + (long)n * a.M * tplane, ln, a.Hy, a.KXP);
    for (int m = 0; m < a.M; ++m) {
        commit_col_tile<T, L, NT, WIDE>(x, pre, a.Hy, a.KX, kx0, tid);
        __syncthreads();
        // W spectra of this atom: the first channel is fetched under the transform, the others while the previous
        // channel is accumulated (two register buffers)
        // (only while the per-thread tile share is small: at E = 18 the buffers would spill)
        constexpr bool PIPE = E <= 12;
        cplx<T> wb[PIPE ? 2 : 1][PIPE ? E : 1];
        if (PIPE) fetch_col<T, L, NT, WIDE>(wb[0], SW + ((long)m * a.C + min(c0, a.C - 1)) * splane, ln, L, a.KXP);
        fetch_col<T, L, NT, WIDE>(pre, Tsrc + ((long)n * a.M + min(m + 1, a.M - 1)) * tplane, ln, a.Hy, a.KXP);
        tile_fwd<T, P, ColTile<L, WIDE>::v, (ColTile<L, WIDE>::v + 1), NT>(x, tw, tid);
        if constexpr (PIPE) {
#pragma unroll
            for (int cc = 0; cc < CG; ++cc) {
                if (cc + 1 < CG)
                    fetch_col<T, L, NT, WIDE>(wb[(cc + 1) & 1], SW + ((long)m * a.C + min(c0 + cc + 1, a.C - 1)) * splane, ln,
                                        L, a.KXP);
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int idx = tid + e * NT, pos = idx / ColTile<L, WIDE>::v, col = idx % ColTile<L, WIDE>::v;
                    cfma(acc[cc][e], x[pos * (ColTile<L, WIDE>::v + 1) + col], wb[cc & 1][e]);
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int idx = tid + e * NT, pos = idx / ColTile<L, WIDE>::v, col = idx % ColTile<L, WIDE>::v;
                const cplx<T> h = x[pos * (ColTile<L, WIDE>::v + 1) + col];
#pragma unroll
                for (int cc = 0; cc < CG; ++cc) {
                    const cplx<T> *b = SW + ((long)m * a.C + min(c0 + cc, a.C - 1)) * splane;
                    cfma(acc[cc][e], h, b[(unsigned)(pos * a.KXP + ln.kxc)]);
                }
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
            const int idx = tid + e * NT, pos = idx / ColTile<L, WIDE>::v, col = idx % ColTile<L, WIDE>::v, kx = kx0 + col;
            if (kx < a.KX) SR[((long)n * a.C + c0 + cc) * splane + (long)pos * a.KXP + kx] = acc[cc][e];
        }
    }
}

// kFftGradH: neg^[n,m,f] = sum_c V^[n,c,f] Wf^[m,c,f], pos^ likewise with R^ (NumPy.py:93-120 in the frequency domain;
//   Wf = W flipped along the shift axes), inverse transform along y, rows [0, Hy) kept.
//   src0 = V spectra [N*C][L][KXP], src1 = R spectra, src2 = Wf spectra [M*C][L][KXP];
//   dst0, dst1 = row spectra of neg, pos [(n-n0)*M+m][Hy][KXP];  grid (samples of the window, tiles, 2 * atom groups):
//   blockIdx.z selects neg (from V^) or pos (from R^), so that a block keeps one spectrum of its sample in registers,
//   and a group of mper atoms (the window is a few samples only, see fft.hip: the atom groups fill the chip).
//   CH = 1..4: that many channels held in registers (CH == C); CH = 0: any C, reloaded per atom.
template <typename T, int L, int NT, int CH>
__global__ __launch_bounds__(NT) void k_fft_grad_H(FftArgs a) {
    using P = FftPlanFor<T, L>;
    constexpr bool WIDE = true;
    constexpr int E = L * ColTile<L, WIDE>::v / NT, CR = CH > 0 ? CH : 1;
    extern __shared__ __align__(16) unsigned char smem[];
    cplx<T> *x = reinterpret_cast<cplx<T> *>(smem);
    cplx<T> *tw = x + L * (ColTile<L, WIDE>::v + 1);
    const int tid = threadIdx.x, nl = blockIdx.x, n = a.n0 + nl, kx0 = blockIdx.y * ColTile<L, WIDE>::v;
    make_twiddles<T, L>(tw, tid, NT);
    const int which = blockIdx.z & 1, mbeg = (blockIdx.z >> 1) * a.mper, mend = min(a.M, mbeg + a.mper);
    const cplx<T> *S = static_cast<const cplx<T> *>(which ? a.src1 : a.src0);
    const cplx<T> *SWf = static_cast<const cplx<T> *>(a.src2);
    cplx<T> *Tout = static_cast<cplx<T> *>(which ? a.dst1 : a.dst0);
    const long tplane = (long)a.Hy * a.KXP, splane = (long)L * a.KXP;
    const ColLane ln = col_lane<L, NT, WIDE>(a.KX, kx0, tid);
    cplx<T> sv[CR][E], wb[CH > 0 ? 2 : 1][CH > 0 ? E : 1];
    if (CH > 0) {
#pragma unroll
        for (int c = 0; c < CR; ++c) fetch_col<T, L, NT, WIDE>(sv[c], S + ((long)n * a.C + c) * splane, ln, L, a.KXP);
        fetch_col<T, L, NT, WIDE>(wb[0], SWf + (long)mbeg * a.C * splane, ln, L, a.KXP);
    }
    __syncthreads();
    for (int m = mbeg; m < mend; ++m) {
        if (CH > 0) {
            // channel 0 of this atom arrived under the previous inverse transform; the others are fetched while the
            // previous channel is multiplied (two register buffers)
            cplx<T> g[E];
#pragma unroll
            for (int e = 0; e < E; ++e) g[e] = {0, 0};
#pragma unroll
            for (int c = 0; c < CR; ++c) {
                if (c + 1 < CR)
                    fetch_col<T, L, NT, WIDE>(wb[(c + 1) & 1], SWf + ((long)m * a.C + c + 1) * splane, ln, L, a.KXP);
#pragma unroll
                for (int e = 0; e < E; ++e) cfma(g[e], sv[c][e], wb[c & 1][e]);
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int idx = tid + e * NT;
                x[(idx / ColTile<L, WIDE>::v) * (ColTile<L, WIDE>::v + 1) + (idx % ColTile<L, WIDE>::v)] = g[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int idx = tid + e * NT, pos = idx / ColTile<L, WIDE>::v, col = idx % ColTile<L, WIDE>::v;
                cplx<T> g = {0, 0};
                for (int c = 0; c < a.C; ++c)
                    cfma(g, (S + ((long)n * a.C + c) * splane)[(unsigned)(pos * a.KXP + ln.kxc)],
                         (SWf + ((long)m * a.C + c) * splane)[(unsigned)(pos * a.KXP + ln.kxc)]);
                x[pos * (ColTile<L, WIDE>::v + 1) + col] = g;
            }
        }
        __syncthreads();
        if (CH > 0) fetch_col<T, L, NT, WIDE>(wb[0], SWf + (long)min(m + 1, mend - 1) * a.C * splane, ln, L, a.KXP);
        tile_inv<T, P, ColTile<L, WIDE>::v, (ColTile<L, WIDE>::v + 1), NT>(x, tw, tid);
        store_col_tile_rows<T, L, NT, WIDE>(x, Tout + ((long)nl * a.M + m) * tplane, a.Hy, 0, a.KXP, a.KX, kx0, tid);
        __syncthreads();
    }
}

// kFftGradW: Gneg^[m,c,f] = sum_n H^[n,m,f] conj(V^[n,c,f]), Gpos^ with R^  (NumPy.py:69-91 in the frequency domain)
//   src0 = row spectra of H [N*M][Hy][KXP], src1 = V spectra, src2 = R spectra;
//   dst0, dst1 = partial spectra [group][M*C][L][KXP];  grid (M, tiles, groups * channel groups of CG)
template <typename T, int L, int NT, int CG>
__global__ __launch_bounds__(NT) void k_fft_grad_W(FftArgs a) {
    using P = FftPlanFor<T, L>;
    constexpr bool WIDE = false;
    constexpr int E = L * ColTile<L, WIDE>::v / NT;
    extern __shared__ __align__(16) unsigned char smem[];
    cplx<T> *x = reinterpret_cast<cplx<T> *>(smem);
    cplx<T> *tw = x + L * (ColTile<L, WIDE>::v + 1);
    const int tid = threadIdx.x, m = blockIdx.x, kx0 = blockIdx.y * ColTile<L, WIDE>::v;
    const int grp = blockIdx.z % a.ngroups, c0 = (blockIdx.z / a.ngroups) * CG;
    make_twiddles<T, L>(tw, tid, NT);
    const cplx<T> *Tsrc = static_cast<const cplx<T> *>(a.src0);
    const cplx<T> *SV = static_cast<const cplx<T> *>(a.src1);
    const cplx<T> *SR = static_cast<const cplx<T> *>(a.src2);
    const long tplane = (long)a.Hy * a.KXP, splane = (long)L * a.KXP;
    cplx<T> an[CG][E], ap[CG][E], pre[E];
#pragma unroll
    for (int cc = 0; cc < CG; ++cc)
#pragma unroll
        for (int e = 0; e < E; ++e) {
            an[cc][e] = {0, 0};
            ap[cc][e] = {0, 0};
        }
    const int nbeg = grp * a.nper, nend = min(a.N, nbeg + a.nper);
    const ColLane ln = col_lane<L, NT, WIDE>(a.KX, kx0, tid);
    if (nbeg < nend) fetch_col<T, L, NT, WIDE>(pre, Tsrc + ((long)nbeg * a.M + m) * tplane, ln, a.Hy, a.KXP);
    for (int n = nbeg; n < nend; ++n) {
        commit_col_tile<T, L, NT, WIDE>(x, pre, a.Hy, a.KX, kx0, tid);
        __syncthreads();
        constexpr bool PRE = CG == 1 && E <= 12;   // at E = 18 the staged spectra would spill
        cplx<T> v[PRE ? E : 1], r[PRE ? E : 1];
        if (PRE) {
            fetch_col<T, L, NT, WIDE>(v, SV + ((long)n * a.C + c0) * splane, ln, L, a.KXP);
            fetch_col<T, L, NT, WIDE>(r, SR + ((long)n * a.C + c0) * splane, ln, L, a.KXP);
        }
        fetch_col<T, L, NT, WIDE>(pre, Tsrc + ((long)min(n + 1, nend - 1) * a.M + m) * tplane, ln, a.Hy, a.KXP);
        tile_fwd<T, P, ColTile<L, WIDE>::v, (ColTile<L, WIDE>::v + 1), NT>(x, tw, tid);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int idx = tid + e * NT, pos = idx / ColTile<L, WIDE>::v, col = idx % ColTile<L, WIDE>::v;
            const cplx<T> h = x[pos * (ColTile<L, WIDE>::v + 1) + col];
            if (PRE) {
                cfmac(an[0][e], h, v[PRE ? e : 0]);
                cfmac(ap[0][e], h, r[PRE ? e : 0]);
            } else {
#pragma unroll
                for (int cc = 0; cc < CG; ++cc) {
                    const long pl = ((long)n * a.C + min(c0 + cc, a.C - 1)) * splane;
                    const unsigned o = (unsigned)(pos * a.KXP + ln.kxc);
                    cfmac(an[cc][e], h, (SV + pl)[o]);
                    cfmac(ap[cc][e], h, (SR + pl)[o]);
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
            const int idx = tid + e * NT, pos = idx / ColTile<L, WIDE>::v, col = idx % ColTile<L, WIDE>::v, kx = kx0 + col;
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

// the dynamic-LDS opt-in belongs to the (function, device) pair: one flag per device, not one per process
#define TNMF_FFT_LAUNCH(kernel, grid, threads, lds)                         \
    do {                                                                    \
        static std::atomic<unsigned long long> attr_done{0};                \
        int dev_ = 0;                                                       \
        TNMF_HIP_TRY(hipGetDevice(&dev_));                                  \
        const unsigned long long bit_ = 1ull << (dev_ & 63);                \
        if (!(attr_done.load(std::memory_order_acquire) & bit_)) {          \
            /* contexts on different devices may get here from different threads: setting the attribute twice is */ \
            /* harmless, losing another device's bit is not -- hence the atomic OR */                                 \
            const int _rc = set_lds_limit(kernel, (lds));                   \
            if (_rc != TNMF_OK) return _rc;                                 \
            attr_done.fetch_or(bit_, std::memory_order_release);            \
        }                                                                   \
        hipLaunchKernelGGL(kernel, grid, dim3(threads), (lds), s, *a);      \
        TNMF_LAUNCH_CHECK();                                                \
        return TNMF_OK;                                                     \
    } while (0)

template <typename T, int L>
int fft_run_typed(int op, const FftArgs *a, hipStream_t s) {
    using Cfg = LenCfg<L>;
    // (float64 tiles are twice the bytes: half the row pairs per tile for the longer transforms)
    constexpr int NB = (sizeof(T) == 8 && L > 144) ? Cfg::row_pairs / 2 : Cfg::row_pairs;
    constexpr int NTR = Cfg::row_threads, NTC = Cfg::col_threads;
    constexpr size_t row_tile = (size_t)L * (NB + 1) * sizeof(cplx<T>), tw_bytes = (size_t)L * sizeof(cplx<T>);
    constexpr size_t col_lds = (size_t)L * (LenCfg<L>::col_tile + 1) * sizeof(cplx<T>) + tw_bytes;
    constexpr size_t wide_lds = (size_t)L * 17 * sizeof(cplx<T>) + tw_bytes;
    if (row_tile * 2 + tw_bytes > 160 * 1024 || col_lds > 160 * 1024 || wide_lds > 160 * 1024) return TNMF_E_UNSUPPORTED;
    // 1-D signals: planes of ONE row each.  The row kernels take tiles of 2 NB rows of one plane, so the planes are handed
    // over as the rows of a single plane (contiguous either way: plane stride = one row).
    FftArgs flat;
    if ((op == kFftRowsFwd || op == kFftRowsInv) && a->rows == 1 && a->planes > 1) {
        flat = *a;
        flat.rows = a->planes;
        flat.planes = 1;
        a = &flat;
    }
    const dim3 rgrid((unsigned)cdiv(a->rows, 2 * NB), (unsigned)a->planes);
    const unsigned tiles = (unsigned)cdiv(a->KX, LenCfg<L>::col_tile), wide_tiles = (unsigned)cdiv(a->KX, 16);
    switch (op) {
        case kFftRowsFwd: TNMF_FFT_LAUNCH((k_fft_rows_fwd<T, L, NB, NTR>), rgrid, NTR, row_tile + tw_bytes);
        case kFftRowsInv: TNMF_FFT_LAUNCH((k_fft_rows_inv<T, L, NB, NTR, 0>), rgrid, NTR, row_tile + tw_bytes);
        case kFftRowsInv2: TNMF_FFT_LAUNCH((k_fft_rows_inv<T, L, NB, NTR, 1>), rgrid, NTR, 2 * row_tile + tw_bytes);
        case kFftRowsMu: {
            constexpr int NBM = (sizeof(T) == 8 && L > 144) ? Cfg::mu_pairs / 2 : Cfg::mu_pairs;
            const dim3 mgrid((unsigned)cdiv(a->rows, 2 * NBM), (unsigned)a->planes);
            constexpr int NTM = Cfg::mu_threads;
            constexpr size_t stash = L > NTM ? (size_t)(L - NTM) * 2 * NBM * sizeof(T) : 0;
            TNMF_FFT_LAUNCH((k_fft_rows_mu<T, L, NBM, NTM>), mgrid, NTM,
                            (size_t)L * (NBM + 1) * sizeof(cplx<T>) + tw_bytes + stash);
        }
        case kFftColsFwd:
            // (measured in round 4 and not kept: a persistent form -- a workgroup walks tiles with the next tile prefetched
            // into registers: no gain over three one-tile workgroups per CU, profiles/r04_ab_cols_persistent_vs_plain.txt --
            // and a whole-line form -- 16-column tiles transformed as two halves of eight: slower, 7.9 vs 7.55 ms for the
            // reconstruct group at the config-5 shard; with the three LDS stages removed altogether the group still takes
            // 7.0-7.2 ms, profiles/r04_ab_cols_whole_lines_and_stage_ablation.txt)
            TNMF_FFT_LAUNCH((k_fft_cols_fwd<T, L, NTC>), dim3(tiles, (unsigned)a->planes), NTC, col_lds);
        case kFftColsInv:
            TNMF_FFT_LAUNCH((k_fft_cols_inv<T, L, NTC>), dim3(tiles, (unsigned)a->planes), NTC, col_lds);
        case kFftContractR: {
            const int cg = a->C <= 3 ? a->C : 4;
            const dim3 grid((unsigned)a->N, tiles, (unsigned)cdiv(a->C, cg));
            if (cg == 1) TNMF_FFT_LAUNCH((k_fft_contract_R<T, L, NTC, 1>), grid, NTC, col_lds);
            if (cg == 2) TNMF_FFT_LAUNCH((k_fft_contract_R<T, L, NTC, 2>), grid, NTC, col_lds);
            if (cg == 3) TNMF_FFT_LAUNCH((k_fft_contract_R<T, L, NTC, 3>), grid, NTC, col_lds);
            TNMF_FFT_LAUNCH((k_fft_contract_R<T, L, NTC, 4>), grid, NTC, col_lds);
        }
        case kFftGradH: {
            const dim3 grid((unsigned)a->planes, wide_tiles, 2u * (unsigned)a->mgroups);
            if (a->C == 1) TNMF_FFT_LAUNCH((k_fft_grad_H<T, L, NTC, 1>), grid, NTC, wide_lds);
            if (a->C == 2) TNMF_FFT_LAUNCH((k_fft_grad_H<T, L, NTC, 2>), grid, NTC, wide_lds);
            if (a->C == 3) TNMF_FFT_LAUNCH((k_fft_grad_H<T, L, NTC, 3>), grid, NTC, wide_lds);
            if (a->C == 4) TNMF_FFT_LAUNCH((k_fft_grad_H<T, L, NTC, 4>), grid, NTC, wide_lds);
            TNMF_FFT_LAUNCH((k_fft_grad_H<T, L, NTC, 0>), grid, NTC, wide_lds);
        }
        case kFftGradW: {
            // channels per pass: two accumulator sets per channel live in registers
            const int cg = a->C <= 2 ? a->C : (Cfg::col_elems <= 9 ? 3 : 2);
            const dim3 grid((unsigned)a->M, tiles, (unsigned)(a->ngroups * cdiv(a->C, cg)));
            if (cg == 1) TNMF_FFT_LAUNCH((k_fft_grad_W<T, L, NTC, 1>), grid, NTC, col_lds);
            if (cg == 2) TNMF_FFT_LAUNCH((k_fft_grad_W<T, L, NTC, 2>), grid, NTC, col_lds);
            TNMF_FFT_LAUNCH((k_fft_grad_W<T, L, NTC, 3>), grid, NTC, col_lds);
        }
        default: break;
    }
    return TNMF_E_UNSUPPORTED;
}

}  // namespace
