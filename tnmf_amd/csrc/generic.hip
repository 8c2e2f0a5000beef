// generic.hip -- dtype- and shape-generic gfx950 kernels of the shift-invariant MU path (f32 and f64, 1-D and 2-D,
// any atom/channel/atom-count).  They are the f64 path (the reference's own acceptance tests are float64), the path
// for shapes the MFMA kernels do not cover, and the in-library cross-check of the MFMA kernels.
//
// Index forms (reference: /root/reference/tnmf/backends/NumPy.py):
//   reconstruct :122-132   R[n,c,y,x] = sum_m sum_{a,b} H[n,m,y+a,x+b] * W[m,c,Ay-1-a,Ax-1-b]
//   corr_W      :101-119   O[n,m,u,v] = sum_c sum_{a,b} W[m,c,a,b] * X[n,c,u+a-(Ay-1),v+b-(Ax-1)]      X in {V, R}
//   corr_H      :77-90     G[m,c,a,b] = sum_n sum_{y,x} H[n,m,y+Ay-1-a,x+Ax-1-b] * X[n,c,y,x]          X in {V, R}
#include "generic.h"

namespace {

constexpr int kBlock = 256;

struct Tile {
    int TY, TX;      // output tile (TY * TX == kBlock)
    int tiles_y, tiles_x;
};

inline Tile make_tile(int rows, int cols) {
    Tile t;
    if (rows == 1) {
        t.TY = 1;
        t.TX = kBlock;
    } else if (cols >= 32 || rows < 16) {
        t.TY = 8;
        t.TX = 32;
    } else {
        t.TY = 16;
        t.TX = 16;
    }
    t.tiles_y = cdiv(rows, t.TY);
    t.tiles_x = cdiv(cols, t.TX);
    return t;
}

// ------------------------------------------------------------------------------------------------------------
// reconstruct: one block = one (n, c, output tile); loop over atoms, H tile + halo and flipped W[m,c] in LDS.
// ------------------------------------------------------------------------------------------------------------
// (the bodies are device functions of a block index: the __global__ kernels below launch one workgroup per block, the
// persistent schedule kernel at the end of this file walks the blocks of one phase after the other)
template <typename T>
__device__ __forceinline__ void reconstruct_block(const Geo &g, const Tile &t, unsigned bid, const T *__restrict__ W,
                                                  const T *__restrict__ H, T *__restrict__ R, unsigned char *smem_raw) {
    T *Hs = reinterpret_cast<T *>(smem_raw);
    const int SH = t.TY + g.Ay - 1, SW = t.TX + g.Ax - 1;
    const int nA = g.Ay * g.Ax;
    T *Ws = Hs + SH * SW;

    const int txi = bid % t.tiles_x;
    bid /= t.tiles_x;
    const int tyi = bid % t.tiles_y;
    bid /= t.tiles_y;
    const int c = bid % g.C;
    const int n = bid / g.C;

    const int y0 = tyi * t.TY, x0 = txi * t.TX;
    const int ty = threadIdx.x / t.TX, tx = threadIdx.x % t.TX;

    T acc = T(0);
    // Staging through registers: the H tile (+ halo) and the flipped atom of the atoms AHEAD are fetched -- all loads of an
    // atom in flight at once, row and column of element tid + 256 k by constant increments -- while the current atom is
    // multiplied, and parked in LDS behind it.  (One load per loop iteration, each waited for, made small calls a chain of
    // memory latencies: 37 us for 3 samples of 32 x 32 with ten atoms -- the batches of the stochastic schedules.)
    //   small tiles (<= 3 values per thread, atoms <= 256 taps): kDeep = 4 atoms in flight (ring of register slots,
    //   the atom loop unrolled by four so that every slot index is static);
    //   larger tiles (<= kPre values per thread): one atom ahead;  beyond that: the plain loop.
    constexpr int kPre = 12, kDeep = 4, kPer = kPre / kDeep, kWPre = 4;
    const int nel = SH * SW;
    const int dr = kBlock / SW, dq = kBlock - dr * SW;
    const int r_first = (int)threadIdx.x / SW, q_first = (int)threadIdx.x - r_first * SW;
    const bool deep = nel <= kPer * kBlock && nA <= kBlock;
    const bool pre_ok = nel <= kPre * kBlock, wpre_ok = nA <= kWPre * kBlock;
    T pre[kPre], wpre[kWPre];
    // fetch atom m_ into pre[p0 .. p0 + cnt) and wpre[w0 .. w0 + wcnt)
    auto fetch = [&](int m_, int p0, int cnt, int w0, int wcnt) {
        const T *h = H + ((size_t)n * g.M + m_) * g.Hy * g.Hs;   // (rows of H may be padded: g.Hs >= g.Hx)
        const T *w_ = W + ((size_t)m_ * g.C + c) * nA;
#pragma unroll
        for (int k = 0; k < kWPre; ++k) {
            const int i = k * kBlock + threadIdx.x;
            if (k < wcnt) wpre[w0 + k] = i < nA ? w_[nA - 1 - i] : T(0);
        }
        int r = r_first, q = q_first;
#pragma unroll
        for (int k = 0; k < kPre; ++k) {
            if (k < cnt) {
                const int hy = y0 + r, hx = x0 + q;
                pre[p0 + k] =
                    (k * kBlock + (int)threadIdx.x < nel && hy < g.Hy && hx < g.Hx) ? h[(size_t)hy * g.Hs + hx] : T(0);
                r += dr;
                q += dq;
                if (q >= SW) {
                    q -= SW;
                    ++r;
                }
            }
        }
    };
    auto park = [&](int p0, int cnt, int w0, int wcnt) {
#pragma unroll
        for (int k = 0; k < kPre; ++k) {
            const int i = k * kBlock + threadIdx.x;
            if (k < cnt && i < nel) Hs[i] = pre[p0 + k];
        }
#pragma unroll
        for (int k = 0; k < kWPre; ++k) {
            const int i = k * kBlock + threadIdx.x;
            if (k < wcnt && i < nA) Ws[i] = wpre[w0 + k];   // flipped atom (fetched with the tile)
        }
    };
    auto multiply = [&]() {
        for (int a = 0; a < g.Ay; ++a) {
            const T *hr = Hs + (ty + a) * SW + tx;
            const T *wr = Ws + a * g.Ax;
            // (unrolled: eight LDS reads in flight instead of one -- the additions keep their order)
#pragma unroll 8
            for (int b = 0; b < g.Ax; ++b) acc += hr[b] * wr[b];
        }
    };
    if (deep) {
#pragma unroll
        for (int d = 0; d < kDeep; ++d)
            if (d < g.M) fetch(d, d * kPer, kPer, d, 1);
        for (int m0 = 0; m0 < g.M; m0 += kDeep) {
#pragma unroll
            for (int d = 0; d < kDeep; ++d) {
                const int m = m0 + d;
                if (m < g.M) {   // (uniform)
                    __syncthreads();
                    park(d * kPer, kPer, d, 1);
                    __syncthreads();
                    if (m + kDeep < g.M) fetch(m + kDeep, d * kPer, kPer, d, 1);   // in flight under the next atoms
                    multiply();
                }
            }
        }
    } else {
        if (pre_ok && wpre_ok) fetch(0, 0, kPre, 0, kWPre);
        for (int m = 0; m < g.M; ++m) {
            const T *h = H + ((size_t)n * g.M + m) * g.Hy * g.Hs;
            const T *w = W + ((size_t)m * g.C + c) * nA;
            __syncthreads();
            if (pre_ok && wpre_ok) {
                park(0, kPre, 0, kWPre);
            } else {
                for (int i = threadIdx.x; i < nel; i += kBlock) {
                    const int r = i / SW, q = i - r * SW;
                    const int hy = y0 + r, hx = x0 + q;
                    Hs[i] = (hy < g.Hy && hx < g.Hx) ? h[(size_t)hy * g.Hs + hx] : T(0);
                }
                for (int i = threadIdx.x; i < nA; i += kBlock) Ws[i] = w[nA - 1 - i];  // flipped atom
            }
            __syncthreads();
            if (pre_ok && wpre_ok && m + 1 < g.M) fetch(m + 1, 0, kPre, 0, kWPre);   // in flight under the multiply-adds
            multiply();
        }
    }
    const int y = y0 + ty, x = x0 + tx;
    if (y < g.Dy && x < g.Dx) R[(((size_t)n * g.C + c) * g.Dy + y) * g.Dx + x] = acc;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_reconstruct(Geo g, Tile t, const T *__restrict__ W,
                                                        const T *__restrict__ H, T *__restrict__ R) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    reconstruct_block<T>(g, t, blockIdx.x, W, H, R, smem_raw);
}

// reconstruct for SMALL calls (the batches of the stochastic schedules: three samples of 32 x 32): the kernel above would
// run a dozen workgroups, each walking all atoms one after the other -- 26 us of a chip that is otherwise idle.  Here a
// workgroup owns 2 rows x 32 columns of one (sample, channel) and its four waves take the atoms m = q, q + 4, ... each with
// an LDS tile of its own; their partial sums are added in wave order (fixed) at the end: four times the workgroups, a
// quarter of the serial chain.  Rows of H may be padded (g.Hs).
constexpr int kSmallTY = 2, kSmallTX = 32, kSmallQ = 4;

// (kSmallQ = waves per workgroup.  The launched kernel takes up to SIXTEEN -- one atom per wave, one round: with four, the
// ten atoms of the reference's mini-batch geometry were three serial rounds of load -> LDS -> 49 multiply-adds, 12 us of
// the 46 us of KERNEL time of a 47 us batch step (rocprof, round 4: the step is not launch bound -- the gaps between its
// five kernels are 2.9 us on average); the persistent schedule kernel, whose workgroups are four waves, stays at four)
template <typename T, int kSmallQ = 4>
__device__ __forceinline__ void reconstruct_small_block(const Geo &g, int tiles_y, int tiles_x, unsigned bid,
                                                        const T *__restrict__ W, const T *__restrict__ H,
                                                        T *__restrict__ R, unsigned char *smem_raw) {
    const int SH = kSmallTY + g.Ay - 1, SW = kSmallTX + g.Ax - 1, nel = SH * SW, nA = g.Ay * g.Ax;
    const int q = threadIdx.x >> 6, lane = threadIdx.x & 63;
    T *Hs = reinterpret_cast<T *>(smem_raw) + (size_t)q * (nel + nA);
    T *Ws = Hs + nel;
    T *red = reinterpret_cast<T *>(smem_raw) + (size_t)kSmallQ * (nel + nA);   // [kSmallQ][64]
    const int txi = bid % tiles_x;
    bid /= tiles_x;
    const int tyi = bid % tiles_y;
    bid /= tiles_y;
    const int c = bid % g.C, n = bid / g.C;
    const int y0 = tyi * kSmallTY, x0 = txi * kSmallTX;
    const int ty = lane / kSmallTX, tx = lane % kSmallTX;
    constexpr int kPre = 12;   // tile elements per lane in the register stage (64 lanes per tile); beyond: plain loop
    const bool pre_ok = nel <= kPre * 64 && nA <= 4 * 64;
    T acc = T(0);
    for (int m0 = 0; m0 < g.M; m0 += kSmallQ) {
        const int m = m0 + q;
        const bool has = m < g.M;   // (wave-uniform)
        const T *h = H + ((size_t)n * g.M + (has ? m : 0)) * g.Hy * g.Hs;
        const T *w = W + ((size_t)(has ? m : 0) * g.C + c) * nA;
        __syncthreads();   // every wave is done with its previous tile
        if (has) {
            if (pre_ok) {
                T pre[kPre], wpre[4];
#pragma unroll
                for (int k = 0; k < kPre; ++k) {
                    const int i = k * 64 + lane;
                    const int r = i / SW, qq = i - r * SW;
                    const int hy = y0 + r, hx = x0 + qq;
                    pre[k] = (i < nel && hy < g.Hy && hx < g.Hx) ? h[(size_t)hy * g.Hs + hx] : T(0);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = k * 64 + lane;
                    wpre[k] = i < nA ? w[nA - 1 - i] : T(0);   // flipped atom
                }
#pragma unroll
                for (int k = 0; k < kPre; ++k)
                    if (k * 64 + lane < nel) Hs[k * 64 + lane] = pre[k];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k * 64 + lane < nA) Ws[k * 64 + lane] = wpre[k];
            } else {
                for (int i = lane; i < nel; i += 64) {
                    const int r = i / SW, qq = i - r * SW;
                    const int hy = y0 + r, hx = x0 + qq;
                    Hs[i] = (hy < g.Hy && hx < g.Hx) ? h[(size_t)hy * g.Hs + hx] : T(0);
                }
                for (int i = lane; i < nA; i += 64) Ws[i] = w[nA - 1 - i];
            }
        }
        __syncthreads();
        if (has) {
            for (int a = 0; a < g.Ay; ++a) {
                const T *hr = Hs + (ty + a) * SW + tx;
                const T *wr = Ws + a * g.Ax;
#pragma unroll 8
                for (int b = 0; b < g.Ax; ++b) acc += hr[b] * wr[b];
            }
        }
    }
    __syncthreads();
    red[q * 64 + lane] = acc;
    __syncthreads();
    if (q == 0) {
        T tot = red[lane];
#pragma unroll
        for (int k = 1; k < kSmallQ; ++k) tot += red[k * 64 + lane];   // wave order: fixed
        const int y = y0 + ty, x = x0 + tx;
        if (y < g.Dy && x < g.Dx) R[(((size_t)n * g.C + c) * g.Dy + y) * g.Dx + x] = tot;
    }
}

template <typename T, int Q>
__global__ __launch_bounds__(64 * Q) void k_reconstruct_small(Geo g, int tiles_y, int tiles_x, const T *__restrict__ W,
                                                              const T *__restrict__ H, T *__restrict__ R) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    reconstruct_small_block<T, Q>(g, tiles_y, tiles_x, blockIdx.x, W, H, R, smem_raw);
}

// LDS of the small-call form with q waves per workgroup
template <typename T>
inline size_t reconstruct_small_lds(const Geo &g, int q) {
    return ((size_t)q * ((size_t)(kSmallTY + g.Ay - 1) * (kSmallTX + g.Ax - 1) + (size_t)g.Ay * g.Ax) + (size_t)q * 64) * sizeof(T);
}

// whether a reconstruct call of this geometry takes the small-call form (launch_reconstruct and the schedule kernel agree)
template <typename T>
inline bool reconstruct_is_small(const Geo &g, const Tile &t, size_t *lds_small) {
    const size_t blocks = (size_t)g.N * g.C * t.tiles_y * t.tiles_x;
    *lds_small = reconstruct_small_lds<T>(g, kSmallQ);
    return blocks < 64 && g.M >= kSmallQ && g.Dy > 1 && *lds_small <= 64 * 1024;
}

// ------------------------------------------------------------------------------------------------------------
// corr_W (H gradient): one block = one (n, m, tile of the shift plane); loop over channels, zero-padded V and R
// tiles in LDS.  FUSED: H = (H * neg) / (pos + reg) in place instead of writing neg/pos.
// ------------------------------------------------------------------------------------------------------------
template <typename T, bool FUSED>
__device__ __forceinline__ void corr_W_block(const Geo &g, const Tile &t, unsigned bid, const T *__restrict__ V,
                                             const T *__restrict__ Rr, const T *__restrict__ W, T *__restrict__ Hio,
                                             T *__restrict__ neg, T *__restrict__ pos, T reg,
                                             const T *__restrict__ extra, unsigned char *smem_raw) {
    const int SH = t.TY + g.Ay - 1, SW = t.TX + g.Ax - 1;
    const int nA = g.Ay * g.Ax;
    T *Vs = reinterpret_cast<T *>(smem_raw);
    T *Rs = Vs + SH * SW;
    T *Ws = Rs + SH * SW;

    const int txi = bid % t.tiles_x;
    bid /= t.tiles_x;
    const int tyi = bid % t.tiles_y;
    bid /= t.tiles_y;
    const int m = bid % g.M;
    const int n = bid / g.M;

    const int u0 = tyi * t.TY, v0 = txi * t.TX;
    const int ty = threadIdx.x / t.TX, tx = threadIdx.x % t.TX;

    T an = T(0), ap = T(0);
    for (int c = 0; c < g.C; ++c) {
        const T *v = V + ((size_t)n * g.C + c) * g.Dy * g.Dx;
        const T *r = Rr + ((size_t)n * g.C + c) * g.Dy * g.Dx;
        const T *w = W + ((size_t)m * g.C + c) * nA;
        __syncthreads();
        for (int i = threadIdx.x; i < SH * SW; i += kBlock) {
            const int rr = i / SW, q = i - rr * SW;
            const int y = u0 + rr - (g.Ay - 1), x = v0 + q - (g.Ax - 1);
            const bool in = (y >= 0 && y < g.Dy && x >= 0 && x < g.Dx);
            const size_t o = (size_t)y * g.Dx + x;
            Vs[i] = in ? v[o] : T(0);
            Rs[i] = in ? r[o] : T(0);
        }
        for (int i = threadIdx.x; i < nA; i += kBlock) Ws[i] = w[i];
        __syncthreads();
        for (int a = 0; a < g.Ay; ++a) {
            const T *vr = Vs + (ty + a) * SW + tx;
            const T *rr = Rs + (ty + a) * SW + tx;
            const T *wr = Ws + a * g.Ax;
#pragma unroll 8
            for (int b = 0; b < g.Ax; ++b) {
                an += wr[b] * vr[b];
                ap += wr[b] * rr[b];
            }
        }
    }
    const int u = u0 + ty, vv = v0 + tx;
    if (u < g.Hy && vv < g.Hx) {
        const size_t o = (((size_t)n * g.M + m) * g.Hy + u) * g.Hx + vv;
        if (FUSED) {
            const size_t oh = (((size_t)n * g.M + m) * g.Hy + u) * g.Hs + vv;   // (rows of H may be padded)
            const T h = Hio[oh];
            if (extra) ap += extra[oh];   // lateral inhibition terms (TransformInvariantNMF.py:253-269), laid out like H
            Hio[oh] = (h * an) / (ap + reg);
        } else {
            neg[o] = an;
            pos[o] = ap;
        }
    }
}

template <typename T, bool FUSED>
__global__ __launch_bounds__(kBlock) void k_corr_W(Geo g, Tile t, const T *__restrict__ V, const T *__restrict__ Rr,
                                                   const T *__restrict__ W, T *__restrict__ Hio,
                                                   T *__restrict__ neg, T *__restrict__ pos, T reg,
                                                   const T *__restrict__ extra) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    corr_W_block<T, FUSED>(g, t, blockIdx.x, V, Rr, W, Hio, neg, pos, reg, extra, smem_raw);
}

// ------------------------------------------------------------------------------------------------------------
// corr_H (W gradient), stage 1: grid (P, M*C).  Block p walks work items (n, tile) = p, p+P, ...; per item the H
// tile (+halo) of atom m and the V, R tiles of channel c sit in LDS; thread (group, shift) sums its shift over the
// group's pixels in T, then folds the item's partial into a double.  Output: partials[p][mc][shift][2] (double).
// ------------------------------------------------------------------------------------------------------------
constexpr int kMaxShiftsPerThread = 4;  // atoms up to 4 * 256 = 1024 shifts (e.g. 32 x 32)

template <typename T>
__device__ __forceinline__ void corr_H_block(const Geo &g, const Tile &t, int P, int p, int mc, const T *__restrict__ V,
                                             const T *__restrict__ Rr, const T *__restrict__ H,
                                             double *__restrict__ partials, unsigned char *smem_raw) {
    const int SH = t.TY + g.Ay - 1, SW = t.TX + g.Ax - 1;
    const int nA = g.Ay * g.Ax;
    const int npix = t.TY * t.TX;
    T *Hs = reinterpret_cast<T *>(smem_raw);
    T *Vs = Hs + SH * SW;
    T *Rs = Vs + npix;

    const int m = mc / g.C, c = mc % g.C;

    const int gs = nA < kBlock ? nA : kBlock;  // threads per group = shifts handled side by side
    const int G = kBlock / gs;                 // pixel groups
    const int grp = threadIdx.x / gs, sl = threadIdx.x - grp * gs;
    const bool active = grp < G;

    double accn[kMaxShiftsPerThread], accp[kMaxShiftsPerThread];
#pragma unroll
    for (int k = 0; k < kMaxShiftsPerThread; ++k) accn[k] = accp[k] = 0.0;

    const int items = g.N * t.tiles_y * t.tiles_x;
    for (int it = p; it < items; it += P) {
        int r = it;
        const int txi = r % t.tiles_x;
        r /= t.tiles_x;
        const int tyi = r % t.tiles_y;
        const int n = r / t.tiles_y;
        const int y0 = tyi * t.TY, x0 = txi * t.TX;
        const T *h = H + ((size_t)n * g.M + m) * g.Hy * g.Hs;   // (rows of H may be padded: g.Hs >= g.Hx)
        const T *v = V + ((size_t)n * g.C + c) * g.Dy * g.Dx;
        const T *rr = Rr + ((size_t)n * g.C + c) * g.Dy * g.Dx;
        __syncthreads();
        for (int i = threadIdx.x; i < SH * SW; i += kBlock) {
            const int a = i / SW, q = i - a * SW;
            const int hy = y0 + a, hx = x0 + q;
            Hs[i] = (hy < g.Hy && hx < g.Hx) ? h[(size_t)hy * g.Hs + hx] : T(0);
        }
        for (int i = threadIdx.x; i < npix; i += kBlock) {
            const int a = i / t.TX, q = i - a * t.TX;
            const int y = y0 + a, x = x0 + q;
            const bool in = (y < g.Dy && x < g.Dx);
            const size_t o = (size_t)y * g.Dx + x;
            Vs[i] = in ? v[o] : T(0);
            Rs[i] = in ? rr[o] : T(0);
        }
        __syncthreads();
        if (active) {
#pragma unroll
            for (int k = 0; k < kMaxShiftsPerThread; ++k) {
                const int s = sl + k * gs;
                if (s < nA) {
                    const int a = s / g.Ax, b = s - a * g.Ax;
                    T pn = T(0), pp = T(0);
#pragma unroll 4
                    for (int pix = grp; pix < npix; pix += G) {
                        const int y = pix / t.TX, x = pix - y * t.TX;
                        const T hv = Hs[(y + a) * SW + x + b];
                        pn += hv * Vs[pix];
                        pp += hv * Rs[pix];
                    }
                    accn[k] += (double)pn;
                    accp[k] += (double)pp;
                }
            }
        }
    }
    // fold the pixel groups (fixed order) and write this block's partial
    __syncthreads();
    double *red = reinterpret_cast<double *>(smem_raw);  // [G][nA][2], reuses the tiles
#pragma unroll
    for (int k = 0; k < kMaxShiftsPerThread; ++k) {
        const int s = sl + k * gs;
        if (active && s < nA) {
            red[((size_t)grp * nA + s) * 2 + 0] = accn[k];
            red[((size_t)grp * nA + s) * 2 + 1] = accp[k];
        }
    }
    __syncthreads();
    for (int s = threadIdx.x; s < nA; s += kBlock) {
        double sn = 0.0, sp = 0.0;
        for (int q = 0; q < G; ++q) {
            sn += red[((size_t)q * nA + s) * 2 + 0];
            sp += red[((size_t)q * nA + s) * 2 + 1];
        }
        double *out = partials + (((size_t)p * (g.M * g.C) + mc) * nA + s) * 2;
        out[0] = sn;
        out[1] = sp;
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_corr_H(Geo g, Tile t, int P, const T *__restrict__ V,
                                                   const T *__restrict__ Rr, const T *__restrict__ H,
                                                   double *__restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    corr_H_block<T>(g, t, P, blockIdx.x, blockIdx.y, V, Rr, H, partials, smem_raw);
}

}  // namespace

// stage 2 of every W gradient (generic and MFMA): sum the P partials in double in a fixed order; shift s of the
// partial layout is the *unflipped* offset (a', b') = (Ay-1-a, Ax-1-b), so the flip is a reversed linear index.
// One block = 32 consecutive outputs x 8 interleaved p-slices (slice q sums p = q, q+8, ... in order), then the 8
// slice sums are folded in order: the result does not depend on scheduling.
constexpr int kFinOut = 32, kFinSlices = 8;

template <typename T>
__global__ __launch_bounds__(kFinOut * kFinSlices) void k_corr_H_finalize(int MC, int nA, int P,
                                                                            const double *__restrict__ partials,
                                                                            T *__restrict__ neg, T *__restrict__ pos) {
    __shared__ double sh[kFinSlices][kFinOut][2];
    const int il = threadIdx.x % kFinOut, q = threadIdx.x / kFinOut;
    const int i = blockIdx.x * kFinOut + il;
    const int total = MC * nA;
    double sn = 0.0, sp = 0.0;
    if (i < total) {
        const double2 *in = reinterpret_cast<const double2 *>(partials) + i;
        for (int p = q; p < P; p += kFinSlices) {
            const double2 v = in[(size_t)p * total];
            sn += v.x;
            sp += v.y;
        }
    }
    sh[q][il][0] = sn;
    sh[q][il][1] = sp;
    __syncthreads();
    if (q == 0 && i < total) {
        double tn = 0.0, tp = 0.0;
#pragma unroll
        for (int k = 0; k < kFinSlices; ++k) {
            tn += sh[k][il][0];
            tp += sh[k][il][1];
        }
        const int mc = i / nA, s = i - mc * nA;
        const size_t o = (size_t)mc * nA + (nA - 1 - s);
        neg[o] = (T)tn;
        pos[o] = (T)tp;
    }
}

// ------------------------------------------------------------------------------------------------------------
// elementwise / small kernels
// ------------------------------------------------------------------------------------------------------------
namespace {

template <typename T>
__global__ void k_mu_update(T *__restrict__ arr, const T *__restrict__ neg, T *__restrict__ pos, T reg, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const T p = pos[i] + reg;
        pos[i] = p;
        arr[i] = (arr[i] * neg[i]) / p;
    }
}

// out[i] = ((parts[0][i] + parts[1][i]) + parts[2][i]) + ...: the sum of `n_parts` buffers in their order, in T
template <typename T>
__global__ void k_sum_parts(const T *__restrict__ parts, int n_parts, size_t n, T *__restrict__ out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        T acc = parts[i];
        for (int r = 1; r < n_parts; ++r) acc = acc + parts[(size_t)r * n + i];
        out[i] = acc;
    }
}

// acc = a * acc + b * g  (a == 0: acc = b * g, whatever acc held): the gradient accumulators of the mini-batch schedules
// (TransformInvariantNMF.py:444-455: `acc *= (1 - lambda); acc += lambda * g`, or `acc += g`)
template <typename T>
__global__ void k_axpby(T *__restrict__ acc, const T *__restrict__ g, T a, T b, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const T t = b == T(1) ? g[i] : b * g[i];
        acc[i] = a == T(0) ? t : (a == T(1) ? acc[i] + t : a * acc[i] + t);
    }
}

template <typename T>
__device__ double block_sum(double v, double *sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double tot = 0.0;
    const int nw = (blockDim.x + 63) >> 6;
    for (int w = 0; w < nw; ++w) tot += sh[w];  // every thread, same order
    return tot;
}

// W[mc, :] = (W * neg) / (pos + eps) if APPLY, then W[mc, :] /= sum.  One block per (m, c) row.
template <typename T, bool APPLY>
__device__ __forceinline__ void apply_normalize_row(int nA, unsigned row, T *__restrict__ W, const T *__restrict__ neg,
                                                    T *__restrict__ pos, T eps) {
    __shared__ double sh[kBlock / 64];
    const size_t base = (size_t)row * nA;
    double part = 0.0;
    for (int i = threadIdx.x; i < nA; i += kBlock) {
        T w = W[base + i];
        if (APPLY) {
            const T p = pos[base + i] + eps;
            pos[base + i] = p;
            w = (w * neg[base + i]) / p;
            W[base + i] = w;
        }
        part += (double)w;
    }
    const T tot = (T)block_sum<T>(part, sh);
    for (int i = threadIdx.x; i < nA; i += kBlock) W[base + i] = W[base + i] / tot;
}

template <typename T, bool APPLY>
__global__ __launch_bounds__(kBlock) void k_apply_normalize_W(int nA, T *__restrict__ W, const T *__restrict__ neg,
                                                              T *__restrict__ pos, T eps) {
    apply_normalize_row<T, APPLY>(nA, blockIdx.x, W, neg, pos, eps);
}

// W gradient of a mini-batch step behind the split-K kernel, in one launch: fixed-order sum of the P partials of row
// (m, c) in double, flip to the reference's orientation, acc = a * acc + b * g, and -- when the W update follows at once
// (ASG / ASAG: every batch) -- W = W * acc_neg / (acc_pos + eps), normalised.  One workgroup per (m, c) row.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_finalize_blend_apply(int MC, int nA, int P, const double *__restrict__ partials,
                                                                 T *__restrict__ acc, T ca, T cb, int apply,
                                                                 T *__restrict__ W, T eps) {
    const unsigned r = blockIdx.x;
    const int total = MC * nA;
    for (int sh = threadIdx.x; sh < nA; sh += kBlock) {
        const int e = (int)r * nA + sh;
        // (same order of additions; the loads of four partials are in flight together instead of one round trip each)
        double sn = 0.0, sp = 0.0;
        int p = 0;
        for (; p + 4 <= P; p += 4) {
            double vn[4], vp[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                vn[k] = partials[((size_t)(p + k) * total + e) * 2 + 0];
                vp[k] = partials[((size_t)(p + k) * total + e) * 2 + 1];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                sn += vn[k];
                sp += vp[k];
            }
        }
        for (; p < P; ++p) {
            sn += partials[((size_t)p * total + e) * 2 + 0];
            sp += partials[((size_t)p * total + e) * 2 + 1];
        }
        const size_t o = (size_t)r * nA + (nA - 1 - sh);
        const T gn = (T)sn, gp = (T)sp;
        const T tn = cb == T(1) ? gn : cb * gn, tp = cb == T(1) ? gp : cb * gp;
        acc[o] = ca == T(0) ? tn : (ca == T(1) ? acc[o] + tn : ca * acc[o] + tn);
        acc[total + o] = ca == T(0) ? tp : (ca == T(1) ? acc[total + o] + tp : ca * acc[total + o] + tp);
    }
    __syncthreads();
    if (apply) apply_normalize_row<T, true>(nA, r, W, acc, acc + (size_t)total, eps);
}

// ------------------------------------------------------------------------------------------------------------
// Persistent schedule kernel: a whole list of mini-batch operations (tnmf_hip_run_schedule) in ONE launch.
//
// With batch_size 3 a step of the stochastic schedules is eight kernels of a few microseconds each, 256 times per epoch:
// the epoch is launch latency.  Here the workgroups of one co-resident grid walk the operations themselves: every phase
// of an operation (reconstruct -> H update;  reconstruct -> W-gradient partials -> fixed-order sum + blend;  W update) is a
// loop over the blocks the kernels above would have been launched with, and a grid-wide barrier (one atomic counter,
// agent-scope release / acquire around it) stands where a kernel boundary was.  Same device functions, same arithmetic.
// Every workgroup executes the same operation list and therefore the same number of barriers: the grid drains.
// ------------------------------------------------------------------------------------------------------------
struct SchedArgs {
    Geo g;                  // the resident problem (g.N = all samples)
    Tile tR, tW, tH;        // tiles of reconstruct (sample plane), corr_W (shift plane), corr_H (sample plane)
    const void *V;
    void *W, *H, *R, *acc;
    double *partials;
    const tnmf_hip_op *ops;
    int n_ops, P;
    double reg, eps;
    unsigned *counter;      // zeroed before the launch: [0] arrivals, [32] generation
    int small_max;          // reconstruct of slices of up to this many samples takes the small-call form (0: never)
};

// counter[0]: arrivals (monotonic), counter[32]: generation flag on a line of its own.  The last workgroup to arrive
// publishes the generation; the others poll the flag (plain loads of a line nobody is doing atomics on).  Only the first
// wave of a workgroup talks to the memory system: __syncthreads() has drained every wave's stores (workgroup-scope
// release) before it issues the agent-scope release, and its agent-scope acquire invalidates the CU's vector cache for
// all four waves.
__device__ __forceinline__ void grid_barrier(unsigned *counter, unsigned &target) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (every wave: its own stores have left the CU before wave 0 releases)
    __syncthreads();
    target += gridDim.x;
    if (threadIdx.x < 64) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (threadIdx.x == 0) {
            const unsigned gen = target / gridDim.x;
            if (atomicAdd(counter, 1u) + 1u == target) {
                __hip_atomic_store(counter + 32, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                while (__hip_atomic_load(counter + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen)
                    __builtin_amdgcn_s_sleep(1);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// (An XCD-LOCAL flavour -- the workgroups of ONE XCD, elected through the hardware XCC id register, walk the list and
// meet at barriers without the agent-scope release, for the small batches of a large problem -- was built and measured in
// round 4 and not kept: 22.3 ms per ASG epoch on the reference's mini-batch geometry against 12.0 on the per-operation path;
// its barriers alone cost 5 us each.  profiles/r04_xcd_local_schedule_kernel.txt.)
__device__ __forceinline__ int cdiv_dev(int a, int b) { return (a + b - 1) / b; }

template <typename T>
__global__ __launch_bounds__(kBlock) void k_schedule(SchedArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const Geo &g = a.g;
    const int nA = g.Ay * g.Ax, MC = g.M * g.C;
    const size_t vs = (size_t)g.C * g.Dy * g.Dx, hs = (size_t)g.M * g.Hy * g.Hs;
    T *W = static_cast<T *>(a.W), *acc = static_cast<T *>(a.acc);
    unsigned target = 0;
    const unsigned wg = blockIdx.x, nwg = gridDim.x;
    auto barrier = [&]() { grid_barrier(a.counter, target); };
    for (int i = 0; i < a.n_ops; ++i) {
        const tnmf_hip_op op = a.ops[i];
        Geo gs = g;
        gs.N = op.n1 - op.n0;
        const T *Vb = static_cast<const T *>(a.V) + (size_t)op.n0 * vs;
        T *Hb = static_cast<T *>(a.H) + (size_t)op.n0 * hs;
        T *Rb = static_cast<T *>(a.R) + (size_t)op.n0 * vs;
        if (op.kind == TNMF_OP_UPDATE_H || op.kind == TNMF_OP_GRAD_W) {
            if (gs.N <= a.small_max) {   // (the small-call form, under launch_reconstruct's own condition)
                const int sty = cdiv_dev(g.Dy, kSmallTY), stx = cdiv_dev(g.Dx, kSmallTX);
                const unsigned nb = (unsigned)(gs.N * g.C * sty * stx);
                for (unsigned b = wg; b < nb; b += nwg) {
                    reconstruct_small_block<T>(gs, sty, stx, b, W, Hb, Rb, smem_raw);
                    __syncthreads();   // (the block function leaves its LDS tiles in use by wave 0's final sum)
                }
            } else {
                const unsigned nb = (unsigned)(gs.N * g.C * a.tR.tiles_y * a.tR.tiles_x);
                for (unsigned b = wg; b < nb; b += nwg) reconstruct_block<T>(gs, a.tR, b, W, Hb, Rb, smem_raw);
            }
            barrier();
        }
        if (op.kind == TNMF_OP_UPDATE_H) {
            const unsigned nb = (unsigned)(gs.N * g.M * a.tW.tiles_y * a.tW.tiles_x);
            for (unsigned b = wg; b < nb; b += nwg)
                corr_W_block<T, true>(gs, a.tW, b, Vb, Rb, W, Hb, (T *)nullptr, (T *)nullptr, (T)a.reg, (const T *)nullptr,
                                      smem_raw);
            barrier();
        } else if (op.kind == TNMF_OP_GRAD_W) {
            const int items = gs.N * a.tH.tiles_y * a.tH.tiles_x;
            const int P = items < a.P ? (items > 0 ? items : 1) : a.P;
            for (unsigned b = wg; b < (unsigned)(P * MC); b += nwg)
                corr_H_block<T>(gs, a.tH, P, (int)(b % P), (int)(b / P), Vb, Rb, Hb, a.partials, smem_raw);
            barrier();
            // fixed-order sum of the partials (double), flip to the reference's orientation, blend into the accumulator:
            // one workgroup per (m, c) row -- and when the W update follows immediately (ASG / ASAG: every batch), that
            // workgroup applies it to its row on the spot: one barrier less per batch step
            const int total = MC * nA;
            const bool apply_now = i + 1 < a.n_ops && a.ops[i + 1].kind == TNMF_OP_APPLY_W;
            for (unsigned r = wg; r < (unsigned)MC; r += nwg) {
                for (int sh = threadIdx.x; sh < nA; sh += kBlock) {
                    const int e = (int)r * nA + sh;
                    double sn = 0.0, sp = 0.0;
                    for (int p = 0; p < P; ++p) {
                        sn += a.partials[((size_t)p * total + e) * 2 + 0];
                        sp += a.partials[((size_t)p * total + e) * 2 + 1];
                    }
                    const size_t o = (size_t)r * nA + (nA - 1 - sh);
                    const T gn = (T)sn, gp = (T)sp, ca = (T)op.a, cb = (T)op.b;
                    const T tn = cb == T(1) ? gn : cb * gn, tp = cb == T(1) ? gp : cb * gp;
                    acc[o] = ca == T(0) ? tn : (ca == T(1) ? acc[o] + tn : ca * acc[o] + tn);
                    acc[total + o] = ca == T(0) ? tp : (ca == T(1) ? acc[total + o] + tp : ca * acc[total + o] + tp);
                }
                __syncthreads();   // the row of acc is complete (written by this workgroup's own threads)
                if (apply_now) apply_normalize_row<T, true>(nA, r, W, acc, acc + (size_t)MC * nA, (T)a.eps);
            }
            if (apply_now) ++i;    // (the W update has been done)
            barrier();
        } else if (op.kind == TNMF_OP_APPLY_W) {
            for (unsigned r = wg; r < (unsigned)MC; r += nwg) {
                __syncthreads();   // (the row reduction's shared words are reused row after row)
                apply_normalize_row<T, true>(nA, r, W, acc, acc + (size_t)MC * nA, (T)a.eps);
            }
            barrier();
        }
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_sqdiff_partial(const T *__restrict__ V, const T *__restrict__ R, size_t n,
                                                           double *__restrict__ partial) {
    __shared__ double sh[kBlock / 64];
    double acc = 0.0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const T d = V[i] - R[i];
        acc += (double)d * (double)d;
    }
    const double tot = block_sum<T>(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

__global__ __launch_bounds__(kBlock) void k_sum_partials(const double *__restrict__ partial, int n, double scale,
                                                         double *__restrict__ out) {
    __shared__ double sh[kBlock / 64];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += kBlock) acc += partial[i];
    const double tot = block_sum<double>(acc, sh);
    if (threadIdx.x == 0) *out = scale * tot;
}

// zero-padded 'same' 1-D convolution along one axis of arr[rows, len, inner] (scipy.ndimage.convolve1d, mode='constant'):
//   out[r, i, j] = sum_t k[t] * in[r, i + rad - t, j]
struct Taps {
    double k[kMaxTaps];
};

template <typename T>
__global__ void k_convolve_axis(const T *__restrict__ in, T *__restrict__ out, size_t rows, int len, int inner,
                                Taps taps, int ntaps) {
    const size_t total = rows * (size_t)len * inner;
    const int rad = (ntaps - 1) / 2;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int j = (int)(e % inner);
        const size_t rest = e / inner;
        const int i = (int)(rest % len);
        const size_t r = rest / len;
        const T *line = in + r * (size_t)len * inner + j;
        T acc = T(0);
        for (int t = 0; t < ntaps; ++t) {
            const int src = i + rad - t;
            if (src >= 0 && src < len) acc += (T)taps.k[t] * line[(size_t)src * inner];
        }
        out[e] = acc;
    }
}

inline int grid_for(size_t n, const tnmf_hip_ctx *ctx) {
    const size_t want = (n + kBlock - 1) / kBlock;
    const size_t cap = (size_t)ctx->num_cu * 8;
    return (int)(want < cap ? (want ? want : 1) : cap);
}

template <typename T>
int launch_reconstruct(const Geo &g, const void *W, const void *H, void *R, hipStream_t s) {
    const Tile t = make_tile(g.Dy, g.Dx);
    const size_t lds = ((size_t)(t.TY + g.Ay - 1) * (t.TX + g.Ax - 1) + (size_t)g.Ay * g.Ax) * sizeof(T);
    if (lds > 64 * 1024) return TNMF_E_UNSUPPORTED;
    const size_t blocks = (size_t)g.N * g.C * t.tiles_y * t.tiles_x;
    if (blocks > 0x7fffffffull) return TNMF_E_GEOM;
    size_t lds_small = 0;
    if (reconstruct_is_small<T>(g, t, &lds_small)) {
        // a small call (a mini-batch of a few samples): four waves per tile of 2 x 32 pixels, each on its own atoms
        // waves per workgroup: one atom per wave where that fits (up to 16 waves and 64 KB of LDS), else rounds of 8 or 4
        const int tiles_y = cdiv(g.Dy, kSmallTY), tiles_x = cdiv(g.Dx, kSmallTX);
        const dim3 grid((unsigned)((size_t)g.N * g.C * tiles_y * tiles_x));
        if (g.M > 8 && reconstruct_small_lds<T>(g, 16) <= 64 * 1024)
            hipLaunchKernelGGL((k_reconstruct_small<T, 16>), grid, dim3(64 * 16), reconstruct_small_lds<T>(g, 16), s, g, tiles_y,
                               tiles_x, (const T *)W, (const T *)H, (T *)R);
        else if (g.M > 4 && reconstruct_small_lds<T>(g, 8) <= 64 * 1024)
            hipLaunchKernelGGL((k_reconstruct_small<T, 8>), grid, dim3(64 * 8), reconstruct_small_lds<T>(g, 8), s, g, tiles_y,
                               tiles_x, (const T *)W, (const T *)H, (T *)R);
        else
            hipLaunchKernelGGL((k_reconstruct_small<T, 4>), grid, dim3(kBlock), lds_small, s, g, tiles_y, tiles_x,
                               (const T *)W, (const T *)H, (T *)R);
        TNMF_LAUNCH_CHECK();
        return TNMF_OK;
    }
    hipLaunchKernelGGL(k_reconstruct<T>, dim3((unsigned)blocks), dim3(kBlock), lds, s, g, t, (const T *)W,
                       (const T *)H, (T *)R);
    TNMF_LAUNCH_CHECK();
    return TNMF_OK;
}

template <typename T>
int launch_corr_W(const Geo &g, const void *V, const void *R, const void *W, void *Hio, void *neg, void *pos,
                  bool fused, double reg, hipStream_t s, const void *extra) {
    const Tile t = make_tile(g.Hy, g.Hx);
    const size_t lds = (2 * (size_t)(t.TY + g.Ay - 1) * (t.TX + g.Ax - 1) + (size_t)g.Ay * g.Ax) * sizeof(T);
    if (lds > 64 * 1024) return TNMF_E_UNSUPPORTED;
    const size_t blocks = (size_t)g.N * g.M * t.tiles_y * t.tiles_x;
    if (blocks > 0x7fffffffull) return TNMF_E_GEOM;
    if (fused)
        hipLaunchKernelGGL((k_corr_W<T, true>), dim3((unsigned)blocks), dim3(kBlock), lds, s, g, t, (const T *)V,
                           (const T *)R, (const T *)W, (T *)Hio, (T *)nullptr, (T *)nullptr, (T)reg, (const T *)extra);
    else
        hipLaunchKernelGGL((k_corr_W<T, false>), dim3((unsigned)blocks), dim3(kBlock), lds, s, g, t, (const T *)V,
                           (const T *)R, (const T *)W, (T *)nullptr, (T *)neg, (T *)pos, (T)0, (const T *)nullptr);
    TNMF_LAUNCH_CHECK();
    return TNMF_OK;
}

template <typename T>
int launch_corr_H(const Geo &g, const void *V, const void *R, const void *H, double *partials, int P,
                  hipStream_t s) {
    const Tile t = make_tile(g.Dy, g.Dx);
    const int nA = g.Ay * g.Ax;
    if (nA > kBlock * kMaxShiftsPerThread) return TNMF_E_UNSUPPORTED;
    const int gs = nA < kBlock ? nA : kBlock;
    const int G = kBlock / gs;
    size_t lds = ((size_t)(t.TY + g.Ay - 1) * (t.TX + g.Ax - 1) + 2 * (size_t)t.TY * t.TX) * sizeof(T);
    const size_t red = (size_t)G * nA * 2 * sizeof(double);
    if (red > lds) lds = red;
    if (lds > 64 * 1024) return TNMF_E_UNSUPPORTED;
    hipLaunchKernelGGL(k_corr_H<T>, dim3(P, g.M * g.C), dim3(kBlock), lds, s, g, t, P, (const T *)V, (const T *)R,
                       (const T *)H, partials);
    TNMF_LAUNCH_CHECK();
    return TNMF_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------
// host entry points used by api.hip
// ------------------------------------------------------------------------------------------------------------
int generic_reconstruct(tnmf_hip_ctx *, const Geo &g, int dtype, const void *W, const void *H, void *R,
                        hipStream_t s) {
    return dtype == 0 ? launch_reconstruct<float>(g, W, H, R, s) : launch_reconstruct<double>(g, W, H, R, s);
}

int generic_corr_W(tnmf_hip_ctx *, const Geo &g, int dtype, const void *V, const void *R, const void *W,
                   void *H_inout, void *neg, void *pos, bool fused, double reg, hipStream_t s, const void *extra) {
    return dtype == 0 ? launch_corr_W<float>(g, V, R, W, H_inout, neg, pos, fused, reg, s, extra)
                      : launch_corr_W<double>(g, V, R, W, H_inout, neg, pos, fused, reg, s, extra);
}

int generic_corr_H_chunks(const tnmf_hip_ctx *ctx, const Geo &g) {
    const Tile t = make_tile(g.Dy, g.Dx);
    const long items = (long)g.N * t.tiles_y * t.tiles_x;
    long P = ((long)ctx->num_cu * 8 + (long)g.M * g.C - 1) / ((long)g.M * g.C);  // ~8 blocks per CU in total
    if (P < 1) P = 1;
    if (P > items) P = items;
    if (P > 4096) P = 4096;
    return (int)P;
}

int generic_corr_H(tnmf_hip_ctx *, const Geo &g, int dtype, const void *V, const void *R, const void *H,
                   double *partials, int P, hipStream_t s) {
    return dtype == 0 ? launch_corr_H<float>(g, V, R, H, partials, P, s)
                      : launch_corr_H<double>(g, V, R, H, partials, P, s);
}

int finalize_corr_H(const Geo &g, int dtype, const double *partials, int P, void *neg, void *pos, hipStream_t s) {
    const int MC = g.M * g.C, nA = g.Ay * g.Ax;
    const int blocks = cdiv(MC * nA, kFinOut);
    if (dtype == 0)
        hipLaunchKernelGGL(k_corr_H_finalize<float>, dim3(blocks), dim3(kFinOut * kFinSlices), 0, s, MC, nA, P,
                           partials, (float *)neg, (float *)pos);
    else
        hipLaunchKernelGGL(k_corr_H_finalize<double>, dim3(blocks), dim3(kFinOut * kFinSlices), 0, s, MC, nA, P,
                           partials, (double *)neg, (double *)pos);
    TNMF_LAUNCH_CHECK();
    return TNMF_OK;
}

int launch_mu_update(const tnmf_hip_ctx *ctx, int dtype, void *arr, const void *neg, void *pos, double reg, size_t n,
                     hipStream_t s) {
    if (n == 0) return TNMF_OK;
    const int grid = grid_for(n, ctx);
    if (dtype == 0)
        hipLaunchKernelGGL(k_mu_update<float>, dim3(grid), dim3(kBlock), 0, s, (float *)arr, (const float *)neg,
                           (float *)pos, (float)reg, n);
    else
        hipLaunchKernelGGL(k_mu_update<double>, dim3(grid), dim3(kBlock), 0, s, (double *)arr, (const double *)neg,
                           (double *)pos, reg, n);
    TNMF_LAUNCH_CHECK();
    return TNMF_OK;
}

int launch_axpby(const tnmf_hip_ctx *ctx, int dtype, void *acc, const void *g, double a, double b, size_t n,
                 hipStream_t s) {
    if (n == 0) return TNMF_OK;
    const int grid = grid_for(n, ctx);
    if (dtype == 0)
        hipLaunchKernelGGL(k_axpby<float>, dim3(grid), dim3(kBlock), 0, s, (float *)acc, (const float *)g, (float)a, (float)b, n);
    else
        hipLaunchKernelGGL(k_axpby<double>, dim3(grid), dim3(kBlock), 0, s, (double *)acc, (const double *)g, a, b, n);
    TNMF_LAUNCH_CHECK();
    return TNMF_OK;
}

int launch_sum_parts(const tnmf_hip_ctx *ctx, int dtype, const void *parts, int n_parts, size_t n, void *out,
                     hipStream_t s) {
    if (n == 0) return TNMF_OK;
    const int grid = grid_for(n, ctx);
    if (dtype == 0)
        hipLaunchKernelGGL(k_sum_parts<float>, dim3(grid), dim3(kBlock), 0, s, (const float *)parts, n_parts, n, (float *)out);
    else
        hipLaunchKernelGGL(k_sum_parts<double>, dim3(grid), dim3(kBlock), 0, s, (const double *)parts, n_parts, n,
                           (double *)out);
    TNMF_LAUNCH_CHECK();
    return TNMF_OK;
}

int launch_apply_normalize_W(const Geo &g, int dtype, void *W, const void *neg, void *pos, double eps, bool apply,
                             hipStream_t s) {
    const int nA = g.Ay * g.Ax, rows = g.M * g.C;
    if (dtype == 0) {
        if (apply)
            hipLaunchKernelGGL((k_apply_normalize_W<float, true>), dim3(rows), dim3(kBlock), 0, s, nA, (float *)W,
                               (const float *)neg, (float *)pos, (float)eps);
        else
            hipLaunchKernelGGL((k_apply_normalize_W<float, false>), dim3(rows), dim3(kBlock), 0, s, nA, (float *)W,
                               (const float *)nullptr, (float *)nullptr, 0.f);
    } else {
        if (apply)
            hipLaunchKernelGGL((k_apply_normalize_W<double, true>), dim3(rows), dim3(kBlock), 0, s, nA, (double *)W,
                               (const double *)neg, (double *)pos, eps);
        else
            hipLaunchKernelGGL((k_apply_normalize_W<double, false>), dim3(rows), dim3(kBlock), 0, s, nA, (double *)W,
                               (const double *)nullptr, (double *)nullptr, 0.0);
    }
    TNMF_LAUNCH_CHECK();
    return TNMF_OK;
}

int launch_half_sqdiff(const tnmf_hip_ctx *ctx, int dtype, const void *V, const void *R, size_t n, double *partials,
                       double *out_dev, hipStream_t s) {
    int grid = grid_for(n, ctx);
    if (grid > kEnergyPartials) grid = kEnergyPartials;
    if (dtype == 0)
        hipLaunchKernelGGL(k_sqdiff_partial<float>, dim3(grid), dim3(kBlock), 0, s, (const float *)V,
                           (const float *)R, n, partials);
    else
        hipLaunchKernelGGL(k_sqdiff_partial<double>, dim3(grid), dim3(kBlock), 0, s, (const double *)V,
                           (const double *)R, n, partials);
    TNMF_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(kBlock), 0, s, partials, grid, 0.5, out_dev);
    TNMF_LAUNCH_CHECK();
    return TNMF_OK;
}

int launch_convolve_axis(const tnmf_hip_ctx *ctx, int dtype, const void *in, void *out, size_t rows, int len,
                         int inner, const double *kernel_host, int ntaps, hipStream_t s) {
    if (ntaps < 1 || ntaps > kMaxTaps || (ntaps & 1) == 0) return TNMF_E_UNSUPPORTED;
    Taps taps;
    for (int i = 0; i < kMaxTaps; ++i) taps.k[i] = i < ntaps ? kernel_host[i] : 0.0;
    const size_t total = rows * (size_t)len * inner;
    if (total == 0) return TNMF_OK;
    const int grid = grid_for(total, ctx);
    if (dtype == 0)
        hipLaunchKernelGGL(k_convolve_axis<float>, dim3(grid), dim3(kBlock), 0, s, (const float *)in, (float *)out,
                           rows, len, inner, taps, ntaps);
    else
        hipLaunchKernelGGL(k_convolve_axis<double>, dim3(grid), dim3(kBlock), 0, s, (const double *)in,
                           (double *)out, rows, len, inner, taps, ntaps);
    TNMF_LAUNCH_CHECK();
    return TNMF_OK;
}

// ------------------------------------------------------------------------------------------------------------
// reconstruction modes: pad the activations / fold the gradient (reference: backends/_PyTorchBackend.py:42-52)
// ------------------------------------------------------------------------------------------------------------
namespace {

// activation index copied to padded position j of one axis (-1: a zero); S = activation length, a = atom length
__device__ __forceinline__ int pad_src(int j, int S, int a, int mode) {
    const int l = a - 1;
    if (mode == TNMF_MODE_FULL) {
        const int u = j - l;
        return (u >= 0 && u < S) ? u : -1;
    }
    if (j >= l) return j - l;
    return mode == TNMF_MODE_CIRCULAR ? S - l + j : l - j;   // wrap / mirror (without repeating the edge)
}

// second padded position (besides u + a - 1) that copies activation u, or -1
__device__ __forceinline__ int pad_dup(int u, int S, int a, int mode) {
    const int l = a - 1;
    if (mode == TNMF_MODE_CIRCULAR) return u >= S - l ? u - (S - l) : -1;
    if (mode == TNMF_MODE_REFLECT) return (u >= 1 && u <= l) ? l - u : -1;
    return -1;
}

template <typename T>
__global__ void k_pad_H(size_t rows, int Sy, int Sx, int Ay, int Ax, int Py, int Px, int mode,
                        const T *__restrict__ H, T *__restrict__ Hp) {
    const size_t total = rows * (size_t)Py * Px;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int jx = (int)(e % Px);
        const size_t rest = e / Px;
        const int jy = (int)(rest % Py);
        const size_t r = rest / Py;
        const int uy = Py == 1 ? 0 : pad_src(jy, Sy, Ay, mode);
        const int ux = pad_src(jx, Sx, Ax, mode);
        Hp[e] = (uy >= 0 && ux >= 0) ? H[(r * Sy + uy) * Sx + ux] : T(0);
    }
}

template <typename T>
__global__ void k_fold_H(size_t rows, int Sy, int Sx, int Ay, int Ax, int Py, int Px, int mode,
                         const T *__restrict__ Gp, T *__restrict__ G) {
    const size_t total = rows * (size_t)Sy * Sx;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int ux = (int)(e % Sx);
        const size_t rest = e / Sx;
        const int uy = (int)(rest % Sy);
        const size_t r = rest / Sy;
        const int jy[2] = {Py == 1 ? 0 : uy + Ay - 1, Py == 1 ? -1 : pad_dup(uy, Sy, Ay, mode)};
        const int jx[2] = {ux + Ax - 1, pad_dup(ux, Sx, Ax, mode)};
        T acc = T(0);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                if (jy[a] >= 0 && jx[b] >= 0) acc += Gp[(r * Py + jy[a]) * Px + jx[b]];
        G[e] = acc;
    }
}

}  // namespace

// shift length of one axis for a mode (reference: backends/_Backend.py:60-73)
static inline int mode_shift(int d, int a, int mode) {
    return mode == TNMF_MODE_VALID ? d + a - 1 : (mode == TNMF_MODE_FULL ? d - a + 1 : d);
}

int launch_pad_fold(const tnmf_hip_ctx *ctx, const Geo &g, int dtype, int mode, bool fold, const void *in, void *out,
                    hipStream_t s) {
    const int Sy = g.Dy == 1 && g.Ay == 1 ? 1 : mode_shift(g.Dy, g.Ay, mode), Sx = mode_shift(g.Dx, g.Ax, mode);
    if (Sy < 1 || Sx < 1 || g.Ay - 1 > Sy || g.Ax - 1 > Sx) return TNMF_E_GEOM;
    // 'reflect' mirrors without repeating the edge: a pad of A-1 needs A-1 < S (torch's reflect pad raises otherwise,
    // _PyTorchBackend.py:42-52 / torch.nn.functional.pad); pad == S would read one element past the activation row
    if (mode == TNMF_MODE_REFLECT && (g.Ay - 1 >= Sy || g.Ax - 1 >= Sx)) return TNMF_E_GEOM;
    const size_t rows = (size_t)g.N * g.M;
    const size_t total = rows * (fold ? (size_t)Sy * Sx : (size_t)g.Hy * g.Hx);
    if (total == 0) return TNMF_OK;
    const int grid = grid_for(total, ctx);
#define LAUNCH_PF(K_, T_)                                                                                     \
    hipLaunchKernelGGL(K_<T_>, dim3(grid), dim3(kBlock), 0, s, rows, Sy, Sx, g.Ay, g.Ax, g.Hy, g.Hx, mode,    \
                       (const T_ *)in, (T_ *)out)
    if (fold) {
        if (dtype == 0) LAUNCH_PF(k_fold_H, float); else LAUNCH_PF(k_fold_H, double);
    } else {
        if (dtype == 0) LAUNCH_PF(k_pad_H, float); else LAUNCH_PF(k_pad_H, double);
    }
#undef LAUNCH_PF
    TNMF_LAUNCH_CHECK();
    return TNMF_OK;
}


// ------------------------------------------------------------------------------------------------------------
// the persistent schedule kernel (see k_schedule)
// ------------------------------------------------------------------------------------------------------------
namespace {

template <typename T>
size_t schedule_lds(const Geo &g, const Tile &tR, const Tile &tW, const Tile &tH) {
    const size_t nA = (size_t)g.Ay * g.Ax;
    const size_t r = ((size_t)(tR.TY + g.Ay - 1) * (tR.TX + g.Ax - 1) + nA) * sizeof(T);
    const size_t w = (2 * (size_t)(tW.TY + g.Ay - 1) * (tW.TX + g.Ax - 1) + nA) * sizeof(T);
    size_t h = ((size_t)(tH.TY + g.Ay - 1) * (tH.TX + g.Ax - 1) + 2 * (size_t)tH.TY * tH.TX) * sizeof(T);
    const int gs = nA < (size_t)kBlock ? (int)nA : kBlock;
    const size_t red = (size_t)(kBlock / gs) * nA * 2 * sizeof(double);
    if (red > h) h = red;
    return r > w ? (r > h ? r : h) : (w > h ? w : h);
}

}  // namespace

// largest slice (samples) whose reconstruct takes the small-call form, 0 for none; its LDS need
template <typename T>
static int schedule_small_max(const Geo &g, const Tile &tR, size_t *lds_small) {
    Geo one = g;
    one.N = 1;
    if (!reconstruct_is_small<T>(one, tR, lds_small)) return 0;
    const int per = g.C * tR.tiles_y * tR.tiles_x;   // blocks of the plain form per sample: small while N * per < 64
    return per > 0 ? (63 / per) : 0;
}

static size_t schedule_lds_any(const Geo &g, int dtype, const Tile &tR, const Tile &tW, const Tile &tH, int *small_max) {
    size_t lds = dtype == 0 ? schedule_lds<float>(g, tR, tW, tH) : schedule_lds<double>(g, tR, tW, tH);
    size_t ls = 0;
    *small_max = dtype == 0 ? schedule_small_max<float>(g, tR, &ls) : schedule_small_max<double>(g, tR, &ls);
    if (*small_max > 0 && ls > lds) lds = ls;
    return lds;
}

bool generic_schedule_fits(const tnmf_hip_ctx *, const Geo &g, int dtype) {
    const Tile tR = make_tile(g.Dy, g.Dx), tW = make_tile(g.Hy, g.Hx), tH = make_tile(g.Dy, g.Dx);
    if (g.Ay * g.Ax > kBlock * kMaxShiftsPerThread) return false;
    int small_max = 0;
    return schedule_lds_any(g, dtype, tR, tW, tH, &small_max) <= 64 * 1024;
}

// Workgroups of k_schedule that are resident AT ONCE on this device, capped at min(CUs, 128): the kernel's grid.  Its
// barrier spins until every workgroup of the grid has arrived, so a workgroup that is not resident (waiting for a slot a
// spinning workgroup holds) would hang the launch: residency is a checked precondition (occupancy query with the
// kernel's real register / LDS footprint), not an assumption about "one workgroup of <= 64 KB per CU".  0: not even one.
static int schedule_grid(const tnmf_hip_ctx *ctx, int dtype, size_t lds) {
    int per_cu = 0;
    const hipError_t e = dtype == 0 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_schedule<float>, kBlock, lds)
                                    : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_schedule<double>, kBlock, lds);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    const long resident = (long)per_cu * ctx->num_cu;
    const int want = ctx->num_cu < 128 ? ctx->num_cu : 128;
    return (int)(resident < want ? resident : want);
}

int generic_schedule_chunks(const tnmf_hip_ctx *ctx, const Geo &g) {
    // split of the pixel sum of the W gradient: enough (chunk, atom, channel) blocks for the grid, no more
    const int grid = ctx->num_cu < 128 ? ctx->num_cu : 128;
    int P = cdiv(grid, g.M * g.C);
    return P < 1 ? 1 : (P > 64 ? 64 : P);
}

int generic_run_schedule(tnmf_hip_ctx *ctx, const Geo &g, int dtype, const void *V, void *W, void *H, void *R, void *acc,
                         double *partials, int P, const tnmf_hip_op *ops_dev, int n_ops, double reg, double eps,
                         unsigned *counter, hipStream_t s) {
    SchedArgs a;
    a.g = g;
    a.tR = make_tile(g.Dy, g.Dx);
    a.tW = make_tile(g.Hy, g.Hx);
    a.tH = make_tile(g.Dy, g.Dx);
    a.V = V;
    a.W = W;
    a.H = H;
    a.R = R;
    a.acc = acc;
    a.partials = partials;
    a.ops = ops_dev;
    a.n_ops = n_ops;
    a.P = P;
    a.reg = reg;
    a.eps = eps;
    a.counter = counter;
    const size_t lds = schedule_lds_any(g, dtype, a.tR, a.tW, a.tH, &a.small_max);
    if (lds > 64 * 1024 || ctx->persistent == 0) return TNMF_E_UNSUPPORTED;
    // the whole grid must be resident at once (the barrier): as many workgroups as the occupancy query says fit, at most
    // one per CU and 128 (the kernel's loops stride by their number: any grid size computes the same thing)
    const int grid = schedule_grid(ctx, dtype, lds);
    if (grid < 1) return TNMF_E_UNSUPPORTED;   // (nothing launched, nothing written: the caller walks the list per operation)
    TNMF_HIP_TRY(hipMemsetAsync(counter, 0, 64 * sizeof(unsigned), s));
    const void *fn = dtype == 0 ? (const void *)k_schedule<float> : (const void *)k_schedule<double>;
    void *params[] = {&a};
    if (ctx->persistent == 2) {
        // cooperative launch: the runtime refuses a grid it cannot make co-resident instead of starting it
        const hipError_t e = hipLaunchCooperativeKernel(fn, dim3(grid), dim3(kBlock), params, (unsigned)lds, s);
        if (e == hipErrorCooperativeLaunchTooLarge || e == hipErrorNotSupported) {
            (void)hipGetLastError();
            return TNMF_E_UNSUPPORTED;
        }
        if (e != hipSuccess) return (int)e;
        return TNMF_OK;
    }
    TNMF_HIP_TRY(hipLaunchKernel(fn, dim3(grid), dim3(kBlock), params, lds, s));
    return TNMF_OK;
}

int launch_finalize_blend_apply(const Geo &g, int dtype, const double *partials, int P, void *acc, double a, double b,
                                bool apply, void *W, double eps, hipStream_t s) {
    const int MC = g.M * g.C, nA = g.Ay * g.Ax;
    if (dtype == 0)
        hipLaunchKernelGGL(k_finalize_blend_apply<float>, dim3(MC), dim3(kBlock), 0, s, MC, nA, P, partials, (float *)acc,
                           (float)a, (float)b, apply ? 1 : 0, (float *)W, (float)eps);
    else
        hipLaunchKernelGGL(k_finalize_blend_apply<double>, dim3(MC), dim3(kBlock), 0, s, MC, nA, P, partials, (double *)acc,
                           a, b, apply ? 1 : 0, (double *)W, eps);
    TNMF_LAUNCH_CHECK();
    return TNMF_OK;
}
