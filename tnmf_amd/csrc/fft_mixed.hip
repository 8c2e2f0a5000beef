// fft_mixed.hip -- "mixed" contractions of the FFT family: transform along x only, the few taps along y directly.
//
// After the row transforms an atom is only Ay rows tall, so the contraction over (atom, atom row) costs Ay complex
// multiply-adds per row-spectrum entry -- fewer issue slots than the column transform it replaces (two LDS round trips
// per radix stage) as long as C*Ay stays small -- and needs no LDS and no barrier at all:
//
//   reconstruct   OT[n,c,y,kx]  = sum_m sum_a T[n,m,y+a,kx] * WT[m,c,Ay-1-a,kx]          (then C2R along x, crop)
//   W gradient    GT[m,c,a,kx]  = sum_n sum_y T[n,m,y+a,kx] * conj(VT[n,c,y,kx])        (then C2R along x, crop, flip)
//
// T, VT, WT = row spectra (fft_kernels.h).  Same mathematics as NumPy.py:122-132 / 69-91 of the reference with the
// x axis in the frequency domain.  Lanes run along kx (coalesced 128-byte segments), everything else is registers.
#include "fft.h"

namespace {

// float only: a complex value is one 64-bit register pair and every complex multiply-add is two packed FMAs whose
// operand broadcasts, swaps and sign flips ride on the op_sel / neg modifiers (left to itself hipcc materialises the
// broadcast and swapped operand pairs in extra registers: 300-400 VGPRs for these kernels instead of ~130).
//   D.lo = A[op_sel[0]] * B[op_sel[1]] + C[op_sel[2]],  D.hi likewise with op_sel_hi
typedef float f2v __attribute__((ext_vector_type(2)));
template <typename T>
struct cplx_of;
template <>
struct cplx_of<float> {
    using type = f2v;
};
template <typename T>
using cplx = typename cplx_of<T>::type;

// acc += a * b
__device__ __forceinline__ void cfma(f2v &acc, f2v a, f2v b) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(acc) : "v"(a), "v"(b));
}
// acc += a * conj(b)
__device__ __forceinline__ void cfmac(f2v &acc, f2v a, f2v b) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "+v"(acc) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(a), "v"(b));
}

#define CHECK(rc_expr)                  \
    do {                                \
        const int _rc = (rc_expr);      \
        if (_rc != TNMF_OK) return _rc; \
    } while (0)

constexpr int kMixCols = 16;   // kx per block row group: 16 lanes x 8 bytes = one 128-byte segment

// ---- reconstruct ---------------------------------------------------------------------------------------------------
// block = 16 kx x STRIPS strips of S output rows; a thread keeps S outputs per channel in registers and, per atom, the
// S+AY-1 row-spectrum entries they depend on.  grid (kx tiles, row blocks, samples)
template <typename T, int AY, int CG, int S, int STRIPS>
__global__ __launch_bounds__(kMixCols *STRIPS, 2) void k_mix_reconstruct(const cplx<T> *Tsp, const cplx<T> *WT,
                                                                      cplx<T> *OT, int M, int C, int Hy, int Dy, int KX,
                                                                      int KXP) {
    const int col = threadIdx.x & (kMixCols - 1), strip = threadIdx.x / kMixCols;
    const int kx = blockIdx.x * kMixCols + col, kxc = min(kx, KX - 1);
    const int y0 = (blockIdx.y * STRIPS + strip) * S, n = blockIdx.z;
    cplx<T> acc[CG][S];
#pragma unroll
    for (int c = 0; c < CG; ++c)
#pragma unroll
        for (int y = 0; y < S; ++y) acc[c][y] = {0, 0};
    // Addresses: a buffer descriptor per plane (wave-uniform, scalar registers), a scalar row offset, and ONE per-lane
    // byte offset (first row of the strip, kx) that never changes -- no vector address arithmetic and no offset
    // registers.  Rows past the plane (y0 + r >= Hy) are read only under outputs that are never stored (output row y needs
    // rows up to y + AY - 1 <= Hy - 1 whenever y < Dy).  The descriptor's range check does NOT catch them -- on gfx9 raw
    // buffers it covers the vector offset only, not the scalar one -- so they come from the next plane or, behind the last
    // plane, from the slack rows the workspace layout keeps after T (fft.hip: make_layout); whatever they hold only
    // reaches accumulators that are dropped.  The rows of the next atom are fetched while the current ones are multiplied.
    static_assert(sizeof(cplx<T>) == 8, "float spectra");
    typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
    auto ld = [](const __amdgpu_buffer_rsrc_t &rs, int lane_off, int row_off) {
        const u32x2v w2 = __builtin_amdgcn_raw_buffer_load_b64(rs, lane_off, row_off, 0);
        const unsigned re = w2[0], im = w2[1];   // (bit_cast straight on a vector element reads element 0: hipcc 7.2)
        return cplx<T>{__builtin_bit_cast(float, re), __builtin_bit_cast(float, im)};
    };
    const int rowb = KXP * 8;
    const long tplane = (long)Hy * KXP, wplane = (long)AY * KXP;
    const int lane_t = (y0 * KXP + kxc) * 8, lane_w = kxc * 8;
    const __amdgpu_buffer_rsrc_t wrs =
        __builtin_amdgcn_make_buffer_rsrc((void *)WT, 0, (int)((long)M * C * wplane * 8), 0x00020000);
    auto plane = [&](int m) {
        return __builtin_amdgcn_make_buffer_rsrc((void *)(Tsp + ((long)n * M + m) * tplane), 0, (int)(tplane * 8),
                                                 0x00020000);
    };
    cplx<T> t[S + AY - 1], tn[S + AY - 1];
    {
        const __amdgpu_buffer_rsrc_t rs = plane(0);
#pragma unroll
        for (int r = 0; r < S + AY - 1; ++r) t[r] = ld(rs, lane_t, r * rowb);
    }
    if constexpr (CG == 1) {
        // One channel: the AY rows of W spectra of an atom are the same for all STRIPS strips of the workgroup.  Loaded
        // per thread they were 12 of the 39 memory instructions of an atom step and -- although they hit in L2 -- a
        // fifth of the kernel (measured with constants in their place: 0.69 -> 0.575 ms).  They are staged through LDS in
        // chunks of MC atoms: 16-byte loads one chunk ahead (in flight under the atoms of the current chunk), one
        // barrier per chunk, and an atom step reads its rows with AY ds_read_b64 (all strips of a wave read the same
        // addresses: a broadcast).
        constexpr int MC = 8, NT = kMixCols * STRIPS, ROWS = MC * AY, PIECES = ROWS * (kMixCols / 2);
        constexpr int NPP = (PIECES + NT - 1) / NT;
        typedef float f4v __attribute__((ext_vector_type(4)));
        __shared__ __align__(16) cplx<T> wl[2][ROWS * kMixCols];
        const cplx<T> *wsrc = WT + blockIdx.x * kMixCols;   // (KXP is a multiple of 16 entries: whole 128-byte segments)
        const int wrows = M * AY;
        f4v pre[NPP];
        auto wfetch = [&](int chunk) {
#pragma unroll
            for (int k = 0; k < NPP; ++k) {
                const int p = threadIdx.x + k * NT, row = min(chunk * ROWS + (p >> 3), wrows - 1);
                pre[k] = *reinterpret_cast<const f4v *>(wsrc + (long)row * KXP + 2 * (p & 7));
            }
        };
        auto wpark = [&](int buf) {
#pragma unroll
            for (int k = 0; k < NPP; ++k) {
                const int p = threadIdx.x + k * NT;
                if (p < PIECES) *reinterpret_cast<f4v *>(&wl[buf][(p >> 3) * kMixCols + 2 * (p & 7)]) = pre[k];
            }
        };
        wfetch(0);
        wpark(0);
        __syncthreads();
#pragma unroll 1
        for (int m0 = 0, ch = 0; m0 < M; m0 += MC, ++ch) {
            const int buf = ch & 1;
            const bool more = m0 + MC < M;
            if (more) wfetch(ch + 1);
#pragma unroll 1
            for (int mi = 0; mi < MC; ++mi) {
                const int m = m0 + mi;
                if (m >= M) break;
                cplx<T> w[AY];
#pragma unroll
                for (int a = 0; a < AY; ++a) w[a] = wl[buf][(mi * AY + (AY - 1 - a)) * kMixCols + col];
                {
                    const __amdgpu_buffer_rsrc_t rs = plane(min(m + 1, M - 1));
#pragma unroll
                    for (int r = 0; r < S + AY - 1; ++r) tn[r] = ld(rs, lane_t, r * rowb);
                }
#pragma unroll
                for (int a = 0; a < AY; ++a)
#pragma unroll
                    for (int y = 0; y < S; ++y) cfma(acc[0][y], t[y + a], w[a]);
#pragma unroll
                for (int r = 0; r < S + AY - 1; ++r) t[r] = tn[r];
            }
            if (more) wpark(buf ^ 1);
            __syncthreads();
        }
    } else {
#pragma unroll 1
    for (int m = 0; m < M; ++m) {
        cplx<T> w[CG][AY];
#pragma unroll
        for (int c = 0; c < CG; ++c) {
            const int wbase = ((m * C + min(c, C - 1)) * AY) * rowb;
#pragma unroll
            for (int a = 0; a < AY; ++a) w[c][a] = ld(wrs, lane_w, wbase + (AY - 1 - a) * rowb);
        }
        {
            const __amdgpu_buffer_rsrc_t rs = plane(min(m + 1, M - 1));
#pragma unroll
            for (int r = 0; r < S + AY - 1; ++r) tn[r] = ld(rs, lane_t, r * rowb);
        }
#pragma unroll
        for (int c = 0; c < CG; ++c)
#pragma unroll
            for (int a = 0; a < AY; ++a)
#pragma unroll
                for (int y = 0; y < S; ++y) cfma(acc[c][y], t[y + a], w[c][a]);
#pragma unroll
        for (int r = 0; r < S + AY - 1; ++r) t[r] = tn[r];
    }
    }
    if (kx >= KX) return;
#pragma unroll
    for (int c = 0; c < CG; ++c) {
        if (c >= C) break;
#pragma unroll
        for (int y = 0; y < S; ++y)
            if (y0 + y < Dy) OT[((long)n * C + c) * ((long)Dy * KXP) + (long)(y0 + y) * KXP + kx] = acc[c][y];
    }
}

// 1-D signals (one row per plane): OT[n,c,kx] = sum_m T[n,m,kx] * WT[m,c,kx] -- a thread per (sample, kx), all channels
// of the sample in registers, the atom loop unrolled by four so that four loads of T are in flight.
template <typename T, int CG, int SAMPLES>
__global__ __launch_bounds__(kMixCols *SAMPLES) void k_mix_reconstruct_1d(const cplx<T> *Tsp, const cplx<T> *WT,
                                                                        cplx<T> *OT, int N, int M, int C, int KX,
                                                                        int KXP) {
    const int col = threadIdx.x & (kMixCols - 1), sub = threadIdx.x / kMixCols;
    const int kx = blockIdx.x * kMixCols + col, kxc = min(kx, KX - 1);
    const int n = blockIdx.y * SAMPLES + sub, nc = min(n, N - 1);
    cplx<T> acc[CG];
#pragma unroll
    for (int c = 0; c < CG; ++c) acc[c] = {0, 0};
    const cplx<T> *tp = Tsp + (long)nc * M * KXP + kxc;
    int m = 0;
    for (; m + 4 <= M; m += 4) {
        cplx<T> t[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) t[q] = tp[(long)(m + q) * KXP];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < CG; ++c) cfma(acc[c], t[q], WT[((long)(m + q) * C + min(c, C - 1)) * KXP + kxc]);
    }
    for (; m < M; ++m) {
        const cplx<T> t = tp[(long)m * KXP];
#pragma unroll
        for (int c = 0; c < CG; ++c) cfma(acc[c], t, WT[((long)m * C + min(c, C - 1)) * KXP + kxc]);
    }
    if (kx >= KX || n >= N) return;
#pragma unroll
    for (int c = 0; c < CG; ++c) {
        if (c >= C) break;
        OT[((long)n * C + c) * KXP + kx] = acc[c];
    }
}

template <typename T, int AY>
int launch_mix_reconstruct(const void *Tsp, const void *WT, void *OT, const Geo &g, int KX, int KXP, hipStream_t s) {
    const unsigned tiles = (unsigned)cdiv(KX, kMixCols);
    if constexpr (AY == 1) {
        if (g.Dy == 1) {   // 1-D signals
            constexpr int SAMPLES = 16;
            hipLaunchKernelGGL((k_mix_reconstruct_1d<T, 3, SAMPLES>), dim3(tiles, (unsigned)cdiv(g.N, SAMPLES)),
                               dim3(kMixCols * SAMPLES), 0, s, (const cplx<T> *)Tsp, (const cplx<T> *)WT, (cplx<T> *)OT,
                               g.N, g.M, g.C, KX, KXP);
            TNMF_LAUNCH_CHECK();
            return TNMF_OK;
        }
    }
    if (g.C == 1) {
        constexpr int S = 16, STRIPS = 8;
        hipLaunchKernelGGL((k_mix_reconstruct<T, AY, 1, S, STRIPS>),
                           dim3(tiles, (unsigned)cdiv(g.Dy, S * STRIPS), (unsigned)g.N), dim3(kMixCols * STRIPS), 0, s,
                           (const cplx<T> *)Tsp, (const cplx<T> *)WT, (cplx<T> *)OT, g.M, g.C, g.Hy, g.Dy, KX, KXP);
    } else {
        // up to three channels: shorter strips keep the three accumulator sets and the atom rows of three channels in
        // registers
        constexpr int S = 8, STRIPS = 8;
        hipLaunchKernelGGL((k_mix_reconstruct<T, AY, 3, S, STRIPS>),
                           dim3(tiles, (unsigned)cdiv(g.Dy, S * STRIPS), (unsigned)g.N), dim3(kMixCols * STRIPS), 0, s,
                           (const cplx<T> *)Tsp, (const cplx<T> *)WT, (cplx<T> *)OT, g.M, g.C, g.Hy, g.Dy, KX, KXP);
    }
    TNMF_LAUNCH_CHECK();
    return TNMF_OK;
}

// ---- W gradient ----------------------------------------------------------------------------------------------------
// block = 16 kx x GROUPS sample groups for one atom; a thread walks the rows of its samples with a rolling window of AY
// row-spectrum entries and accumulates the AY lags for neg (against VT) and pos (against RT).
// grid (atoms, kx tiles, sample-group blocks); partial sums [group][M*C][AY][KXP], summed in order afterwards
template <typename T, int AY, int CG, int GROUPS>
__global__ __launch_bounds__(kMixCols *GROUPS, 2) void k_mix_grad_W(const cplx<T> *Tsp, const cplx<T> *VT,
                                                                 const cplx<T> *RT, cplx<T> *Gn, cplx<T> *Gp, int N,
                                                                 int M, int C, int Hy, int Dy, int KX, int KXP,
                                                                 int nper) {
    const int col = threadIdx.x & (kMixCols - 1), sub = threadIdx.x / kMixCols;
    // the atom index runs fastest over the blocks: the blocks in flight share the V^/R^ rows of one kx tile (L2)
    const int kx = blockIdx.y * kMixCols + col, kxc = min(kx, KX - 1);
    const int m = blockIdx.x, grp = blockIdx.z * GROUPS + sub;
    cplx<T> an[CG][AY], ap[CG][AY];
#pragma unroll
    for (int c = 0; c < CG; ++c)
#pragma unroll
        for (int a = 0; a < AY; ++a) {
            an[c][a] = {0, 0};
            ap[c][a] = {0, 0};
        }
    const int nbeg = grp * nper, nend = min(N, nbeg + nper);
    const long tplane = (long)Hy * KXP, vplane = (long)Dy * KXP;
    static_assert(CG == 1, "one channel per thread");
#pragma unroll 1
    for (int n = nbeg; n < nend; ++n) {
        const cplx<T> *tp = Tsp + ((long)n * M + m) * tplane;   // uniform bases, 32-bit per-thread offsets
        const cplx<T> *vp = VT + (long)n * C * vplane, *rp = RT + (long)n * C * vplane;
        // Rows are taken in blocks of AY: the block [yb, yb+AY) needs T[yb .. yb+2AY-2] and V/R[yb .. yb+AY-1].  The
        // next block's AY new T rows and its V/R rows are fetched while the current block is accumulated.
        cplx<T> t[2 * AY - 1], v[AY], r[AY], tn[AY], vn[AY], rn[AY];
#pragma unroll
        for (int i = 0; i < 2 * AY - 1; ++i) t[i] = tp[(unsigned)(min(i, Hy - 1) * KXP + kxc)];
#pragma unroll
        for (int j = 0; j < AY; ++j) {
            const unsigned o = (unsigned)(min(j, Dy - 1) * KXP + kxc);
            v[j] = vp[o];
            r[j] = rp[o];
        }
#pragma unroll 1
        for (int yb = 0; yb < Dy; yb += AY) {
#pragma unroll
            for (int i = 0; i < AY; ++i) tn[i] = tp[(unsigned)(min(yb + 2 * AY - 1 + i, Hy - 1) * KXP + kxc)];
#pragma unroll
            for (int j = 0; j < AY; ++j) {
                const unsigned o = (unsigned)(min(yb + AY + j, Dy - 1) * KXP + kxc);
                vn[j] = vp[o];
                rn[j] = rp[o];
            }
#pragma unroll
            for (int j = 0; j < AY; ++j) {
                if (yb + j >= Dy) {   // rows past the data contribute nothing
                    v[j] = {0, 0};
                    r[j] = {0, 0};
                }
#pragma unroll
                for (int a = 0; a < AY; ++a) {
                    cfmac(an[0][a], t[j + a], v[j]);
                    cfmac(ap[0][a], t[j + a], r[j]);
                }
            }
#pragma unroll
            for (int i = 0; i < AY - 1; ++i) t[i] = t[i + AY];
#pragma unroll
            for (int i = 0; i < AY; ++i) t[AY - 1 + i] = tn[i];
#pragma unroll
            for (int j = 0; j < AY; ++j) {
                v[j] = vn[j];
                r[j] = rn[j];
            }
        }
    }
    if (kx >= KX) return;
    const long gplane = (long)AY * KXP, gsize = (long)M * C * gplane;
#pragma unroll
    for (int c = 0; c < CG; ++c) {
        if (c >= C) break;
#pragma unroll
        for (int a = 0; a < AY; ++a) {
            const long o = (long)grp * gsize + ((long)m * C + c) * gplane + (long)a * KXP + kx;
            Gn[o] = an[c][a];
            Gp[o] = ap[c][a];
        }
    }
}

// Two atoms per thread: the V^/R^ row entries -- the operand every atom block re-reads -- are loaded once for both, and
// the address arithmetic and loop overhead of a row are shared.  Rows are taken one at a time through register rings
// that are indexed statically (the row loop is unrolled over one ring period): RS >= AY + 4 slots per atom for the row
// spectra of H (row y needs T[y .. y+AY-1]; the rows behind them are in flight), VS = 8 slots for V^ and R^ (7 rows in
// flight); RS is a multiple of VS so that one period of RS rows serves both rings.
template <typename T, int AY, int GROUPS>
__global__ __launch_bounds__(kMixCols *GROUPS, 2) void k_mix_grad_W2(const cplx<T> *Tsp, const cplx<T> *VT,
                                                                  const cplx<T> *RT, cplx<T> *Gn, cplx<T> *Gp, int N,
                                                                  int M, int Hy, int Dy, int KX, int KXP, int nper,
                                                                  int gx, int gy, int gz) {
    constexpr int MA = 2, VS = 8, RS = (AY + 4 + VS - 1) / VS * VS, P = VS - 1;
    const int col = threadIdx.x & (kMixCols - 1), sub = threadIdx.x / kMixCols;
    // XCD-aware block order.  The logical grid is (atom pairs, kx tiles, group blocks), atom pairs fastest: the gx blocks
    // of one (kx tile, group block) read the SAME V^ / R^ rows.  The hardware deals consecutive workgroups round-robin to
    // the eight XCDs, each with an L2 of its own -- dealt in logical order the sixteen sharers landed on eight L2s and the
    // small operand was fetched eight times (PMC: 4.36 GB fetched for 2.8 GB of H spectra).  Launched as a 1-D grid and
    // remapped so that every XCD walks a contiguous chunk of the logical order, the sharers meet in one L2.
    int lin = blockIdx.x;
    {
        const int total = gx * gy * gz, whole = total / 8 * 8;
        if (lin < whole) lin = (lin & 7) * (whole / 8) + (lin >> 3);
    }
    const int bx = lin % gx, by = (lin / gx) % gy, bz = lin / (gx * gy);
    const int kx = by * kMixCols + col, kxc = min(kx, KX - 1);
    const int m0 = bx * MA, grp = bz * GROUPS + sub;
    const int m1 = min(m0 + 1, M - 1);   // odd M: the second atom of the last block repeats the first (not stored)
    cplx<T> an[MA][AY], ap[MA][AY];
#pragma unroll
    for (int q = 0; q < MA; ++q)
#pragma unroll
        for (int a = 0; a < AY; ++a) {
            an[q][a] = {0, 0};
            ap[q][a] = {0, 0};
        }
    // Addresses: a wave holds GROUPS sample groups, so the sample index differs from lane to lane.  Every address is
    // split into a wave-uniform part -- a buffer descriptor based at (sample nr of the block's first group, atom) plus
    // a scalar row offset -- and a per-lane 32-bit byte offset that never changes (group of the lane, kx).  With whole
    // per-lane 64-bit pointers the address arithmetic (20 64-bit vector operations per row next to 96 packed
    // multiply-adds) cost a third of the kernel.  The host checks that the offsets fit 31 bits.
    static_assert(sizeof(cplx<T>) == 8, "float spectra");
    typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
    const long tplane = (long)Hy * KXP, vplane = (long)Dy * KXP;
    const int nbeg0 = bz * GROUPS * nper, nbeg = nbeg0 + sub * nper;
    const int lane_t = (int)(((long)sub * nper * M * tplane + kxc) * 8);
    const int lane_v = (int)(((long)sub * nper * vplane + kxc) * 8);
    const int rowb = KXP * 8;   // bytes per row of spectra
    auto ld = [](const __amdgpu_buffer_rsrc_t &rs, int lane_off, int row_off) {
        const u32x2v w = __builtin_amdgcn_raw_buffer_load_b64(rs, lane_off, row_off, 0);
        const unsigned re = w[0], im = w[1];   // (bit_cast straight on a vector element reads element 0: hipcc 7.2)
        return cplx<T>{__builtin_bit_cast(float, re), __builtin_bit_cast(float, im)};
    };
#pragma unroll 1
    for (int nr = 0; nr < nper; ++nr) {
        if (nbeg0 + nr >= N) break;    // (uniform) no group of this block has a sample nr
        if (nbeg + nr >= N) continue;  // this lane's group has run out of samples
        const cplx<T> *tb = Tsp + ((long)(nbeg0 + nr) * M + m0) * tplane;
        const __amdgpu_buffer_rsrc_t t0 = __builtin_amdgcn_make_buffer_rsrc((void *)tb, 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t t1 =
            __builtin_amdgcn_make_buffer_rsrc((void *)(tb + (long)(m1 - m0) * tplane), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t vb =
            __builtin_amdgcn_make_buffer_rsrc((void *)(VT + (long)(nbeg0 + nr) * vplane), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb =
            __builtin_amdgcn_make_buffer_rsrc((void *)(RT + (long)(nbeg0 + nr) * vplane), 0, 0x7fffffff, 0x00020000);
        cplx<T> t[MA][RS], v[VS], r[VS];
        // rows 0 .. RS-2 of the H spectra, rows 0 .. P-1 of V^/R^ (row indices clamped: always legal addresses)
#pragma unroll
        for (int i = 0; i < RS - 1; ++i) {
            const int o = min(i, Hy - 1) * rowb;
            t[0][i] = ld(t0, lane_t, o);
            t[1][i] = ld(t1, lane_t, o);
        }
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const int o = min(i, Dy - 1) * rowb;
            v[i] = ld(vb, lane_v, o);
            r[i] = ld(rb, lane_v, o);
        }
#pragma unroll 1
        for (int yb = 0; yb < Dy; yb += RS) {
            // one period of both rings: every ring index below is a compile-time constant
#pragma unroll
            for (int i = 0; i < RS; ++i) {
                const int y = yb + i;
                if (y >= Dy) break;   // wave-uniform
                // fetch: H row y + RS - 1 into the slot row y - 1 left, V^/R^ row y + P into the slot row y - 1 left
                {
                    const int o = min(y + RS - 1, Hy - 1) * rowb;
                    t[0][(i + RS - 1) % RS] = ld(t0, lane_t, o);
                    t[1][(i + RS - 1) % RS] = ld(t1, lane_t, o);
                    const int ov = min(y + P, Dy - 1) * rowb;
                    v[(i + P) % VS] = ld(vb, lane_v, ov);
                    r[(i + P) % VS] = ld(rb, lane_v, ov);
                }
                const cplx<T> vy = v[i % VS], ry = r[i % VS];
#pragma unroll
                for (int a = 0; a < AY; ++a) {
                    cfmac(an[0][a], t[0][(i + a) % RS], vy);
                    cfmac(ap[0][a], t[0][(i + a) % RS], ry);
                    cfmac(an[1][a], t[1][(i + a) % RS], vy);
                    cfmac(ap[1][a], t[1][(i + a) % RS], ry);
                }
            }
        }
    }
    if (kx >= KX) return;
    const long gplane = (long)AY * KXP, gsize = (long)M * gplane;
#pragma unroll
    for (int q = 0; q < MA; ++q) {
        if (m0 + q >= M) break;
#pragma unroll
        for (int a = 0; a < AY; ++a) {
            const long o = (long)grp * gsize + (long)(m0 + q) * gplane + (long)a * KXP + kx;
            Gn[o] = an[q][a];
            Gp[o] = ap[q][a];
        }
    }
}

// 1-D signals (one row per plane, AY == 1), up to CG channels per thread:
//   GT[m,c,kx] = sum_n T[n,m,kx] * conj(VT[n,c,kx])   (and with RT); no lags along y, one entry per (sample, atom, kx)
template <typename T, int CG, int GROUPS>
__global__ __launch_bounds__(kMixCols *GROUPS) void k_mix_grad_W_1d(const cplx<T> *Tsp, const cplx<T> *VT, const cplx<T> *RT,
                                                                  cplx<T> *Gn, cplx<T> *Gp, int N, int M, int C, int KX,
                                                                  int KXP, int nper) {
    const int col = threadIdx.x & (kMixCols - 1), sub = threadIdx.x / kMixCols;
    const int kx = blockIdx.y * kMixCols + col, kxc = min(kx, KX - 1);
    const int m = blockIdx.x, grp = blockIdx.z * GROUPS + sub;
    cplx<T> an[CG], ap[CG];
#pragma unroll
    for (int c = 0; c < CG; ++c) {
        an[c] = {0, 0};
        ap[c] = {0, 0};
    }
    const int nbeg = grp * nper, nend = min(N, nbeg + nper);
    for (int n = nbeg; n < nend; ++n) {
        const cplx<T> t = Tsp[((long)n * M + m) * KXP + kxc];
#pragma unroll
        for (int c = 0; c < CG; ++c) {
            const long o = ((long)n * C + min(c, C - 1)) * KXP + kxc;
            cfmac(an[c], t, VT[o]);
            cfmac(ap[c], t, RT[o]);
        }
    }
    if (kx >= KX) return;
    const long gsize = (long)M * C * KXP;
#pragma unroll
    for (int c = 0; c < CG; ++c) {
        if (c >= C) break;
        const long o = (long)grp * gsize + ((long)m * C + c) * KXP + kx;
        Gn[o] = an[c];
        Gp[o] = ap[c];
    }
}

template <typename T, int AY>
int launch_mix_grad_W(const void *Tsp, const void *VT, const void *RT, void *Gn, void *Gp, const Geo &g, int KX,
                      int KXP, int ngroups, int nper, hipStream_t s) {
    constexpr int GROUPS = 4;
    if constexpr (AY == 1) {
        if (g.Dy == 1) {   // 1-D signals: all channels of an atom in one thread
            const dim3 grid((unsigned)g.M, (unsigned)cdiv(KX, kMixCols), (unsigned)cdiv(ngroups, GROUPS));
            hipLaunchKernelGGL((k_mix_grad_W_1d<T, 4, GROUPS>), grid, dim3(kMixCols * GROUPS), 0, s, (const cplx<T> *)Tsp,
                               (const cplx<T> *)VT, (const cplx<T> *)RT, (cplx<T> *)Gn, (cplx<T> *)Gp, g.N, g.M, g.C, KX,
                               KXP, nper);
            TNMF_LAUNCH_CHECK();
            return TNMF_OK;
        }
    }
    if constexpr (AY <= 12) {   // (taller atoms: the two-atom variant no longer fits the register file)
        // per-lane byte offsets of the two-atom kernel: (group within the block) * nper samples of row spectra, plus
        // a scalar row offset -- they must stay below 2^31 (the single-atom kernel below has no such limit)
        const long span = ((long)(GROUPS - 1) * nper * g.M + 2) * g.Hy * KXP * 8;
        if (g.C == 1 && span < (1L << 31)) {
            const int gx = cdiv(g.M, 2), gy = cdiv(KX, kMixCols), gz = cdiv(ngroups, GROUPS);
            hipLaunchKernelGGL((k_mix_grad_W2<T, AY, GROUPS>), dim3((unsigned)(gx * gy * gz)), dim3(kMixCols * GROUPS), 0, s,
                               (const cplx<T> *)Tsp, (const cplx<T> *)VT, (const cplx<T> *)RT, (cplx<T> *)Gn,
                               (cplx<T> *)Gp, g.N, g.M, g.Hy, g.Dy, KX, KXP, nper, gx, gy, gz);
            TNMF_LAUNCH_CHECK();
            return TNMF_OK;
        }
    }
    const dim3 grid((unsigned)g.M, (unsigned)cdiv(KX, kMixCols), (unsigned)cdiv(ngroups, GROUPS));
    hipLaunchKernelGGL((k_mix_grad_W<T, AY, 1, GROUPS>), grid, dim3(kMixCols * GROUPS), 0, s, (const cplx<T> *)Tsp,
                       (const cplx<T> *)VT, (const cplx<T> *)RT, (cplx<T> *)Gn, (cplx<T> *)Gp, g.N, g.M, g.C, g.Hy, g.Dy,
                       KX, KXP, nper);
    TNMF_LAUNCH_CHECK();
    return TNMF_OK;
}

#define MIX_SWITCH(FN, ...)                                   \
    switch (g.Ay) {                                           \
        case 1: return FN<float, 1>(__VA_ARGS__);             \
        case 2: return FN<float, 2>(__VA_ARGS__);             \
        case 3: return FN<float, 3>(__VA_ARGS__);             \
        case 4: return FN<float, 4>(__VA_ARGS__);             \
        case 5: return FN<float, 5>(__VA_ARGS__);             \
        case 6: return FN<float, 6>(__VA_ARGS__);             \
        case 7: return FN<float, 7>(__VA_ARGS__);             \
        case 8: return FN<float, 8>(__VA_ARGS__);             \
        case 9: return FN<float, 9>(__VA_ARGS__);             \
        case 10: return FN<float, 10>(__VA_ARGS__);           \
        case 11: return FN<float, 11>(__VA_ARGS__);           \
        case 12: return FN<float, 12>(__VA_ARGS__);           \
        case 13: return FN<float, 13>(__VA_ARGS__);           \
        case 14: return FN<float, 14>(__VA_ARGS__);           \
        case 15: return FN<float, 15>(__VA_ARGS__);           \
        case 16: return FN<float, 16>(__VA_ARGS__);           \
        default: return TNMF_E_UNSUPPORTED;                   \
    }

}  // namespace

// float32, one channel, atoms up to 16 rows.  With several channels the column-transform kernels of fft_kernels.h,
// whose cost does not grow with C*Ay, are faster (measured with the three-channel variant of k_mix_reconstruct, which
// is kept instantiated: config 4 reconstruct 2.5 -> 3.3 ms, config 5 11.0 -> 15.5 ms), and the W gradient's two
// accumulator sets per channel no longer fit the registers.
// 1-D signals (Dy == Ay == 1): nothing to contract along y, so up to three / four channels fit as well.
bool mixed_has_reconstruct(const Geo &g, int dtype) {
    return dtype == 0 && g.Ay <= 16 && (g.C == 1 || (g.Dy == 1 && g.Ay == 1 && g.C <= 3));
}
bool mixed_has_grad_W(const Geo &g, int dtype) {
    return dtype == 0 && g.Ay <= 16 && (g.C == 1 || (g.Dy == 1 && g.Ay == 1 && g.C <= 4));
}

int mixed_reconstruct(const Geo &g, const void *Tsp, const void *WT, void *OT, int KX, int KXP, hipStream_t s) {
    MIX_SWITCH(launch_mix_reconstruct, Tsp, WT, OT, g, KX, KXP, s);
}

int mixed_grad_W(const Geo &g, const void *Tsp, const void *VT, const void *RT, void *Gn, void *Gp, int KX, int KXP,
                 int ngroups, int nper, hipStream_t s) {
    MIX_SWITCH(launch_mix_grad_W, Tsp, VT, RT, Gn, Gp, g, KX, KXP, ngroups, nper, s);
}
