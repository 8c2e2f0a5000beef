// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access shape of the COLUMN kernels of the FFT
// family at the long transform lengths (k_fft_cols_fwd<float, 576, 512> at BASELINE config 5): a workgroup owns a tile of
// 8 kx columns of one plane of row spectra [planes][rows][KXP] complex64 -- per wave-level instruction 8 rows x 8 lanes x
// 8 bytes = 64-byte HALF lines at the row stride KXP * 8 bytes; the other half of every line belongs to the
// neighbouring tile (is the 2.4x over-fetch the PMC passes show real, or a counter artefact of half lines?).
// Every kernel reads a KNOWN number of bytes exactly once from a buffer far larger than the 256 MiB Infinity Cache (and
// the *_rw kernels write as many to a second buffer):
//   k_cols<8,  false>  8-column tiles, blocks in natural grid order (tile index fastest): the two halves of a line are
//                      read by workgroups that the hardware deals to DIFFERENT XCDs (different L2s)
//   k_cols<8,  true>   the same tiles, blockIdx remapped so that every XCD walks a contiguous chunk of the logical order:
//                      the two halves of a line meet in one L2
//   k_cols<16, false>  16-column tiles: whole 128-byte lines (control)
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/fetch_calib_cols.hip -o tools/probes/fetch_calib_cols
// Run:   rocprofv3 --kernel-trace --pmc FETCH_SIZE ... -- tools/probes/fetch_calib_cols   (second pass: WRITE_SIZE)
#include <hip/hip_runtime.h>

#include <cstdio>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_));                  \
            return 1;                                                              \
        }                                                                          \
    } while (0)

template <int CT, bool REMAP, bool WRITE>
__global__ __launch_bounds__(512) void k_cols(const float2 *__restrict__ src, float2 *__restrict__ dst, float *out,
                                              int tiles, int planes, int rows, int KXP) {
    unsigned b = blockIdx.x;
    if (REMAP) {   // every XCD (b % 8) walks a contiguous chunk of the logical order
        const unsigned nwg = gridDim.x, per = nwg / 8, x = b % 8, i = b / 8;
        b = x < 8 && i < per ? x * per + i : b;   // (grids here are multiples of 8)
    }
    const int tile = b % tiles;
    const long plane = b / tiles;
    const int col = threadIdx.x % CT, y0 = threadIdx.x / CT;
    const float2 *p = src + plane * ((long)rows * KXP) + tile * CT + col;
    float2 *q = dst + plane * ((long)rows * KXP) + tile * CT + col;
    float2 acc = {0.f, 0.f};
    for (int y = y0; y < rows; y += 512 / CT) {
        const float2 v = p[(long)y * KXP];
        if (WRITE)
            q[(long)y * KXP] = make_float2(v.x * 1.0001f, v.y);
        else
            acc.x += v.x, acc.y += v.y;
    }
    if (!WRITE && acc.x + acc.y == 12345.f) out[0] = 1.f;
}

template <int CT, bool REMAP, bool WRITE>
static int run(const float2 *src, float2 *dst, float *out, int planes, int rows, int KXP, const char *what) {
    const int tiles = KXP / CT;
    hipLaunchKernelGGL((k_cols<CT, REMAP, WRITE>), dim3(tiles * planes), dim3(512), 0, 0, src, dst, out, tiles, planes, rows,
                       KXP);
    CK(hipDeviceSynchronize());
    const double bytes = (double)planes * rows * KXP * 8;
    printf("k_cols<%d,%d,%d> %-40s reads %.0f bytes%s\n", CT, (int)REMAP, (int)WRITE, what, bytes,
           WRITE ? " and writes as many" : "");
    return 0;
}

int main() {
    // config-5-like: 64 samples x 64 atoms = 4096 planes of 527 rows x 272 complex64 (Lx = 540: 271 frequencies)
    const int planes = 4096, rows = 527, KXP = 272;
    const size_t elems = (size_t)planes * rows * KXP;
    float2 *src, *dst;
    float *out;
    CK(hipMalloc(&src, elems * 8));
    CK(hipMalloc(&dst, elems * 8));
    CK(hipMalloc(&out, 4));
    CK(hipMemset(src, 0, elems * 8));
    CK(hipMemset(dst, 0, elems * 8));
    if (run<8, false, false>(src, dst, out, planes, rows, KXP, "half lines, natural block order")) return 1;
    if (run<8, true, false>(src, dst, out, planes, rows, KXP, "half lines, XCD-contiguous block order")) return 1;
    if (run<16, false, false>(src, dst, out, planes, rows, KXP, "whole lines (control)")) return 1;
    if (run<8, false, true>(src, dst, out, planes, rows, KXP, "half lines, natural order, read+write")) return 1;
    if (run<8, true, true>(src, dst, out, planes, rows, KXP, "half lines, XCD-contiguous, read+write")) return 1;
    if (run<16, false, true>(src, dst, out, planes, rows, KXP, "whole lines, read+write (control)")) return 1;
    return 0;
}
