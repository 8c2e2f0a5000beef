// fft_len.hip -- one object per transform length: compiled with -DTNMF_FFT_L=<L> (see Makefile), it instantiates the
// kernels of fft_kernels.h for that length (float always, double for the lengths up to 288: the float64 parity tests up to the BASELINE config-3 geometry).
#include "fft_kernels.h"

#ifndef TNMF_FFT_L
#error "compile with -DTNMF_FFT_L=<transform length>"
#endif
#define TNMF_CAT_(a, b) a##b
#define TNMF_CAT(a, b) TNMF_CAT_(a, b)

int TNMF_CAT(fft_run_, TNMF_FFT_L)(int op, int dtype, const FftArgs *a, hipStream_t s) {
    constexpr int L = TNMF_FFT_L;
    if (dtype == 0) return fft_run_typed<float, L>(op, a, s);
    if constexpr (L <= 288) return fft_run_typed<double, L>(op, a, s);
    return TNMF_E_UNSUPPORTED;
}
