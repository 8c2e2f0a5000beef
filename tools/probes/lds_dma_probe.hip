// Probe: semantics of __builtin_amdgcn_global_load_lds on gfx950 (LDS destination = wave-uniform base + lane*size?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

__global__ void k_probe(const float *src, float *out, int nactive) {
    __shared__ float buf[4 * 64 + 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 4 * 64 + 64; i += blockDim.x) buf[i] = -1.f;
    __syncthreads();
    // each wave copies 64 floats: lane l reads src[wave*1000 + 3*l] (strided gather) -> buf[wave*64 + l]
    if (lane < nactive)
        __builtin_amdgcn_global_load_lds((gbl_void *)(src + wave * 1000 + 3 * lane), (lds_void *)(buf + wave * 64), 4, 0, 0);
    __syncthreads();
    out[threadIdx.x] = buf[threadIdx.x];
}

int main() {
    std::vector<float> h(8192);
    for (int i = 0; i < 8192; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, 8192 * 4);
    hipMalloc(&o, 256 * 4);
    hipMemcpy(d, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    for (int nactive : {64, 4}) {
        k_probe<<<1, 256>>>(d, o, nactive);
        std::vector<float> r(256);
        hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int t = 0; t < 256; ++t) {
            const int w = t / 64, l = t % 64;
            const float want = l < nactive ? (float)(w * 1000 + 3 * l) : -1.f;
            if (r[t] != want) { if (bad < 8) printf("nactive %d: thread %d got %g want %g\n", nactive, t, r[t], want); ++bad; }
        }
        printf("nactive=%d mismatches=%d\n", nactive, bad);
    }
    return 0;
}
