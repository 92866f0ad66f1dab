// Latency of dependent operations in a single wave on gfx950 (ns per operation, wall clock 100 MHz):
// what bounds the serial chain of the diagonal-block Cholesky.  Build: hipcc --offload-arch=gfx950 -O3 latency.hip -o latency.bin
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int N = 4096;

__device__ inline double readlane_f64(double v, int l) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, l);
    hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}

template <int MODE>
__global__ void k(double* out, long long* t, double seed) {
    __shared__ double sh[64];
    double x = seed + threadIdx.x * 1e-9, y = 1.0000001;
    sh[threadIdx.x] = x;
    __syncthreads();
    const long long t0 = wall_clock64();
#pragma unroll 16
    for (int i = 0; i < N; ++i) {
        if (MODE == 0) x = fma(x, y, 1e-9);                                    // dependent f64 fma
        if (MODE == 1) x = __builtin_amdgcn_rsq(x) + 1.0;                      // rsq + add
        if (MODE == 2) x = readlane_f64(x, i & 31) * y;                        // readlane (2x) + mul
        if (MODE == 3) { sh[threadIdx.x] = x; __builtin_amdgcn_wave_barrier(); x = sh[(i + 1) & 63] * y; __builtin_amdgcn_wave_barrier(); }  // LDS write -> broadcast read -> mul
        if (MODE == 4) x = __shfl(x, i & 31) * y;                              // ds_bpermute + mul
        if (MODE == 5) x = (float)((float)x * 1.0001f + 1e-9f);                // f32 chain (cvt + fma)
        if (MODE == 6) x = __builtin_amdgcn_mov_dpp(__double2loint(x), 0x130, 0xf, 0xf, false) + x * y;  // dpp mov + fma
    }
    const long long t1 = wall_clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) t[MODE] = t1 - t0;
}

int main() {
    double* out; long long* t;
    hipMalloc(&out, 64 * 8); hipMalloc(&t, 8 * 8);
    for (int rep = 0; rep < 2; ++rep) {
        k<0><<<1, 64>>>(out, t, 1.0); k<1><<<1, 64>>>(out, t, 1.0); k<2><<<1, 64>>>(out, t, 1.0); k<3><<<1, 64>>>(out, t, 1.0);
        k<4><<<1, 64>>>(out, t, 1.0); k<5><<<1, 64>>>(out, t, 1.0); k<6><<<1, 64>>>(out, t, 1.0);
    }
    hipDeviceSynchronize();
    long long h[8];
    hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"f64 fma", "rsq_f64 + add_f64", "2 readlane + mul_f64", "LDS write + read + mul_f64", "ds_bpermute x2 + mul", "f32 cvt chain", "dpp + fma"};
    for (int m = 0; m < 7; ++m) printf("%-28s %7.2f ns per iteration\n", names[m], h[m] * 10.0 / N);
    return 0;
}
