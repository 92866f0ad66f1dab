// Latencies the dense Cholesky's serial chain is made of, measured on one wave of one CU (and with 15 idle-polling waves beside it):
// dependent v_fma_f64, the rsqrt sequence of a pivot, v_readlane of a freshly written register, an LDS write -> flag -> read hand-over,
// dependent v_mfma_f64_16x16x4.  clock64() counts shader clocks, wall_clock64() the 100 MHz reference: their ratio is the clock.
//   hipcc --offload-arch=gfx950 -O3 dp_latency.hip -o dp_latency && ./dp_latency
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ inline double readlane_f64(double v, int l) { int lo = __double2loint(v), hi = __double2hiint(v); lo = __builtin_amdgcn_readlane(lo, l); hi = __builtin_amdgcn_readlane(hi, l); return __hiloint2double(hi, lo); }
__device__ inline double half_rsqrt(double d) { const double y = __builtin_amdgcn_rsq(d); double g = d * y, h = 0.5 * y; double r = fma(-h, g, 0.5); g = fma(g, r, g); h = fma(h, r, h); r = fma(-h, g, 0.5); return fma(h, r, h); }

__global__ void k(double* out, long long* t, int n, int pollers) {
    __shared__ double sh[128];
    __shared__ int flag;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) flag = 0;
    __syncthreads();
    if (wave > 0) {  // idle waves polling an LDS word like the waiting waves of the tile kernel
        if (pollers) while (__hip_atomic_load(&flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
        return;
    }
    double x = 1.0 + lane * 1e-3, y = 0.999;
    long long c0, c1, w0, w1;
    // 1. dependent FMA chain
    w0 = wall_clock64(); c0 = clock64();
    for (int i = 0; i < n; ++i) { x = fma(x, y, 1e-9); x = fma(x, y, 1e-9); x = fma(x, y, 1e-9); x = fma(x, y, 1e-9); }
    c1 = clock64(); w1 = wall_clock64();
    if (lane == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
    // 2. pivot sequence: rsqrt + refinement, dependent
    double d = 2.0 + x * 1e-6;
    c0 = clock64();
    for (int i = 0; i < n; ++i) { const double h = half_rsqrt(d); d = fma(h, 1e-3, 2.0); }
    c1 = clock64();
    if (lane == 0) t[2] = c1 - c0;
    // 3. readlane of a just-written register + dependent fma
    double z = x;
    c0 = clock64();
    for (int i = 0; i < n; ++i) { const double s = readlane_f64(z, 5); z = fma(z, 0.5, s * 1e-3); }
    c1 = clock64();
    if (lane == 0) t[3] = c1 - c0;
    // 4. LDS: write, read back another lane's value (no flag), dependent
    double u = x;
    c0 = clock64();
    for (int i = 0; i < n; ++i) { sh[lane] = u; u = sh[(lane + 1) & 63] * 0.5 + 0.25; }
    c1 = clock64();
    if (lane == 0) t[4] = c1 - c0;
    // 5. dependent MFMA f64
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    c0 = clock64();
    for (int i = 0; i < n; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc, 0, 0, 0);
    c1 = clock64();
    if (lane == 0) t[5] = c1 - c0;
    // 6. independent MFMA f64 (4 accumulators)
    d4 a1 = acc, a2 = acc, a3 = acc;
    c0 = clock64();
    for (int i = 0; i < n; ++i) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
                                  a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0); }
    c1 = clock64();
    if (lane == 0) t[6] = c1 - c0;
    // 7. v_rsq_f64 alone, dependent
    double q = 2.0;
    c0 = clock64();
    for (int i = 0; i < n; ++i) { q = __builtin_amdgcn_rsq(q) + 1.5; }
    c1 = clock64();
    if (lane == 0) t[7] = c1 - c0;
    out[lane] = x + d + z + u + acc[0] + a1[1] + a2[2] + a3[3] + q;
    if (lane == 0) __hip_atomic_store(&flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
int main() {
    double* out; long long* t; hipMalloc(&out, 64 * 8); hipMalloc(&t, 64);
    const int n = 20000;
    for (int pass = 0; pass < 3; ++pass) {
        const int threads = pass == 0 ? 64 : 1024, pollers = pass == 2;
        hipMemset(t, 0, 64);
        hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, out, t, n, pollers);
        hipDeviceSynchronize();
        long long h[8]; hipMemcpy(h, t, 64, hipMemcpyDeviceToHost);
        printf("%s: clock %.2f GHz (clock64 / wall)  | dependent fma %.1f clk | rsqrt+refine+fma %.1f clk | readlane+fma %.1f clk | LDS write->read %.1f clk | dependent mfma_f64 %.1f clk | independent mfma_f64 %.1f clk | rsq+add %.1f clk\n",
               pass == 0 ? "1 wave        " : (pass == 1 ? "16 waves, 15 exit" : "16 waves, 15 poll"), (double)h[0] / h[1] * 0.1, h[0] / (4.0 * n), h[2] / (double)n, h[3] / (double)n, h[4] / (double)n, h[5] / (double)n,
               h[6] / (4.0 * n), h[7] / (double)n);
    }
    return 0;
}
