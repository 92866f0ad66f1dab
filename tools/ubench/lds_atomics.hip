// Micro-benchmark: LDS atomic / read / write throughput per CU on gfx950 for the accumulation patterns of satba.
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/ubench/lds_atomics.hip -o /tmp/lds_atomics && /tmp/lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int TABLE = 4096;  // doubles (32 KB)
constexpr int ITERS = 2048;

__device__ inline unsigned rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int MODE>
__global__ __launch_bounds__(1024) void k(double* out, int spread) {
    __shared__ double tab[TABLE];
    for (int i = threadIdx.x; i < TABLE; i += blockDim.x) tab[i] = 0.0;
    __syncthreads();
    unsigned s = threadIdx.x * 9781u + blockIdx.x * 7919u + 1u;
    double acc = 0.0;
    for (int it = 0; it < ITERS; ++it) {
        const int idx = (spread == 0) ? ((threadIdx.x + it * 67) & (TABLE - 1)) : (rnd(s) & (TABLE - 1));
        if (MODE == 0) atomicAdd(&tab[idx], 1.0);                                           // ds_add_f64
        else if (MODE == 1) atomicAdd((unsigned long long*)&tab[idx], 1ull);                // ds_add_u64
        else if (MODE == 2) atomicAdd((float*)&tab[idx], 1.0f);                             // ds_add_f32
        else if (MODE == 3) atomicAdd((unsigned*)&tab[idx], 1u);                            // ds_add_u32
        else if (MODE == 4) acc += tab[idx];                                                // ds_read_b64
        else if (MODE == 5) tab[idx] = (double)it;                                          // ds_write_b64
        else if (MODE == 6) acc += (double)atomicAdd((unsigned*)&tab[idx], 1u);             // ds_add_rtn_u32
        else if (MODE == 7) { const double2 t = *reinterpret_cast<const double2*>(&tab[idx & ~1]); acc += t.x + t.y; }  // ds_read_b128
        else if (MODE == 8) *reinterpret_cast<double2*>(&tab[idx & ~1]) = make_double2((double)it, 1.0);           // ds_write_b128
        else if (MODE == 9) { acc += tab[idx] + tab[(idx + 33 * 8) & (TABLE - 1)]; }                               // two b64 reads (ds_read2_b64 when the offsets allow)
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = acc + tab[5];
}

template <int MODE>
void run(const char* name, int spread) {
    double* out;
    hipMalloc(&out, 8 * 4096);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256;  // one per CU
    k<MODE><<<blocks, 1024>>>(out, spread);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE><<<blocks, 1024>>>(out, spread);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double lane_ops = (double)blocks * 1024 * ITERS;
    printf("%-18s %-8s %8.3f ms  %7.2f lane-ops/clk/CU (at 2.4 GHz)  %8.1f G lane-ops/s chip\n", name,
           spread ? "random" : "strided", ms, lane_ops / blocks / (ms * 1e-3 * 2.4e9), lane_ops / (ms * 1e-3) / 1e9);
    hipFree(out);
}

__global__ void kg(double* g, int n, int spread) {
    unsigned s = threadIdx.x * 9781u + blockIdx.x * 7919u + 1u;
    for (int it = 0; it < 256; ++it) {
        const int idx = rnd(s) % n;
        atomicAdd(&g[idx], 1.0);
    }
}

int main() {
    for (int spread = 0; spread < 2; ++spread) {
        run<0>("ds_add_f64", spread);
        run<1>("ds_add_u64", spread);
        run<2>("ds_add_f32", spread);
        run<3>("ds_add_u32", spread);
        run<6>("ds_add_rtn_u32", spread);
        run<4>("ds_read_b64", spread);
        run<5>("ds_write_b64", spread);
        run<7>("ds_read_b128", spread);
        run<8>("ds_write_b128", spread);
        run<9>("ds_read2_b64", spread);
    }
    for (int n : {1000000, 15000, 1000}) {
        double* g;
        hipMalloc(&g, 8 * (size_t)n);
        hipMemset(g, 0, 8 * (size_t)n);
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        kg<<<2048, 256>>>(g, n, 1);
        hipDeviceSynchronize();
        hipEventRecord(a);
        kg<<<2048, 256>>>(g, n, 1);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        printf("global_atomic_add_f64 over %8d addresses: %8.3f ms  %8.2f G atomics/s\n", n, ms, 2048.0 * 256 * 256 / (ms * 1e-3) / 1e9);
        hipFree(g);
    }
    return 0;
}
