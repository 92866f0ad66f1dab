// Micro-benchmark (round 6): what the bank conflicts of k_linearize's LDS traffic cost, and what a conflict-aware order of a
// point's observations could buy.  One workgroup of 1024 threads per CU emulates the kernel's per-observation LDS pattern:
//   13 ds_add_u64 into row (cam << rep_shift | lane & (replicas - 1)) of a table with row stride 15 (cam_sum_stride(5)),
//   optionally 10 ds_read_b64 of the camera-constant row (stride 21 here: odd, like CAMC),
// with the camera of a lane drawn in four ways:
//   random     uniform over M cameras (what a slot of the sliced ELL looks like today)
//   band       uniform over a band of 60 cameras (slot k of camera-ascending tracks)
//   half       the 32 lanes of each half wave hit 32 different residues cam mod 32 (what a scheduled slot order could reach)
//   quarter    the 16 lanes of each quarter wave hit 16 different residues mod 32 (a weaker schedule)
// hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_conflicts.hip -o tools/ubench/lds_conflicts.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int M = 200;
constexpr int CUS = 15, CAMC = 21;
constexpr int ITERS = 1024;

__device__ inline unsigned rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

// camera of (lane, iteration) for pattern `pat`; wave-uniform seed su makes the residues of the scheduled patterns a random
// permutation per iteration
__device__ inline int draw_cam(int pat, int lane, unsigned& s, unsigned su, int rep_shift) {
    const unsigned r = rnd(s);
    if (pat == 0) return r % M;
    if (pat == 1) return 40 + r % 60;
    const int group = (pat == 2) ? 32 : 16;
    if (rep_shift == 2) {
        // four replicas chosen by lane & 3: rows (cam << 2 | lane & 3) are distinct mod 32 when the lanes that share lane & 3
        // inside the group see cameras distinct mod 8
        const int m = (lane & (group - 1)) >> 2;        // 0 .. 7 (half) or 0 .. 3 (quarter)
        const int res = (m * 3 + su) & 7;
        return res + 8 * (r % ((M - res + 7) / 8));
    }
    const int j = lane & (group - 1);
    const int res = (j * 5 + su) & 31;                 // distinct residues mod 32 inside the group (5 is odd: a permutation for group 32)
    const int hi = r % ((M - res + 31) / 32);          // cameras res, res + 32, ...
    return res + 32 * hi;
}

template <int NADD, int NREAD>
__global__ __launch_bounds__(1024) void k(double* out, int pat, int rep_shift) {
    extern __shared__ unsigned long long tab[];
    const int rows = M << rep_shift;
    double* camc = reinterpret_cast<double*>(tab + (size_t)rows * CUS);
    for (int i = threadIdx.x; i < rows * CUS; i += 1024) tab[i] = 0ull;
    for (int i = threadIdx.x; i < M * CAMC; i += 1024) camc[i] = 1.0 + i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned s = threadIdx.x * 9781u + blockIdx.x * 7919u + 1u;
    unsigned su = (threadIdx.x >> 6) * 31u + blockIdx.x * 17u;
    double acc = 0.0;
    for (int it = 0; it < ITERS; ++it) {
        su = su * 1103515245u + 12345u;
        const int cam = draw_cam(pat, lane, s, __builtin_amdgcn_readfirstlane(su >> 10), rep_shift);
        const int row = (cam << rep_shift) | (lane & ((1 << rep_shift) - 1));
        if (NREAD) {
            const double* c = camc + cam * CAMC;
#pragma unroll
            for (int i = 0; i < NREAD; ++i) acc += c[i];
        }
        unsigned long long* a = tab + (size_t)row * CUS;
#pragma unroll
        for (int i = 0; i < NADD; ++i) atomicAdd(a + i, (unsigned long long)(it + i));
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = acc + (double)tab[5];
}

template <int NADD, int NREAD>
void run(const char* what, int pat, int rep_shift) {
    static const char* names[4] = {"random", "band60", "half-wave distinct", "quarter-wave distinct"};
    double* out;
    hipMalloc(&out, 8 * 4096);
    const size_t lds = sizeof(unsigned long long) * ((size_t)(M << rep_shift) * CUS) + sizeof(double) * M * CAMC;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<NADD, NREAD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<NADD, NREAD><<<256, 1024, lds>>>(out, pat, rep_shift);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) k<NADD, NREAD><<<256, 1024, lds>>>(out, pat, rep_shift);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms /= 5;
    const double obs = 256.0 * 1024 * ITERS;
    printf("%-22s %-22s replicas %2d  %7.3f ms  per 10 M observations: %7.1f us   (%6.1f clk per wave-observation and CU at 2.4 GHz)\n", what,
           names[pat], 1 << rep_shift, ms, ms * 1e3 * 1e7 / obs, ms * 1e-3 * 2.4e9 / (1024.0 / 64 * ITERS));
    hipFree(out);
}

int main() {
    for (int pat = 0; pat < 4; ++pat) {
        for (int rs : {0, 2}) {
            run<13, 0>("13 ds_add_u64", pat, rs);
            run<13, 10>("13 add + 10 read_b64", pat, rs);
        }
        run<0, 10>("10 ds_read_b64", pat, 0);
        run<0, 20>("20 ds_read_b64", pat, 0);
    }
    return 0;
}
