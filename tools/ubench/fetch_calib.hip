// Calibration of rocprofv3's FETCH_SIZE for the access pattern of k_schur_pairs: the 80 used bytes of
// 128-byte records at known indices, fetched cooperatively as 16-byte pieces (load t of lane l = piece 64 t + l, i.e. piece
// (64 t + l) % 5 of the record of hit (64 t + l) / 5), against two references whose byte counts are beyond doubt:
//   stream16   every lane reads 16 consecutive bytes of a large array once (the guide: FETCH_SIZE reports 1/2 of these bytes)
//   gather80   the pattern above over a random permutation of ALL records, each record exactly once: the bytes that must come
//              from memory are 2 x 64-byte sectors per record (bytes 0..79 of a 128-byte line) = N x 128, or N x 80 if only
//              the touched 16-byte pieces counted
//   gather80r  the same with every record gathered 8 times by different waves in a row (7 of 8 hits served by the caches)
// Run each kernel under  rocprofv3 --pmc FETCH_SIZE --kernel-trace  (and TCC_EA0_RDREQ_sum, TCC_EA0_RDREQ_32B_sum in a second
// pass) and compare the counter with the known bytes printed here.   usage: ./fetch_calib.bin [records]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__global__ __launch_bounds__(256) void k_stream16(const double2* __restrict__ a, size_t n16, double* __restrict__ out) {
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const double2 v = a[i]; s += v.x + v.y; }
    if (s == 1.2345e-300) out[0] = s;
}
// one wave per 64 hits; idx[h] = record of hit h; the 320 pieces of the 64 records in 5 loads per lane
__global__ __launch_bounds__(256) void k_gather80(const double2* __restrict__ rec, const int* __restrict__ idx, long long n_hits, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    double s = 0.0;
    for (long long h0 = wave * 64; h0 < n_hits; h0 += n_waves * 64) {
        const int my = (h0 + lane < n_hits) ? idx[h0 + lane] : idx[n_hits - 1];
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            const int piece = 64 * t + lane, hit = piece / 5, part = piece % 5;
            const int r = __shfl(my, hit);
            const double2 v = rec[(size_t)r * 8 + part];  // 128-byte records = 8 pieces, the first 5 are used
            s += v.x + v.y;
        }
    }
    if (s == 1.2345e-300) out[0] = s;
}
// NPC pieces (16 NPC bytes) of records with a stride of STRIDE pieces: 4 of 8 = the first 64-byte half of a 128-byte line per hit
// (does a miss move a half line?), 4 of 4 = compact 64-byte records, two per line
template <int NPC, int STRIDE>
__global__ __launch_bounds__(256) void k_gather(const double2* __restrict__ rec, const int* __restrict__ idx, long long n_hits, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    double s = 0.0;
    for (long long h0 = wave * 64; h0 < n_hits; h0 += n_waves * 64) {
        const int my = (h0 + lane < n_hits) ? idx[h0 + lane] : idx[n_hits - 1];
#pragma unroll
        for (int t = 0; t < NPC; ++t) {
            const int piece = 64 * t + lane, hit = piece / NPC, part = piece % NPC;
            const int r = __shfl(my, hit);
            const double2 v = rec[(size_t)r * STRIDE + part];
            s += v.x + v.y;
        }
    }
    if (s == 1.2345e-300) out[0] = s;
}
int main(int argc, char** argv) {
    const long long N = argc > 1 ? atoll(argv[1]) : 4000000;  // 4 M records = 512 MB: twice the Infinity Cache
    double2* rec; int *idx, *idx8; double* out;
    CK(hipMalloc(&rec, (size_t)N * 128)); CK(hipMalloc(&idx, sizeof(int) * N)); CK(hipMalloc(&idx8, sizeof(int) * N * 8)); CK(hipMalloc(&out, 8));
    CK(hipMemset(rec, 0, (size_t)N * 128));
    std::vector<int> perm(N); std::iota(perm.begin(), perm.end(), 0);
    std::mt19937_64 g(1); std::shuffle(perm.begin(), perm.end(), g);
    CK(hipMemcpy(idx, perm.data(), sizeof(int) * N, hipMemcpyHostToDevice));
    // repeated: blocks of 64 records, each block gathered by 8 consecutive waves
    std::vector<int> rep((size_t)N * 8);
    for (long long b = 0; b < N / 64; ++b) for (int k = 0; k < 8; ++k) for (int l = 0; l < 64; ++l) rep[((size_t)b * 8 + k) * 64 + l] = perm[b * 64 + (l * 7 + k) % 64];
    CK(hipMemcpy(idx8, rep.data(), sizeof(int) * N * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timed = [&](const char* name, auto launch, double bytes_lo, double bytes_hi) {
        launch();  // warm-up (page tables)
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-10s %.3f ms   known bytes from memory: %.1f MB (touched 16-byte pieces only: %.1f MB)  -> %.0f GB/s\n", name, ms, bytes_hi / 1e6, bytes_lo / 1e6, bytes_hi / ms / 1e6);
    };
    timed("stream16", [&] { hipLaunchKernelGGL(k_stream16, dim3(2048), dim3(256), 0, 0, rec, (size_t)N * 8, out); }, N * 128.0, N * 128.0);
    timed("gather80", [&] { hipLaunchKernelGGL(k_gather80, dim3(4096), dim3(256), 0, 0, rec, idx, N, out); }, N * 80.0, N * 128.0);
    timed("gather80r", [&] { hipLaunchKernelGGL(k_gather80, dim3(4096), dim3(256), 0, 0, rec, idx8, N * 8, out); }, N * 80.0, N * 128.0);
    // half lines: 4 M hits, each the first 64 bytes of its own 128-byte line
    timed("gather64h", [&] { hipLaunchKernelGGL((k_gather<4, 8>), dim3(4096), dim3(256), 0, 0, rec, idx, N, out); }, N * 64.0, N * 128.0);
    // compact 64-byte records, a random HALF of them (the even ones: one record per line, the neighbour is never asked for) ...
    std::vector<int> even(N);
    for (long long k = 0; k < N; ++k) even[k] = 2 * perm[k];
    CK(hipMemcpy(idx, even.data(), sizeof(int) * N, hipMemcpyHostToDevice));
    timed("gather64c1", [&] { hipLaunchKernelGGL((k_gather<4, 4>), dim3(4096), dim3(256), 0, 0, rec, idx, N, out); }, N * 64.0, N * 128.0);
    // ... and neighbours together (hits 2 k, 2 k + 1 = the two records of one line): 2 M lines for 4 M hits
    for (long long k = 0; k < N; k += 2) { even[k] = 2 * perm[k / 2]; even[k + 1] = 2 * perm[k / 2] + 1; }
    CK(hipMemcpy(idx, even.data(), sizeof(int) * N, hipMemcpyHostToDevice));
    timed("gather64c2", [&] { hipLaunchKernelGGL((k_gather<4, 4>), dim3(4096), dim3(256), 0, 0, rec, idx, N, out); }, N * 64.0, N * 64.0);
    printf("records %lld (%.0f MB of 128-byte records)\n", N, N * 128.0 / 1e6);
    return 0;
}
