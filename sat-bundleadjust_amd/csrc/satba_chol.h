// satba_chol.h -- dense SPD solve of the reduced camera system S dc = rhs on one GPU.
//
// S is (M n_p)^2 <= ~1200^2 float64, column-major, lower triangle valid (the Schur kernels only produce camera
// block pairs (a, b) with a <= b, which land in the column-major lower triangle).  This replaces the LSMR
// iteration of scipy:optimize/_lsq/trf.py:479-480 by an exact factorisation.
//
// More than 63 unknowns: ONE persistent launch factorises the matrix in 64 x 64 tiles and carries the right-hand side along
// (k_chol_tiles, satba_chol3.h), then the multi-workgroup backward substitution (k_trsv_back_mw).  Up to 63 unknowns (the
// reference's usual 2 .. 20 images x 3 parameters): the whole solve phase is one workgroup (k_solve_small, satba_chol3.h) and never
// comes here; n_c = (cameras) x (3, 5 or 6) is never 64.  The panel-step kernels of rounds 1 - 2 were removed in round 5.
// fp64 MFMA runs at the vector rate on gfx950 and the matrix is tiny: the solve is bound by the latency of the
// dependent pivot chain (tools/ubench/dp_latency.hip: 92 ns per column at best), not by flops; see DESIGN.md.
#pragma once
#include <hip/hip_runtime.h>

// gate of the device-resident LM loop (satba_kernels.h): a kernel returns at once when its gate word is 0; null: no gate
#ifndef SATBA_GATE
#define SATBA_GATE(g) do { if ((g) != nullptr && *(g) == 0) return; } while (0)
#endif

namespace satba {

constexpr int CH_NB = 32;

__device__ inline double readlane_f64(double v, int src_lane) {  // src_lane must be wave-uniform
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src_lane);
    hi = __builtin_amdgcn_readlane(hi, src_lane);
    return __hiloint2double(hi, lo);
}

// h = 0.5 / sqrt(d): v_rsq_f64 seed + two coupled Goldschmidt steps (six dependent operations; the library sqrt
// followed by a division is ~4x as many, and this sits on the serial chain of the diagonal-block factorisation 32
// times per panel).  d must be a normal positive number; sqrt(d) = 2 d h to rounding.
__device__ inline double half_rsqrt(double d) {
    const double y = __builtin_amdgcn_rsq(d);
    double g = d * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    return fma(h, r, h);
}

typedef double chol_d4 __attribute__((ext_vector_type(4)));

// L^T z = y in place in b (b holds y on entry), left-looking, one workgroup of 1024 threads:
// z_k = L_kk^-T (y_k - L[tail, k-block]^T z_tail).  The right-hand side lives in LDS; each wave takes two columns of
// the block, lanes run down the column (coalesced) with four independent partial sums in flight; the 32 x 32
// triangular solve is a readlane loop in wave 0.
__global__ __launch_bounds__(1024) void k_trsv_back(const double* __restrict__ L, int n, double* __restrict__ b) {
    extern __shared__ double yb[];  // n doubles
    __shared__ double t[CH_NB];
    __shared__ double Dk[CH_NB][CH_NB + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < n; i += 1024) yb[i] = b[i];
    __syncthreads();
    const int nblk = (n + CH_NB - 1) / CH_NB;
    for (int kb = nblk - 1; kb >= 0; --kb) {
        const int k0 = kb * CH_NB;
        const int nb = min(CH_NB, n - k0);
        const int tail0 = k0 + nb;
        // both columns of this wave at once, 8 rows per lane and column in flight (16 independent loads)
        {
            const int ca = wave * 2, cb = wave * 2 + 1;
            const double* cola = L + (size_t)(k0 + min(ca, nb - 1)) * n;
            const double* colb = L + (size_t)(k0 + min(cb, nb - 1)) * n;
            double sa = 0.0, sb = 0.0;
            for (int rb = tail0; rb < n; rb += 512) {
                double va[8], vb[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int r = rb + u * 64 + lane;
                    va[u] = (r < n) ? cola[r] : 0.0;
                    vb[u] = (r < n) ? colb[r] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int r = rb + u * 64 + lane;
                    const double y = (r < n) ? yb[r] : 0.0;
                    sa += va[u] * y;
                    sb += vb[u] * y;
                }
            }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) { sa += __shfl_xor(sa, d); sb += __shfl_xor(sb, d); }
            if (lane == 0 && ca < nb) t[ca] = yb[k0 + ca] - sa;
            if (lane == 0 && cb < nb) t[cb] = yb[k0 + cb] - sb;
        }
        // diagonal block into LDS (coalesced) while the dot products are in flight
        for (int idx = tid; idx < CH_NB * CH_NB; idx += 1024) {
            const int r = idx % CH_NB, c = idx / CH_NB;
            Dk[r][c] = (r < nb && c < nb && r >= c) ? L[(size_t)(k0 + r) + (size_t)(k0 + c) * n] : ((r == c) ? 1.0 : 0.0);
        }
        __syncthreads();
        if (wave == 0) {  // L_kk^T z = t: lane c, solved from the bottom up
            double v = (lane < nb) ? t[lane] : 0.0;
            const int cl = min(lane, CH_NB - 1);
            for (int j = nb - 1; j >= 0; --j) {
                const double zj = __shfl(v, j) / Dk[j][j];
                if (lane == j) v = zj;
                else if (lane < j) v -= Dk[j][cl] * zj;  // L[j][c]
            }
            if (lane < nb) yb[k0 + lane] = v;
        }
        __syncthreads();
    }
    for (int i = tid; i < n; i += 1024) b[i] = yb[i];
}

// L^T z = y for n <= 1024 by SEVERAL workgroups: workgroup k owns the 128 rows k0 = 128 k .. of the solution ("superblock").
// Bottom up, every workgroup subtracts L[J rows, k columns]^T z_J from its y_k for the superblocks J below it as their z_J
// appear (published in b behind flag[J]) and then solves its own 128 x 128 triangle with the four inverted 32 x 32 diagonal
// blocks.  One workgroup alone (round 2) paid the load latency of a 32 x 1000 operand block 32 times (2.6 us per
// step: 1024 threads x 128 registers hold exactly one step's operands): 82 us.  Here the operands of a workgroup's NEXT product
// are requested before it waits for the z they meet, every workgroup holds its diagonal triangle in registers from the start,
// and the chain is 8 hand-overs.  512 threads: thread (c, g) = column c of the superblock, rows 32 g .. 32 g + 31 of an operand
// block.  L^T is read from the strict upper triangle (written by k_chol_tiles): consecutive columns are consecutive addresses.
// flag: one int per superblock, zero on entry.  All workgroups are resident (at most 8 of them).
constexpr int CH_SB = 128;
// What follows the substitution in the solve phase (k_unscale), done by the last workgroup of k_trsv_back_mw when dc is set: the step in
// unscaled variables dc = z / scale_inv and the header of the phase (zero; slot 4 = lead x status word; the kept scalars) -- one
// launch less per iteration when the kernels run one after the other (4.8 us of 322 at 50 cameras).
struct TrsvTail {
    double* dc = nullptr;
    const double* scale_inv = nullptr;
    double* hdr = nullptr;
    int hdr_len = 0;
    const int* fail = nullptr;
    double lead = 1.0;
    const double* keep = nullptr;
    int keep_at = 0, keep_len = 0;
};
// done (or null): = done_epoch when z is complete -- a kernel on another stream may spin on it instead of waiting for an event (k_unscale)
__global__ __launch_bounds__(512) void k_trsv_back_mw(const double* __restrict__ L, const double* __restrict__ dinv, int n, double* __restrict__ b,
                                                      int* __restrict__ flag, const int* gate, int* __restrict__ done = nullptr, int done_epoch = 0,
                                                      TrsvTail tail = TrsvTail()) {
    SATBA_GATE(gate);
    __shared__ double ys[CH_SB], zs[CH_SB], part[4][CH_SB];
    __shared__ double Dk[4][CH_NB][CH_NB + 1];  // Dk[blk][r][c] = (D_blk^-1)[r][c] of my four diagonal blocks
    const int k = blockIdx.x, nsb = gridDim.x, k0 = k * CH_SB, tid = threadIdx.x;
    const int c = tid & (CH_SB - 1), g = tid >> 7, lane = tid & 63, wave = tid >> 6;
    const int col = min(k0 + c, n - 1);  // columns past the end of the matrix (last superblock) read a valid address, weight 0
    auto load_block = [&](double (&dst)[CH_NB], int row0, bool lower_only) {
        // dst[r] = L[row0 + 32 g + r][k0 + c] from the mirrored triangle; lower_only: my own triangle, rows in blocks below c's block
#pragma unroll
        for (int r = 0; r < CH_NB; ++r) {
            const int row = row0 + CH_NB * g + r;
            const bool ok = row < n && k0 + c < n && (!lower_only || g > (c >> 5));
            dst[r] = ok ? L[(size_t)col + (size_t)min(row, n - 1) * n] : 0.0;
        }
    };
    double Ld[CH_NB], La[CH_NB], Lb[CH_NB];
    load_block(Ld, k0, true);
    if (k + 1 < nsb) load_block(La, (nsb - 1) * CH_SB, false);  // operands of the first product
    if (tid < CH_SB) ys[tid] = (k0 + tid < n) ? b[k0 + tid] : 0.0;
    for (int idx = tid; idx < 4 * CH_NB * CH_NB; idx += 512) {
        const int blk = idx / (CH_NB * CH_NB), e = idx % (CH_NB * CH_NB), r = e / CH_NB, cc = e % CH_NB;
        const int kb = 4 * k + blk;
        Dk[blk][r][cc] = (kb * CH_NB < n) ? dinv[((size_t)kb * CH_NB + r) * CH_NB + cc] : ((r == cc) ? 1.0 : 0.0);
    }
    __syncthreads();
    // ---- the superblocks below me, bottom up; the operands of product J - 1 are requested before the wait for z_J
    auto product = [&](const double (&Lx)[CH_NB], int J) {
        if (wave == 0) {
            while (__hip_atomic_load(flag + J, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        if (tid < CH_SB) zs[tid] = (J * CH_SB + tid < n) ? __hip_atomic_load(b + J * CH_SB + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        __syncthreads();
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int r = 0; r < CH_NB; r += 2) { s0 = fma(Lx[r], zs[CH_NB * g + r], s0); s1 = fma(Lx[r + 1], zs[CH_NB * g + r + 1], s1); }
        part[g][c] = s0 + s1;
        __syncthreads();
        if (tid < CH_SB) ys[tid] -= part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
        // (the barrier in front of the next reader of ys follows below or in the next product)
    };
    for (int J = nsb - 1; J > k; J -= 2) {
        if (J - 1 > k) load_block(Lb, (J - 1) * CH_SB, false);
        product(La, J);
        if (J - 1 > k) {
            if (J - 2 > k) load_block(La, (J - 2) * CH_SB, false);
            product(Lb, J - 1);
        }
    }
    __syncthreads();
    // ---- my own triangle, 32-row blocks bottom up: z_blk = D_blk^-T y_blk (wave 0), then the columns in front of the block
    for (int blk = 3; blk >= 0; --blk) {
        if (wave == 0) {  // z_j = sum_r Dinv[r][j] y[32 blk + r]; lanes 32..63 take the second half of the sum
            const int j = lane & (CH_NB - 1), h = lane >> 5;
            double v = 0.0;
#pragma unroll
            for (int r = 0; r < CH_NB / 2; ++r) v = fma(Dk[blk][h * (CH_NB / 2) + r][j], ys[CH_NB * blk + h * (CH_NB / 2) + r], v);
            v += __shfl_xor(v, 32);
            __builtin_amdgcn_wave_barrier();  // every lane has read y_blk
            if (lane < CH_NB) { zs[CH_NB * blk + lane] = v; ys[CH_NB * blk + lane] = v; }
        }
        __syncthreads();
        if (blk > 0 && g == blk && c < CH_NB * blk) {  // rows of block blk are exactly the rows thread (c, g = blk) holds
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int r = 0; r < CH_NB; r += 2) { s0 = fma(Ld[r], zs[CH_NB * blk + r], s0); s1 = fma(Ld[r + 1], zs[CH_NB * blk + r + 1], s1); }
            ys[c] -= s0 + s1;
        }
        __syncthreads();
    }
    // ---- publish z_k (it is also the result): write-through stores, then the flag
    if (tid < CH_SB && k0 + tid < n) __hip_atomic_store(b + k0 + tid, ys[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_store(flag + k, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (k == 0 && done) __hip_atomic_store(done, done_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (superblock 0 is the last one)
    }
    if (k == 0 && tail.dc) {  // every z is published (the other superblocks' behind their flags, this one's drained above)
        for (int i = tid; i < n; i += 512) tail.dc[i] = __hip_atomic_load(b + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / tail.scale_inv[i];
        for (int i = tid; i < tail.hdr_len; i += 512) {
            double v = 0.0;
            if (i == 4) v = tail.lead * (double)*tail.fail;
            if (i >= tail.keep_at && i < tail.keep_at + tail.keep_len) v = tail.lead * tail.keep[i - tail.keep_at];
            tail.hdr[i] = v;
        }
    }
}

}  // namespace satba
#include "satba_chol3.h"
namespace satba {

constexpr int CH_MAX_STEPS = 256;  // flag words behind the not-SPD flag (the backward substitution's sit at CH_TRSV_FLAGS)
constexpr int CH_TRSV_FLAGS = 64;  // k_trsv_back_mw's flags

inline void cholesky_init() { chol_tiles_init(); }

// Scratch of the tile kernel (per handle): tile flags (zeroed once: they carry epochs), the inverted 64 x 64 diagonal blocks,
// the tiles' shares of the forward substitution, the ticket counters (zero between launches).
struct CholWork {
    int* flags = nullptr;
    double* Linv = nullptr;
    double* Cc = nullptr;
    int* ctr = nullptr;
    int epoch = 0;
};
inline hipError_t chol_work_alloc(CholWork& w, int n) {  // (tools; the library allocates through its handle)
    const size_t T = (size_t)(n + 63) / 64;
    hipError_t e;
    if ((e = hipMalloc((void**)&w.flags, sizeof(int) * (T * T + 1))) != hipSuccess) return e;
    if ((e = hipMalloc((void**)&w.Linv, sizeof(double) * (T * 4096 + 1))) != hipSuccess) return e;
    if ((e = hipMalloc((void**)&w.Cc, sizeof(double) * (T * T * 64 + 1))) != hipSuccess) return e;
    if ((e = hipMalloc((void**)&w.ctr, sizeof(int) * 4)) != hipSuccess) return e;
    if ((e = hipMemset(w.flags, 0, sizeof(int) * (T * T + 1))) != hipSuccess) return e;
    if ((e = hipMemset(w.ctr, 0, sizeof(int) * 4)) != hipSuccess) return e;
    w.epoch = 0;
    return hipSuccess;
}
inline void chol_work_free(CholWork& w) {
    (void)hipFree(w.flags); (void)hipFree(w.Linv); (void)hipFree(w.Cc); (void)hipFree(w.ctr);
    w = CholWork();
}

// the tile kernel alone: factor in place, y = L^-1 b, L^T in the strict upper triangle and the 32 x 32 block inverses if asked
inline void cholesky_tiles(double* A, int n, double* b, int* fail, hipStream_t stream, CholWork& w, double* dinv, bool mirror, const int* gate,
                           long long* ts = nullptr) {
    C3Args g;
    g.A = A; g.n = n; g.b = b; g.fail = fail; g.flags = w.flags; g.epoch = ++w.epoch; g.Linv = w.Linv; g.Cc = w.Cc; g.ctr = w.ctr;
    g.dinv = dinv; g.ts = ts; g.mirror = mirror ? 1 : 0;
    hipLaunchKernelGGL(k_chol_tiles, dim3(chol_tiles_grid(n, g.mirror)), dim3(1024), c3_lds_bytes(), stream, g, gate);
}

// Factorise A (n x n, column-major lower, in place) and solve A z = b in place.  *fail != 0 if A was not SPD (or a wait inside the
// tile kernel timed out: bit 1).  flags: CH_MAX_STEPS ints of scratch directly behind *fail (flags == fail + 1); both are cleared here
// unless `cleared` (satba_solve does it in its scaling kernel).  dinv: (n / 32 rounded up) x 1024 doubles of scratch.
// ts (tools): C3_TS time stamps per step of the tile kernel (a build with -DC3_STAMPS).
// tail (or null): what k_unscale does, by the backward substitution's last workgroup where that kernel runs (returns true), else left to the caller
inline bool cholesky_solve(double* A, int n, double* b, int* fail, int* flags, hipStream_t stream, CholWork& w, double* dinv, bool cleared = false,
                           const int* gate = nullptr, long long* ts = nullptr, const TrsvTail* tail = nullptr) {
    cholesky_init();
    if (!cleared) (void)hipMemsetAsync(fail, 0, sizeof(int) * (1 + CH_MAX_STEPS), stream);  // flags == fail + 1: one fill for both
    const bool mw = n <= 1024;  // the multi-workgroup backward substitution reads L^T from the upper triangle and the 32 x 32 inverses
    cholesky_tiles(A, n, b, fail, stream, w, mw ? dinv : nullptr, mw, gate, ts);
    if (mw) {
        hipLaunchKernelGGL(k_trsv_back_mw, dim3((n + CH_SB - 1) / CH_SB), dim3(512), 0, stream, A, dinv, n, b, flags + CH_TRSV_FLAGS, gate, (int*)nullptr, 0,
                           tail ? *tail : TrsvTail());
        return tail != nullptr;
    }
    hipLaunchKernelGGL(k_trsv_back, dim3(1), dim3(1024), sizeof(double) * n, stream, A, n, b);
    return false;
}

}  // namespace satba
