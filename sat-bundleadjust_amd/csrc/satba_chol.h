// satba_chol.h -- dense SPD solve of the reduced camera system S dc = rhs on one GPU.
//
// S is (M n_p)^2 <= ~1200^2 float64, column-major, lower triangle valid (the Schur kernels only produce camera
// block pairs (a, b) with a <= b, which land in the column-major lower triangle).  This replaces the LSMR
// iteration of scipy:optimize/_lsq/trf.py:479-480 by an exact factorisation.
//
// Blocked right-looking Cholesky, 32-column panels, two launches per panel:
//   k_potrf_trsm  register-resident: every wave factorises the 32x32 diagonal block redundantly (lane = row,
//                 column broadcasts by v_readlane, no LDS, no barriers) and solves 64 panel rows against it; the
//                 right-hand side rides along as one more panel row, so the forward substitution is folded in;
//   k_syrk        64x64 tiles of the trailing matrix, 4x4 per thread, panel staged in LDS, rows mapped to the
//                 fast thread index so the read-modify-write of A is coalesced; tiles of the first tile column
//                 also apply the panel to the right-hand side;
// then k_trsv_back: one workgroup, left-looking backward substitution with the right-hand side in LDS.
// fp64 MFMA runs at the vector rate on gfx950 and the matrix is tiny: the solve is bound by the latency of 64
// dependent launches (~27 us + ~13 us per panel at n = 1000), not by flops; see DESIGN.md.
#pragma once
#include <hip/hip_runtime.h>

namespace satba {

constexpr int CH_NB = 32;

__device__ inline double readlane_f64(double v, int src_lane) {  // src_lane must be wave-uniform
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src_lane);
    hi = __builtin_amdgcn_readlane(hi, src_lane);
    return __hiloint2double(hi, lo);
}

// Panel step, entirely in registers, no LDS and no barriers.  Every wave first factorises the 32 x 32 diagonal
// block redundantly -- lane r (< 32) holds row r in 32 registers, column values are broadcast with v_readlane --
// and then solves 64 panel rows (one per lane, also in registers) against it, again through v_readlane
// broadcasts of L.  The right-hand side is treated as one more panel row (forward substitution folded in).
// Wave 0 of workgroup 0 writes the factorised diagonal block back.
// rows handled: base .. n-1 (base = k0 + nb) and the virtual row n = right-hand side b.
__global__ __launch_bounds__(256) void k_potrf_trsm(double* __restrict__ A, int n, int k0, int* __restrict__ fail,
                                                    double* __restrict__ b) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nb = min(CH_NB, n - k0);
    const int base = k0 + nb;
    const int gw = blockIdx.x * 4 + wave;  // global wave index
    if (gw * 64 > n - base) return;        // this wave's first row is beyond the virtual row n

    // ---- diagonal block into registers (identity padding beyond nb)
    double a[CH_NB];
    const int dr = min(lane, CH_NB - 1);
#pragma unroll
    for (int c = 0; c < CH_NB; ++c) {
        double v = (c == dr) ? 1.0 : 0.0;
        if (lane < nb && c < nb && c <= lane) v = A[(size_t)(k0 + lane) + (size_t)(k0 + c) * n];
        a[c] = v;
    }
    // ---- Cholesky of the block: after step j, a[j] of lane r >= j holds L[r][j]
    double my_inv = 1.0;  // 1 / L[lane][lane]
    bool bad = false;
#pragma unroll
    for (int j = 0; j < CH_NB; ++j) {
        double d = readlane_f64(a[j], j);
        if (!(d > 0.0)) { bad = true; d = 1.0; }
        const double sq = sqrt(d), inv = 1.0 / sq;
        const double l = (lane == j) ? sq : a[j] * inv;
        a[j] = l;
        if (lane == j) my_inv = inv;
#pragma unroll
        for (int c = j + 1; c < CH_NB; ++c) a[c] -= l * readlane_f64(l, c);
    }
    if (gw == 0) {
        if (bad && lane == 0) atomicOr(fail, 1);
        if (lane < nb) {
#pragma unroll
            for (int c = 0; c < CH_NB; ++c)
                if (c <= lane && c < nb) A[(size_t)(k0 + lane) + (size_t)(k0 + c) * n] = a[c];
        }
    }
    // ---- this lane's panel row (or the right-hand side): x L_kk^T = p
    const int r = base + gw * 64 + lane;
    const bool is_rhs = (r == n), valid = (r <= n);
    double x[CH_NB];
#pragma unroll
    for (int c = 0; c < CH_NB; ++c) {
        double v = 0.0;
        if (c < nb && valid) v = is_rhs ? b[k0 + c] : A[(size_t)r + (size_t)(k0 + c) * n];
        x[c] = v;
    }
#pragma unroll
    for (int m = 0; m < CH_NB; ++m) {
        const double xm = x[m] * readlane_f64(my_inv, m);
        x[m] = xm;
#pragma unroll
        for (int c = m + 1; c < CH_NB; ++c) x[c] -= xm * readlane_f64(a[m], c);  // L[c][m]
    }
    if (valid) {
#pragma unroll
        for (int c = 0; c < CH_NB; ++c) {
            if (c < nb) {
                if (is_rhs) b[k0 + c] = x[c];
                else A[(size_t)r + (size_t)(k0 + c) * n] = x[c];
            }
        }
    }
}

// trailing update A[base:, base:] -= P P^T (lower tiles only), P = A[base:, k0:k0+nb], base = k0 + nb;
// the first tile column also applies  b[base:] -= P y_k  with y_k = b[k0:k0+nb] (already final).
__global__ __launch_bounds__(256) void k_syrk(double* __restrict__ A, int n, int k0, int nb, double* __restrict__ b) {
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bj > bi) return;
    __shared__ double Pi[CH_NB][64], Pj[CH_NB][64];
    const int base = k0 + nb;
    const int r0 = base + bi * 64, c0 = base + bj * 64;
    const int tid = threadIdx.y * 16 + threadIdx.x;
    for (int idx = tid; idx < nb * 64; idx += 256) {
        const int r = idx & 63, k = idx >> 6;
        Pi[k][r] = (r0 + r < n) ? A[(size_t)(r0 + r) + (size_t)(k0 + k) * n] : 0.0;
        Pj[k][r] = (c0 + r < n) ? A[(size_t)(c0 + r) + (size_t)(k0 + k) * n] : 0.0;
    }
    __syncthreads();
    if (bj == 0 && tid < 64 && r0 + tid < n) {
        double s = 0.0;
        for (int k = 0; k < nb; ++k) s += Pi[k][tid] * b[k0 + k];
        b[r0 + tid] -= s;
    }
    double acc[4][4] = {};
    const int tr = threadIdx.x * 4, tc = threadIdx.y * 4;  // rows on the fast index: coalesced A accesses
#pragma unroll 8
    for (int k = 0; k < nb; ++k) {
        double a[4], c[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = Pi[k][tr + i]; c[i] = Pj[k][tc + i]; }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] += a[i] * c[j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + tr + i, c = c0 + tc + j;
            if (r < n && c < n && r >= c) A[(size_t)r + (size_t)c * n] -= acc[j][i];
        }
}

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

inline size_t cholesky_workspace_doubles(int n) { (void)n; return 1; }

// Factorise A (n x n, column-major lower, in place) and solve A z = b in place.  *fail != 0 if A was not SPD.
inline void cholesky_solve(double* A, int n, double* b, int* fail, double* /*unused*/, hipStream_t stream) {
    for (int k0 = 0; k0 < n; k0 += CH_NB) {
        const int nb = n - k0 < CH_NB ? n - k0 : CH_NB;
        const int rest = n - k0 - nb;
        const int waves = (rest + 1 + 63) / 64;  // panel rows + the right-hand side row
        hipLaunchKernelGGL(k_potrf_trsm, dim3((waves + 3) / 4), dim3(256), 0, stream, A, n, k0, fail, b);
        if (rest > 0) {
            const int tiles = (rest + 63) / 64;
            hipLaunchKernelGGL(k_syrk, dim3(tiles, tiles), dim3(16, 16), 0, stream, A, n, k0, nb, b);
        }
    }
    hipLaunchKernelGGL(k_trsv_back, dim3(1), dim3(1024), sizeof(double) * n, stream, A, n, b);
}

}  // namespace satba
