// satba_chol.h -- dense SPD solve of the reduced camera system S dc = rhs on one GPU.
//
// S is (M n_p)^2 <= ~1200^2 float64, column-major, lower triangle valid (the Schur kernels only produce camera
// block pairs (a, b) with a <= b, which land in the column-major lower triangle).  This replaces the LSMR
// iteration of scipy:optimize/_lsq/trf.py:479-480 by an exact factorisation.
//
// Blocked right-looking Cholesky, 32-column panels, two launches per panel:
//   k_potrf_trsm  every workgroup factorises the 32x32 diagonal block in LDS (redundantly -- cheaper than a
//                 separate launch), workgroup 0 writes it back together with its explicit inverse and advances
//                 the forward substitution of the right-hand side, the others solve 256 panel rows each;
//   k_syrk        64x64 tiles of the trailing matrix, 4x4 per thread, panel staged in LDS, rows mapped to the
//                 fast thread index so the read-modify-write of A is coalesced; tiles of the first tile column
//                 also apply the panel to the right-hand side (forward substitution is thereby folded into the
//                 factorisation: no separate L y = b pass);
// then k_trsv_back: one workgroup, right-looking backward substitution that uses the stored inverses of the
// diagonal blocks (a 32x32 mat-vec instead of a 32-step dependent chain).
// fp64 MFMA runs at the vector rate on gfx950 and the matrix is tiny: the solve is launch / latency bound
// (64 dependent launches), not flop bound; see DESIGN.md.
#pragma once
#include <hip/hip_runtime.h>

namespace satba {

constexpr int CH_NB = 32;
constexpr int CH_ROWS = 256;  // panel rows per workgroup in k_potrf_trsm
constexpr int CH_THREADS = 1024;

// dinv: (n / 32 + 1) blocks of 32 x 32 (row-major) receiving inv(L_kk).
// Workgroup 0: factorise the diagonal block, publish L_kk and inv(L_kk), advance the right-hand side.
// Workgroups 1..: same factorisation (redundant), then 64 panel rows each as a small GEMM  X = P inv(L_kk)^T
// (twice the flops of a triangular solve, but no dependent chain and no 32-deep unrolled register array).
__global__ __launch_bounds__(CH_THREADS) void k_potrf_trsm(double* __restrict__ A, int n, int k0, int* __restrict__ fail,
                                                    double* __restrict__ b, double* __restrict__ dinv) {
    __shared__ double D[CH_NB][CH_NB + 1];
    __shared__ double Di[CH_NB][CH_NB + 1];
    __shared__ double Pt[CH_NB][CH_ROWS];
    const int tid = threadIdx.x;
    const int nb = min(CH_NB, n - k0);
    for (int idx = tid; idx < CH_NB * CH_NB; idx += CH_THREADS) {
        const int r = idx % CH_NB, c = idx / CH_NB;
        D[r][c] = (r < nb && c < nb && r >= c) ? A[(size_t)(k0 + r) + (size_t)(k0 + c) * n] : ((r == c) ? 1.0 : 0.0);
    }
    const int r0 = k0 + nb + ((int)blockIdx.x - 1) * CH_ROWS;
    if (blockIdx.x > 0) {  // stage this workgroup's panel rows (transposed) while the factorisation runs
        for (int idx = tid; idx < CH_NB * CH_ROWS; idx += CH_THREADS) {
            const int r = idx % CH_ROWS, k = idx / CH_ROWS;
            Pt[k][r] = (k < nb && r0 + r < n) ? A[(size_t)(r0 + r) + (size_t)(k0 + k) * n] : 0.0;
        }
    }
    __syncthreads();
    // unblocked Cholesky of the 32 x 32 block, all threads on the rank-1 updates
    for (int j = 0; j < nb; ++j) {
        if (tid == 0) {
            double d = D[j][j];
            if (!(d > 0.0)) {  // not positive definite (or NaN): flag it, keep going with a harmless pivot
                if (blockIdx.x == 0) atomicOr(fail, 1);
                d = 1.0;
            }
            D[j][j] = sqrt(d);
        }
        __syncthreads();
        if (tid > j && tid < nb) D[tid][j] /= D[j][j];
        __syncthreads();
        for (int idx = tid; idx < CH_NB * CH_NB; idx += CH_THREADS) {
            const int r = idx % CH_NB, c = idx / CH_NB;
            if (c > j && r >= c && r < nb) D[r][c] -= D[r][j] * D[c][j];
        }
        __syncthreads();
    }
    // inv(L_kk): half-wave c owns column c, lane m of it holds Di[m][c] in a register; row r of the column is a
    // 32-lane dot product (shuffles), so the whole inverse is 32 short steps with no LDS round trips
    {
        const int c = tid >> 5, m = tid & 31;
        double mine = 0.0;  // Di[m][c]
        for (int r = 0; r < CH_NB; ++r) {
            double v = (m < r && m >= c) ? D[r][m] * mine : 0.0;
#pragma unroll
            for (int d = 16; d > 0; d >>= 1) v += __shfl_xor(v, d);
            if (m == r) mine = (r >= c) ? (((r == c) ? 1.0 : 0.0) - v) / D[r][r] : 0.0;
        }
        Di[m][c] = mine;
    }
    __syncthreads();
    if (blockIdx.x == 0) {
        for (int idx = tid; idx < nb * nb; idx += CH_THREADS) {
            const int r = idx % nb, c = idx / nb;
            if (r >= c) A[(size_t)(k0 + r) + (size_t)(k0 + c) * n] = D[r][c];
        }
        double* out = dinv + (size_t)(k0 / CH_NB) * CH_NB * CH_NB;
        for (int idx = tid; idx < CH_NB * CH_NB; idx += CH_THREADS) out[idx] = Di[idx / CH_NB][idx % CH_NB];
        // forward substitution of this block of the right-hand side: y_k = inv(L_kk) b_k
        double s = 0.0;
        if (tid < nb)
            for (int m = 0; m <= tid; ++m) s += Di[tid][m] * b[k0 + m];
        __syncthreads();
        if (tid < nb) b[k0 + tid] = s;
        return;
    }
    // X[r][c] = sum_k P[r][k] inv(L)[c][k]; thread = 4 rows x 2 columns
    const int tr = (tid & 63) * 4, tc = (tid >> 6) * 2;
    double acc[2][4] = {};
#pragma unroll 8
    for (int k = 0; k < CH_NB; ++k) {
        const double d0 = Di[tc][k], d1 = Di[tc + 1][k];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double a = Pt[k][tr + i];
            acc[0][i] += a * d0;
            acc[1][i] += a * d1;
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + tr + i, c = tc + j;
            if (r < n && c < nb) A[(size_t)r + (size_t)(k0 + c) * n] = acc[j][i];
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
// z_k = inv(L_kk)^T (y_k - L[tail, k-block]^T z_tail).
__global__ __launch_bounds__(1024) void k_trsv_back(const double* __restrict__ L, int n, const double* __restrict__ dinv,
                                                    double* __restrict__ b) {
    extern __shared__ double yb[];  // n doubles: the right-hand side / solution lives in LDS for the whole solve
    __shared__ double t[CH_NB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < n; i += 1024) yb[i] = b[i];
    __syncthreads();
    const int nblk = (n + CH_NB - 1) / CH_NB;
    for (int kb = nblk - 1; kb >= 0; --kb) {
        const int k0 = kb * CH_NB;
        const int nb = min(CH_NB, n - k0);
        const int tail0 = k0 + nb;
        // t_c = y_c - sum_{r >= tail0} L[r][k0 + c] z[r]: each wave takes two columns, lanes run down the column
        // (coalesced), the loads of a column are independent of each other and of the LDS traffic
        for (int cc = 0; cc < 2; ++cc) {
            const int c = wave * 2 + cc;
            double s = 0.0;
            if (c < nb) {
                const double* col = L + (size_t)(k0 + c) * n;
                for (int r = tail0 + lane; r < n; r += 64) s += col[r] * yb[r];
            }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
            if (lane == 0 && c < nb) t[c] = yb[k0 + c] - s;
        }
        __syncthreads();
        // z_k = inv(L_kk)^T t
        if (tid < nb) {
            const double* Di = dinv + (size_t)kb * CH_NB * CH_NB;  // row-major inv(L_kk)
            double s = 0.0;
            for (int m = tid; m < nb; ++m) s += Di[m * CH_NB + tid] * t[m];
            yb[k0 + tid] = s;
        }
        __syncthreads();
    }
    for (int i = tid; i < n; i += 1024) b[i] = yb[i];
}

inline size_t cholesky_workspace_doubles(int n) { return (size_t)(n / CH_NB + 1) * CH_NB * CH_NB; }

// Factorise A (n x n, column-major lower, in place) and solve A z = b in place.  *fail != 0 if A was not SPD.
// dinv: workspace of cholesky_workspace_doubles(n) doubles.
inline void cholesky_solve(double* A, int n, double* b, int* fail, double* dinv, hipStream_t stream) {
    for (int k0 = 0; k0 < n; k0 += CH_NB) {
        const int nb = n - k0 < CH_NB ? n - k0 : CH_NB;
        const int rest = n - k0 - nb;
        hipLaunchKernelGGL(k_potrf_trsm, dim3(1 + (rest + CH_ROWS - 1) / CH_ROWS), dim3(CH_THREADS), 0, stream, A, n, k0, fail, b, dinv);
        if (rest > 0) {
            const int tiles = (rest + 63) / 64;
            hipLaunchKernelGGL(k_syrk, dim3(tiles, tiles), dim3(16, 16), 0, stream, A, n, k0, nb, b);
        }
    }
    hipLaunchKernelGGL(k_trsv_back, dim3(1), dim3(1024), sizeof(double) * n, stream, A, n, dinv, b);
}

}  // namespace satba
