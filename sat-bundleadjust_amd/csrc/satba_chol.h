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

// dinv: (n / 32 + 1) blocks of 32 x 32 (row-major) receiving inv(L_kk)
__global__ __launch_bounds__(CH_ROWS) void k_potrf_trsm(double* __restrict__ A, int n, int k0, int* __restrict__ fail,
                                                        double* __restrict__ b, double* __restrict__ dinv) {
    __shared__ double D[CH_NB][CH_NB + 1];
    __shared__ double Di[CH_NB][CH_NB + 1];
    const int tid = threadIdx.x;
    const int nb = min(CH_NB, n - k0);
    for (int idx = tid; idx < CH_NB * CH_NB; idx += CH_ROWS) {
        const int r = idx % CH_NB, c = idx / CH_NB;
        D[r][c] = (r < nb && c < nb && r >= c) ? A[(size_t)(k0 + r) + (size_t)(k0 + c) * n] : ((r == c) ? 1.0 : 0.0);
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
        for (int idx = tid; idx < CH_NB * CH_NB; idx += CH_ROWS) {
            const int r = idx % CH_NB, c = idx / CH_NB;
            if (c > j && r >= c && r < nb) D[r][c] -= D[r][j] * D[c][j];
        }
        __syncthreads();
    }
    if (blockIdx.x == 0) {
        for (int idx = tid; idx < nb * nb; idx += CH_ROWS) {
            const int r = idx % nb, c = idx / nb;
            if (r >= c) A[(size_t)(k0 + r) + (size_t)(k0 + c) * n] = D[r][c];
        }
        // inverse of the lower-triangular block: thread c solves L x = e_c
        if (tid < CH_NB) {
            const int c = tid;
            for (int r = 0; r < CH_NB; ++r) {
                double s = (r == c) ? 1.0 : 0.0;
                for (int m = c; m < r; ++m) s -= D[r][m] * Di[m][c];
                Di[r][c] = (r >= c) ? s / D[r][r] : 0.0;
            }
        }
        __syncthreads();
        double* out = dinv + (size_t)(k0 / CH_NB) * CH_NB * CH_NB;
        for (int idx = tid; idx < CH_NB * CH_NB; idx += CH_ROWS) out[idx] = Di[idx / CH_NB][idx % CH_NB];
        // forward substitution of this block of the right-hand side: y_k = inv(L_kk) b_k
        double s = 0.0;
        if (tid < nb)
            for (int m = 0; m <= tid; ++m) s += Di[tid][m] * b[k0 + m];
        __syncthreads();
        if (tid < nb) b[k0 + tid] = s;
        return;
    }
    const int r = k0 + nb + (blockIdx.x - 1) * CH_ROWS + tid;
    if (r >= n) return;
    double x[CH_NB];
#pragma unroll
    for (int c = 0; c < CH_NB; ++c) x[c] = (c < nb) ? A[(size_t)r + (size_t)(k0 + c) * n] : 0.0;
#pragma unroll
    for (int c = 0; c < CH_NB; ++c) {
        if (c < nb) {
            double s = x[c];
#pragma unroll
            for (int m = 0; m < c; ++m) s -= x[m] * D[c][m];
            x[c] = s / D[c][c];
        }
    }
#pragma unroll
    for (int c = 0; c < CH_NB; ++c)
        if (c < nb) A[(size_t)r + (size_t)(k0 + c) * n] = x[c];
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

// L^T z = y in place in b (b holds y on entry), right-looking: z_k = inv(L_kk)^T y_k, then every earlier entry
// gets its contribution y_j -= L[k-block, j]^T z_k.  One workgroup of 1024 threads.
__global__ __launch_bounds__(1024) void k_trsv_back(const double* __restrict__ L, int n, const double* __restrict__ dinv,
                                                    double* __restrict__ b) {
    __shared__ double z[CH_NB];
    const int tid = threadIdx.x;
    const int nblk = (n + CH_NB - 1) / CH_NB;
    for (int kb = nblk - 1; kb >= 0; --kb) {
        const int k0 = kb * CH_NB;
        const int nb = min(CH_NB, n - k0);
        if (tid < CH_NB) {
            double s = 0.0;
            if (tid < nb) {
                const double* Di = dinv + (size_t)kb * CH_NB * CH_NB;  // row-major inv(L_kk); need its transpose
                for (int m = tid; m < nb; ++m) s += Di[m * CH_NB + tid] * b[k0 + m];
            }
            z[tid] = s;
        }
        __syncthreads();
        if (tid < nb) b[k0 + tid] = z[tid];
        for (int j = tid; j < k0; j += 1024) {
            const double* col = L + (size_t)j * n + k0;  // rows k0 .. k0+nb of column j: contiguous
            double s = 0.0;
            for (int r = 0; r < nb; ++r) s += col[r] * z[r];
            b[j] -= s;
        }
        __syncthreads();
    }
}

inline size_t cholesky_workspace_doubles(int n) { return (size_t)(n / CH_NB + 1) * CH_NB * CH_NB; }

// Factorise A (n x n, column-major lower, in place) and solve A z = b in place.  *fail != 0 if A was not SPD.
// dinv: workspace of cholesky_workspace_doubles(n) doubles.
inline void cholesky_solve(double* A, int n, double* b, int* fail, double* dinv, hipStream_t stream) {
    for (int k0 = 0; k0 < n; k0 += CH_NB) {
        const int nb = n - k0 < CH_NB ? n - k0 : CH_NB;
        const int rest = n - k0 - nb;
        hipLaunchKernelGGL(k_potrf_trsm, dim3(1 + (rest + CH_ROWS - 1) / CH_ROWS), dim3(CH_ROWS), 0, stream, A, n, k0, fail, b, dinv);
        if (rest > 0) {
            const int tiles = (rest + 63) / 64;
            hipLaunchKernelGGL(k_syrk, dim3(tiles, tiles), dim3(16, 16), 0, stream, A, n, k0, nb, b);
        }
    }
    hipLaunchKernelGGL(k_trsv_back, dim3(1), dim3(1024), 0, stream, A, n, dinv, b);
}

}  // namespace satba
