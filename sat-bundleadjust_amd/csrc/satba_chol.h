// satba_chol.h -- dense SPD solve of the reduced camera system S dc = rhs on one GPU.
//
// S is (M n_p)^2 <= ~1200^2 float64, column-major, lower triangle valid (the Schur kernel only writes camera-block
// pairs (a, b) with a <= b, which land in the column-major lower triangle).  This replaces the LSMR iteration of
// scipy:optimize/_lsq/trf.py:479-480 by an exact factorisation.  Blocked right-looking Cholesky:
//   per 32-column panel: k_potrf_trsm (diagonal block factorised in LDS by every workgroup, panel rows solved
//   64 per workgroup) then k_syrk (64x64 tiles of the trailing matrix, 4x4 per thread, panel staged in LDS);
//   then a single-workgroup blocked forward / backward substitution.
// The matrix is too small for MFMA to matter (fp64 MFMA runs at the vector rate on gfx950) and the whole solve
// is launch/latency bound; see DESIGN.md.
#pragma once
#include <hip/hip_runtime.h>

namespace satba {

constexpr int CH_NB = 32;

__global__ __launch_bounds__(64) void k_potrf_trsm(double* __restrict__ A, int n, int k0, int* __restrict__ fail) {
    __shared__ double D[CH_NB][CH_NB + 1];
    const int tid = threadIdx.x;
    const int nb = min(CH_NB, n - k0);
    for (int idx = tid; idx < nb * nb; idx += 64) {
        const int r = idx % nb, c = idx / nb;
        D[r][c] = (r >= c) ? A[(size_t)(k0 + r) + (size_t)(k0 + c) * n] : 0.0;
    }
    __syncthreads();
    for (int j = 0; j < nb; ++j) {
        if (tid == j) {
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
        if (tid > j && tid < nb) {
            const double l = D[tid][j];
            for (int c = j + 1; c <= tid; ++c) D[tid][c] -= l * D[c][j];
        }
        __syncthreads();
    }
    if (blockIdx.x == 0) {
        for (int idx = tid; idx < nb * nb; idx += 64) {
            const int r = idx % nb, c = idx / nb;
            if (r >= c) A[(size_t)(k0 + r) + (size_t)(k0 + c) * n] = D[r][c];
        }
        return;
    }
    const int r = k0 + nb + (blockIdx.x - 1) * 64 + tid;
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

// trailing update A[base:, base:] -= P P^T (lower tiles only), P = A[base:, k0:k0+nb], base = k0 + nb
__global__ __launch_bounds__(256) void k_syrk(double* __restrict__ A, int n, int k0, int nb) {
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
    double acc[4][4] = {};
    const int ty = threadIdx.y * 4, tx = threadIdx.x * 4;
    for (int k = 0; k < nb; ++k) {
        double a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = Pi[k][ty + i]; b[i] = Pj[k][tx + i]; }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + ty + i, c = c0 + tx + j;
            if (r < n && c < n && r >= c) A[(size_t)r + (size_t)c * n] -= acc[i][j];
        }
}

// L y = b, then L^T z = y, in place in b (length n).  One workgroup of 256 threads.
__global__ __launch_bounds__(256) void k_trsv2(const double* __restrict__ L, int n, double* __restrict__ b) {
    __shared__ double D[CH_NB][CH_NB + 1];
    __shared__ double y[CH_NB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ---- forward
    for (int k0 = 0; k0 < n; k0 += CH_NB) {
        const int nb = min(CH_NB, n - k0);
        for (int idx = tid; idx < nb * nb; idx += 256) {
            const int r = idx % nb, c = idx / nb;
            D[r][c] = L[(size_t)(k0 + r) + (size_t)(k0 + c) * n];
        }
        __syncthreads();
        if (wave == 0) {
            double v = (lane < nb) ? b[k0 + lane] : 0.0;
            for (int j = 0; j < nb; ++j) {
                const double yj = __shfl(v, j) / D[j][j];
                if (lane == j) v = yj;
                else if (lane > j && lane < nb) v -= D[lane][j] * yj;
            }
            if (lane < nb) { y[lane] = v; b[k0 + lane] = v; }
        }
        __syncthreads();
        for (int r = k0 + nb + tid; r < n; r += 256) {
            double s = 0.0;
            for (int c = 0; c < nb; ++c) s += L[(size_t)r + (size_t)(k0 + c) * n] * y[c];
            b[r] -= s;
        }
        __syncthreads();
    }
    // ---- backward: z[k] = (y[k] - sum_{r>k} L[r][k] z[r]) / L[k][k]
    const int nblk = (n + CH_NB - 1) / CH_NB;
    for (int kb = nblk - 1; kb >= 0; --kb) {
        const int k0 = kb * CH_NB;
        const int nb = min(CH_NB, n - k0);
        // contributions of the already solved tail, one column per wave at a time
        for (int c = wave; c < nb; c += 4) {
            double s = 0.0;
            const double* col = L + (size_t)(k0 + c) * n;
            for (int r = k0 + nb + lane; r < n; r += 64) s += col[r] * b[r];
            for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d);
            if (lane == 0) y[c] = b[k0 + c] - s;
        }
        for (int idx = tid; idx < nb * nb; idx += 256) {
            const int r = idx % nb, c = idx / nb;
            D[r][c] = L[(size_t)(k0 + r) + (size_t)(k0 + c) * n];
        }
        __syncthreads();
        if (wave == 0) {
            double v = (lane < nb) ? y[lane] : 0.0;
            for (int j = nb - 1; j >= 0; --j) {
                const double zj = __shfl(v, j) / D[j][j];
                if (lane == j) v = zj;
                else if (lane < j) v -= D[j][lane] * zj;
            }
            if (lane < nb) b[k0 + lane] = v;
        }
        __syncthreads();
    }
}

// Factorise A (n x n, column-major lower, in place) and solve A z = b in place.  *fail != 0 if A was not SPD.
inline void cholesky_solve(double* A, int n, double* b, int* fail, hipStream_t stream) {
    for (int k0 = 0; k0 < n; k0 += CH_NB) {
        const int nb = n - k0 < CH_NB ? n - k0 : CH_NB;
        const int rest = n - k0 - nb;
        const int row_blocks = (rest + 63) / 64;
        hipLaunchKernelGGL(k_potrf_trsm, dim3(1 + row_blocks), dim3(64), 0, stream, A, n, k0, fail);
        if (rest > 0) hipLaunchKernelGGL(k_syrk, dim3(row_blocks, row_blocks), dim3(16, 16), 0, stream, A, n, k0, nb);
    }
    hipLaunchKernelGGL(k_trsv2, dim3(1), dim3(256), 0, stream, A, n, b);
}

}  // namespace satba
