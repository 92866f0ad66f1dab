// satba_chol2.h -- double panel step of the dense Cholesky (included by satba_chol.h).
//
// k_chol_step spends ~20 us per 32-column panel on a chain that cannot be shortened by more parallelism: launch gap
// (3 us), trailing update of the tile column that holds the panel (7 us, 3 of them the first loads after the launch),
// diagonal block (6.5 us), panel solve behind a flag (3.4 us).  k_chol_dstep factorises TWO panels (k0 and k0 + 32)
// per launch: one launch gap and one cold start per 64 columns, the panel solve of the first panel and the update of
// the second half of the tile column overlap with the second diagonal block.
//
//   all tiles        A_tile -= P_i P_j^T for the (up to) two previous panels kpA, kpB (their trailing updates were
//                    deferred to this launch), tile column 0 keeps all 64 columns of its tiles in LDS
//   tile (0, 0)      wave 0 factorises L11 (lanes 32..63 carry the rows of L21 along) and publishes both behind flag[0];
//                    all four waves form A22 - L21 L21^T; wave 0 factorises L22 and publishes it behind flag[1];
//                    wave 1 carries the right-hand side (forward substitution folded in)
//   tiles (i > 0, 0) wave 0 solves its 64 rows against L11 as soon as flag[0] is up, all four waves subtract
//                    x1 L21^T from the second half, wave 0 solves against L22 behind flag[1]
// Tile (0, 0) is workgroup 0 of the 1-D grid and therefore resident before any waiter.  Requires n - k0 >= 64.
#pragma once

namespace satba {

// x L^T = p for one row per lane, in registers; Lb[c][r] = L[r][c] in LDS, inv[m] = 1 / L[m][m] in LDS.
// Row m + 1 of L^T is read from LDS while step m is computed.
__device__ __forceinline__ void chol_panel_rows(double (&x)[CH_NB], const double (*Lb)[CH_NB], const double* inv) {
    double cur[CH_NB], nxt[CH_NB];
#pragma unroll
    for (int c = 1; c < CH_NB; ++c) cur[c] = Lb[0][c];
    cur[0] = inv[0];
#pragma unroll
    for (int m = 0; m < CH_NB; ++m) {
        if (m + 1 < CH_NB) {
#pragma unroll
            for (int c = m + 2; c < CH_NB; ++c) nxt[c] = Lb[m + 1][c];
            nxt[m + 1] = inv[m + 1];
        }
        const double xm = x[m] * cur[m];  // cur[m] = 1 / L[m][m]
        x[m] = xm;
#pragma unroll
        for (int c = m + 1; c < CH_NB; ++c) x[c] -= xm * cur[c];  // L[c][m]
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = m + 1; c < CH_NB; ++c) cur[c] = nxt[c];
    }
}

// a 32 x 32 block of A (rows rb.., columns cb.., lower part if `lower`) published by tile (0, 0) -> LDS, one wave
__device__ __forceinline__ void chol_fetch_block(const double* A, int n, int rb, int cb, bool lower, double (*dst)[CH_NB], int lane) {
    double v[CH_NB * CH_NB / 64];
#pragma unroll
    for (int t = 0; t < CH_NB * CH_NB / 64; ++t) {
        const int idx = t * 64 + lane, r = idx % CH_NB, c = idx / CH_NB;
        v[t] = (!lower || c <= r) ? __hip_atomic_load(A + (size_t)(rb + r) + (size_t)(cb + c) * n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                  : 0.0;
    }
#pragma unroll
    for (int t = 0; t < CH_NB * CH_NB / 64; ++t) (&dst[0][0])[t * 64 + lane] = v[t];  // dst[c][r]
}

__global__ __launch_bounds__(256) void k_chol_dstep(double* __restrict__ A, int n, int kpA, int kpB, int k0, int* __restrict__ fail,
                                                    int* __restrict__ flag, double* __restrict__ b) {
    __shared__ double Pi[CH_NB][CH_LD], Pj[CH_NB][CH_LD];  // after the update: the tile's columns 0..31 (Pj) and 32..63 (Pi), [c][r]
    __shared__ double Lb[CH_NB][CH_NB], Lb2[CH_NB][CH_NB], L21s[CH_NB][CH_NB];  // [c][r] = L[r][c]
    __shared__ double lcol[2][64];
    __shared__ double pan[64][CH_MP];
    __shared__ double brow[64];
    __shared__ double ys[CH_NB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = (n - k0 + 63) / 64;
    int bi, bj;
    if ((int)blockIdx.x < T) { bi = blockIdx.x; bj = 0; }
    else {  // tiles (bi >= bj >= 1), row by row
        int idx = blockIdx.x - T;
        bi = 1;
        while (idx >= bi) { idx -= bi; ++bi; }
        bj = idx + 1;
    }
    if (bj != 0 && kpA < 0 && kpB < 0) return;  // first launch: nothing to apply to the trailing tiles
    const int r0 = k0 + bi * 64, c0 = k0 + bj * 64;

    // ---- trailing update with the previous two panels: A[r0.., c0..] -= P_i P_j^T, b[r0..] -= P_i y_prev
    // v_mfma_f64_16x16x4: wave w owns the 16 tile columns 16 w .. 16 w + 15 and all 64 rows (four 16 x 16 outputs); a lane holds
    // rows 16 rb + (lane & 15), columns 16 w + (lane >> 4) + 4 reg  (chol_mfma_update)
    {
        const int e16 = lane & 15, g4 = lane >> 4;
        chol_d4 old[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int r = r0 + 16 * rb + e16, c = c0 + 16 * wave + g4 + 4 * reg;
                old[rb][reg] = (r < n && c < n && r >= c) ? A[(size_t)r + (size_t)c * n] : 0.0;
            }
        if (bj == 0 && tid < 64) brow[tid] = (r0 + tid < n) ? b[r0 + tid] : 0.0;
        chol_d4 acc[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb] = chol_d4{0.0, 0.0, 0.0, 0.0};
        // the loads of the second panel are issued before the arithmetic of the first (registers, then LDS): one exposed
        // round of loads per launch instead of two
        constexpr int PF = CH_NB * 64 / 256;  // panel elements per thread and operand
        double pfi[PF], pfj[PF], pfy = 0.0;
        const int kq1 = kpA >= 0 ? kpA : kpB, kq2 = kpA >= 0 ? kpB : -1;  // first / second panel to apply (-1: none)
        if (kq1 >= 0) {
#pragma unroll
            for (int t = 0; t < PF; ++t) {
                const int idx = tid + t * 256, r = idx & 63, k = idx >> 6;
                Pi[k][r] = (r0 + r < n) ? A[(size_t)(r0 + r) + (size_t)(kq1 + k) * n] : 0.0;
                Pj[k][r] = (c0 + r < n) ? A[(size_t)(c0 + r) + (size_t)(kq1 + k) * n] : 0.0;
            }
            if (bj == 0 && tid >= 64 && tid < 64 + CH_NB) lcol[0][tid - 64] = b[kq1 + tid - 64];  // y of that panel
        }
        if (kq2 >= 0) {
#pragma unroll
            for (int t = 0; t < PF; ++t) {
                const int idx = tid + t * 256, r = idx & 63, k = idx >> 6;
                pfi[t] = (r0 + r < n) ? A[(size_t)(r0 + r) + (size_t)(kq2 + k) * n] : 0.0;
                pfj[t] = (c0 + r < n) ? A[(size_t)(c0 + r) + (size_t)(kq2 + k) * n] : 0.0;
            }
            if (bj == 0 && tid >= 64 && tid < 64 + CH_NB) pfy = b[kq2 + tid - 64];
        }
        for (int pass = 0; pass < 2; ++pass) {
            if ((pass == 0 ? kq1 : kq2) < 0) continue;
            if (pass == 1) {
                __syncthreads();  // the first pass is done with Pi, Pj, lcol
#pragma unroll
                for (int t = 0; t < PF; ++t) {
                    const int idx = tid + t * 256, r = idx & 63, k = idx >> 6;
                    Pi[k][r] = pfi[t];
                    Pj[k][r] = pfj[t];
                }
                if (bj == 0 && tid >= 64 && tid < 64 + CH_NB) lcol[0][tid - 64] = pfy;
            }
            __syncthreads();
            if (bj == 0 && tid < 64) {
                double s = 0.0;
#pragma unroll 8
                for (int k = 0; k < CH_NB; ++k) s += Pi[k][tid] * lcol[0][k];
                brow[tid] -= s;  // only this thread touches brow[tid] until the next barrier
            }
            chol_mfma_update(Pi, Pj, wave, lane, acc);
        }
        if (bj == 0 && tid < 64 && bi > 0 && r0 + tid < n) b[r0 + tid] = brow[tid];  // tile (0, 0): solved below
        if (bj == 0) __syncthreads();  // everyone is done reading Pi / Pj before they become the stash
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = 16 * rb + e16, col = 16 * wave + g4 + 4 * reg;
                const int r = r0 + row, c = c0 + col;
                const double v = old[rb][reg] - acc[rb][reg];
                if (bj == 0) {
                    if (col < CH_NB) Pj[col][row] = v;
                    else Pi[col - CH_NB][row] = v;
                } else if (r < n && c < n && r >= c) A[(size_t)r + (size_t)c * n] = v;
            }
    }
    if (bj != 0) return;
    __syncthreads();

    if (bi == 0) {
        // ---------------------------------------------------------------- tile (0, 0): L11, L21, then L22
        if (wave == 0) {
            double a[CH_NB];
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) {
                double v = Pj[c][lane];
                if (lane < CH_NB && c > lane) v = 0.0;  // above the diagonal
                a[c] = v;
            }
            // publish L11 and L21 as they are formed: agent-scope stores (write through to the coherence point) + flag, no release fence
            const bool bad = chol_diag_block(a, lane, pan, [&](int c, double v) {
                if (lane >= CH_NB || c <= lane)
                    __hip_atomic_store(A + (size_t)(k0 + lane) + (size_t)(k0 + c) * n, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            });
            __builtin_amdgcn_s_waitcnt(0);  // the stores above are acknowledged
            if (lane == 0) __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (bad && lane == 0) atomicOr(fail, 1);
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) {
                if (lane < CH_NB) Lb[c][lane] = a[c];
                else L21s[c][lane - CH_NB] = a[c];
            }
        }
        __syncthreads();
        {   // A22 <- A22 - L21 L21^T (rows / columns 32..63 of the tile: Pi[c][32 + r]); 4 columns per thread
            const int r = tid & 31, cg = tid >> 5;
            double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 8
            for (int k = 0; k < CH_NB; ++k) {
                const double lr = L21s[k][r];
#pragma unroll
                for (int j = 0; j < 4; ++j) s[j] += lr * L21s[k][cg * 4 + j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) Pi[cg * 4 + j][CH_NB + r] -= s[j];
        }
        if (wave == 1) {  // y1 = L11^-1 b1, then b2 -= L21 y1
            double v = (lane < CH_NB) ? brow[lane] : 0.0;
            const int cl = min(lane, CH_NB - 1);
            for (int m = 0; m < CH_NB; ++m) {
                const double ym = __shfl(v, m) / Lb[m][m];
                if (lane == m) v = ym;
                else if (lane > m && lane < CH_NB) v -= Lb[m][cl] * ym;
            }
            if (lane < CH_NB) { b[k0 + lane] = v; ys[lane] = v; }
            __builtin_amdgcn_wave_barrier();
            if (lane >= CH_NB) {
                double s = 0.0;
#pragma unroll 8
                for (int k = 0; k < CH_NB; ++k) s += L21s[k][lane - CH_NB] * ys[k];
                brow[lane] -= s;
            }
        }
        __syncthreads();
        if (wave == 0) {
            double a[CH_NB];
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) a[c] = (lane < CH_NB && c <= lane) ? Pi[c][CH_NB + lane] : 0.0;  // no riders
            const bool bad = chol_diag_block(a, lane, pan, [&](int c, double v) {
                if (lane < CH_NB && c <= lane)
                    __hip_atomic_store(A + (size_t)(k0 + CH_NB + lane) + (size_t)(k0 + CH_NB + c) * n, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            });
            __builtin_amdgcn_s_waitcnt(0);
            if (lane == 0) __hip_atomic_store(flag + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (bad && lane == 0) atomicOr(fail, 1);
            if (lane < CH_NB) {
#pragma unroll
                for (int c = 0; c < CH_NB; ++c) Lb2[c][lane] = a[c];
            }
        }
        __syncthreads();
        if (wave == 1) {  // y2 = L22^-1 b2
            double v = (lane < CH_NB) ? brow[CH_NB + lane] : 0.0;
            const int cl = min(lane, CH_NB - 1);
            for (int m = 0; m < CH_NB; ++m) {
                const double ym = __shfl(v, m) / Lb2[m][m];
                if (lane == m) v = ym;
                else if (lane > m && lane < CH_NB) v -= Lb2[m][cl] * ym;
            }
            if (lane < CH_NB) b[k0 + CH_NB + lane] = v;
        }
        return;
    }

    // -------------------------------------------------------------------- tiles (i > 0, 0): 64 panel rows, 64 columns
    // lanes past the last row of the matrix (last tile row only) duplicate the last valid row: the same arithmetic, the
    // same stores to the same addresses -- no store sits under a condition (a conditional store lets the compiler sink the
    // whole solve below its LDS reads: kilobytes of spills)
    const int lane_c = min(lane, n - 1 - r0);
    const int row = r0 + lane_c;
    if (wave == 0) {
        double x[CH_NB];
#pragma unroll
        for (int c = 0; c < CH_NB; ++c) x[c] = Pj[c][lane_c];
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(1);
        chol_fetch_block(A, n, k0, k0, true, Lb, lane);
        __builtin_amdgcn_wave_barrier();
        if (lane < CH_NB) lcol[0][lane] = 1.0 / Lb[lane][lane];
        __builtin_amdgcn_wave_barrier();
        chol_panel_rows(x, Lb, lcol[0]);
#pragma unroll
        for (int c = 0; c < CH_NB; ++c) {
            Pj[c][lane] = x[c];  // x1, operand of the update of the second half
            A[(size_t)row + (size_t)(k0 + c) * n] = x[c];
        }
    } else if (wave == 1) {
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(1);
        chol_fetch_block(A, n, k0 + CH_NB, k0, false, L21s, lane);
    }
    __syncthreads();
    {   // second half of the tile column: X2 <- X2 - x1 L21^T; thread = (row, 8 columns)
        const int r = tid & 63, cg = tid >> 6;
        double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
        for (int k = 0; k < CH_NB; ++k) {
            const double xr = Pj[k][r];
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += xr * L21s[k][cg * 8 + j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) Pi[cg * 8 + j][r] -= s[j];
    }
    __syncthreads();
    if (wave != 0) return;
    {
        double x[CH_NB];
#pragma unroll
        for (int c = 0; c < CH_NB; ++c) x[c] = Pi[c][lane_c];
        while (__hip_atomic_load(flag + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(1);
        chol_fetch_block(A, n, k0 + CH_NB, k0 + CH_NB, true, Lb, lane);
        __builtin_amdgcn_wave_barrier();
        if (lane < CH_NB) lcol[0][lane] = 1.0 / Lb[lane][lane];
        __builtin_amdgcn_wave_barrier();
        chol_panel_rows(x, Lb, lcol[0]);
#pragma unroll
        for (int c = 0; c < CH_NB; ++c) A[(size_t)row + (size_t)(k0 + CH_NB + c) * n] = x[c];
    }
}

}  // namespace satba
