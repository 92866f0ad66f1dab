// satba_chol.h -- dense SPD solve of the reduced camera system S dc = rhs on one GPU.
//
// S is (M n_p)^2 <= ~1200^2 float64, column-major, lower triangle valid (the Schur kernels only produce camera
// block pairs (a, b) with a <= b, which land in the column-major lower triangle).  This replaces the LSMR
// iteration of scipy:optimize/_lsq/trf.py:479-480 by an exact factorisation.
//
// More than 64 unknowns: ONE persistent launch factorises the matrix in 64 x 64 tiles and carries the right-hand side along
// (k_chol_tiles, satba_chol3.h), then the multi-workgroup backward substitution (k_trsv_back_mw).  Up to 64 unknowns (one tile; the
// reference's usual 2 .. 20 images x 3 parameters): the panel steps of rounds 1 - 2 below -- k_chol_dstep (two 32-column panels per
// launch, satba_chol2.h) or k_chol_step (one) -- and a one-wave backward substitution; there the tile kernel's fixed cost
// (31 us against 16 - 27) does not pay.
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

constexpr int CH_LD = 80;  // row stride (doubles) of the LDS panel copies: 160 dwords = 32 mod 64 banks, so the four k-rows an
                           // MFMA operand load touches (16 consecutive doubles each) fall into disjoint bank ranges
typedef double chol_d4 __attribute__((ext_vector_type(4)));
// acc[rb] += sum_k Pc[k][16 w + i] Pr[k][16 rb + j] for the 64 x 16 strip of a 64 x 64 tile that wave w owns, K = CH_NB, with
// v_mfma_f64_16x16x4 (A[i][k]: lane i + 16 k, B[k][j]: lane j + 16 k, D[i][j]: lane j + 16 (i & 3), register i >> 2):
// on return acc[rb][reg] of a lane is the update of tile row 16 rb + (lane & 15), tile column 16 w + (lane >> 4) + 4 reg.
// fp64 MFMA runs at the vector rate: what it saves is LDS operand traffic (40 instead of 256 reads per wave and panel) -- with
// one wave per SIMD the register-tiled form was bound by LDS latency (4.3 us per panel against ~1 us of arithmetic).
__device__ __forceinline__ void chol_mfma_update(const double (*Pr)[CH_LD], const double (*Pc)[CH_LD], int wave, int lane, chol_d4 (&acc)[4]) {
    const int kq = lane >> 4, e = lane & 15;
#pragma unroll
    for (int ks = 0; ks < CH_NB / 4; ++ks) {
        const double a = Pc[4 * ks + kq][16 * wave + e];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            const double bb = Pr[4 * ks + kq][16 * rb + e];
            acc[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc[rb], 0, 0, 0);
        }
    }
}

// Diagonal block in registers: lanes 0..31 hold the rows of the block (a[c], zero above the diagonal), lanes 32..63
// ride along (rows below the block: they end up holding L21).  Returns true if a pivot was not positive and finite.
//
// Micro-panels of 8 columns.  Inside a micro-panel everything stays in registers: the serial chain per column is
// l = a_j * rsqrt(d) -> pivot of column j + 1 from the lane's own value (a[j+1] - l * l) -> readlane -> rsqrt; the other
// columns of the micro-panel get L[c][j] by v_readlane (an SGPR operand of the multiply-add).  After a micro-panel its
// 8 values per row go to LDS once and the remaining columns receive a rank-8 update from LDS broadcasts.
// Round 1 broadcast every column through LDS: the LDS write -> read round trip (~60 ns) sat on the chain of the next
// pivot 32 times per block (6.5 us per block; tools/chol_times.py); here it is paid 3 times.
// pan: 64 x 8 doubles of LDS private to the wave.
constexpr int CH_MP = 8;
// pub(c, v): called for every column c of a finished micro-panel with the lane's final value -- the callers publish the block
// with write-through stores there, so that all but the last micro-panel's stores are acknowledged while the factorisation is
// still running (they used to be issued at the end: ~1.5 us of store latency on the chain of every panel).
template <class PUB>
__device__ __forceinline__ bool chol_diag_block(double (&a)[CH_NB], int lane, double (*pan)[CH_MP], PUB&& pub) {
    bool bad = false;
    double d = readlane_f64(a[0], 0);
    bad |= !(d > 1e-300) || !(d < 1e300);
    double h = half_rsqrt(d);
#pragma unroll
    for (int p = 0; p < CH_NB / CH_MP; ++p) {
#pragma unroll
        for (int jj = 0; jj < CH_MP; ++jj) {
            const int j = CH_MP * p + jj;
            const double a2 = a[j] + a[j];
            const double l = a2 * h;  // lane j: 2 d h = sqrt(d)
            a[j] = l;
            if (j + 1 < CH_NB) {
                const double piv = fma(-l, l, a[j + 1]);  // lane j + 1: its own l is L[j+1][j]
                if (jj + 1 < CH_MP) {  // next pivot inside the micro-panel: complete after this column
                    d = readlane_f64(piv, j + 1);
                    bad |= !(d > 1e-300) || !(d < 1e300);
                    h = half_rsqrt(d);
                }
#pragma unroll
                for (int c = j + 1; c < CH_MP * (p + 1); ++c) a[c] = fma(-l, readlane_f64(l, c), a[c]);
            }
        }
#pragma unroll
        for (int jj = 0; jj < CH_MP; ++jj) pub(CH_MP * p + jj, a[CH_MP * p + jj]);
        if (p + 1 < CH_NB / CH_MP) {
            // rank-8 update of the columns behind the micro-panel
            double2* row = reinterpret_cast<double2*>(&pan[lane][0]);
#pragma unroll
            for (int m = 0; m < CH_MP / 2; ++m) row[m] = make_double2(a[CH_MP * p + 2 * m], a[CH_MP * p + 2 * m + 1]);
            __builtin_amdgcn_wave_barrier();  // single wave: its LDS operations execute in order
#pragma unroll
            for (int c = CH_MP * (p + 1); c < CH_NB; ++c) {
                const double2* lc = reinterpret_cast<const double2*>(&pan[c][0]);  // row c of the panel: the same address for every lane
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int m = 0; m < CH_MP / 2; ++m) {
                    const double2 t = lc[m];
                    s0 = fma(a[CH_MP * p + 2 * m], t.x, s0);
                    s1 = fma(a[CH_MP * p + 2 * m + 1], t.y, s1);
                }
                a[c] -= s0 + s1;
                if (c == CH_MP * (p + 1)) {  // the next pivot: start its chain as early as possible
                    d = readlane_f64(a[c], c);
                    bad |= !(d > 1e-300) || !(d < 1e300);
                    h = half_rsqrt(d);
                }
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
        }
    }
    return bad;
}

// One panel step in one launch.  kp: first column of the previous panel (applied to the trailing matrix here; < 0:
// none), k0: first column of the panel that is factorised (k0 = kp + CH_NB or 0).  1-D grid over the lower 64x64
// tiles of A[k0:, k0:], tile column 0 first.  flag: one int per launch, zero on entry.
//
// Column broadcasts go through LDS (every lane reads the same address: one ds_read, no bank conflicts); the
// v_readlane version of the same loops took ~2x as long (two readlanes + hazard nops per multiply-add, measured with
// tools/chol_times.py).
// FULL: the panel has all CH_NB columns (every step but possibly the last).
template <bool FULL>
__global__ __launch_bounds__(256) void k_chol_step(double* __restrict__ A, int n, int npend, int k0, int* __restrict__ fail,
                                                   int* __restrict__ flag, double* __restrict__ b, long long* __restrict__ ts, const int* gate) {
    SATBA_GATE(gate);
    // npend: number of 32-column panels directly before k0 whose trailing update is still pending (1 after a single step, 2
    // after a double step, k_chol_dstep); they are applied one after the other in the same pass over the tile
    // ts (tools only, normally null): 8 wall-clock stamps of this step -- 0 start of tile (0,0), 1 its update done,
    // 2 diagonal block factorised (flag raised), 3 its wave done; 4..7 the same for tile (1, 0): start, update done, flag seen, end
    __shared__ double Pi[CH_NB][CH_LD], Pj[CH_NB][CH_LD];  // Pj doubles as the stash X[c][r] of the tile's first 32 columns
    __shared__ double Lb[CH_NB][CH_NB];              // L_kk, Lb[c][r] = L[r][c] (column-major like A)
    __shared__ double lcol[2][64];
    __shared__ double pan[64][CH_MP];
    __shared__ double brow[64];
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
    const int r0 = k0 + bi * 64, c0 = k0 + bj * 64;
    const bool stamp = ts && bj == 0 && bi < 2 && tid == 0;
    if (stamp) ts[bi * 4 + 0] = wall_clock64();

    if (npend > 0) {
        // ---- trailing update with the previous panel(s): A[r0.., c0..] -= P_i P_j^T, b[r0..] -= P_i y_prev
        // v_mfma_f64_16x16x4 (chol_mfma_update): wave w owns tile columns 16 w .. 16 w + 15; a lane holds rows 16 rb + (lane & 15),
        // columns 16 w + (lane >> 4) + 4 reg
        const int e16 = lane & 15, g4 = lane >> 4;
        chol_d4 old[4];  // the tile itself: in flight together with the panel loads
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int r = r0 + 16 * rb + e16, cc = c0 + 16 * wave + g4 + 4 * reg;
                old[rb][reg] = (r < n && cc < n && r >= cc) ? A[(size_t)r + (size_t)cc * n] : 0.0;
            }
        if (bj == 0 && tid < 64) brow[tid] = (r0 + tid < n) ? b[r0 + tid] : 0.0;
        chol_d4 acc[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb] = chol_d4{0.0, 0.0, 0.0, 0.0};
        for (int pass = 0; pass < npend; ++pass) {
            const int kq = k0 - CH_NB * (npend - pass);
            if (pass > 0) __syncthreads();  // the previous pass is done with Pi, Pj, lcol
            for (int idx = tid; idx < CH_NB * 64; idx += 256) {
                const int r = idx & 63, k = idx >> 6;
                Pi[k][r] = (r0 + r < n) ? A[(size_t)(r0 + r) + (size_t)(kq + k) * n] : 0.0;
                Pj[k][r] = (c0 + r < n) ? A[(size_t)(c0 + r) + (size_t)(kq + k) * n] : 0.0;
            }
            if (bj == 0 && tid >= 64 && tid < 64 + CH_NB) lcol[0][tid - 64] = b[kq + tid - 64];  // y of that panel
            __syncthreads();
            if (bj == 0 && tid < 64) {
                double s = 0.0;
#pragma unroll 8
                for (int k = 0; k < CH_NB; ++k) s += Pi[k][tid] * lcol[0][k];
                brow[tid] -= s;  // only this thread touches brow[tid] until the barrier below
            }
            chol_mfma_update(Pi, Pj, wave, lane, acc);
        }
        if (bj == 0 && tid < 64 && bi > 0 && r0 + tid < n) b[r0 + tid] = brow[tid];  // tile (0, 0): solved and stored below
        if (bj == 0) __syncthreads();  // everyone is done reading Pj before it becomes the stash
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = 16 * rb + e16, col = 16 * wave + g4 + 4 * reg;
                const int r = r0 + row, cc = c0 + col;
                const double v = old[rb][reg] - acc[rb][reg];
                if (bj == 0 && col < CH_NB) Pj[col][row] = v;  // panel columns are stored after the solve
                else if (r < n && cc < n && r >= cc) A[(size_t)r + (size_t)cc * n] = v;
            }
    } else {
        for (int idx = tid; idx < CH_NB * 64; idx += 256) {
            const int r = idx & 63, k = idx >> 6;
            Pj[k][r] = (r0 + r < n && k0 + k < n && r0 + r >= k0 + k) ? A[(size_t)(r0 + r) + (size_t)(k0 + k) * n] : 0.0;
        }
        if (tid < 64) brow[tid] = (r0 + tid < n) ? b[r0 + tid] : 0.0;
    }
    if (bj != 0) return;
    __syncthreads();
    if (stamp) ts[bi * 4 + 1] = wall_clock64();
    const int nb = FULL ? CH_NB : min(CH_NB, n - k0);

    if (bi == 0) {
        // ---- tile (0, 0): lanes 0..31 = rows of the diagonal block, lanes 32..63 = the first 32 panel rows
        if (wave == 0) {
            double a[CH_NB];
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) {
                double v = Pj[c][lane];
                if (lane < CH_NB && (c > lane || lane >= nb || c >= nb)) v = (c == lane) ? 1.0 : 0.0;  // identity padding
                if (c >= nb && lane >= CH_NB) v = 0.0;
                a[c] = v;
            }
            // publish L_kk as it is formed: agent-scope stores (write through to the coherence point) + flag, no release fence --
            // a fence writes back the whole L2 of this XCD (~2.5 us measured) while the other tiles are still storing.
            // A failed factorisation only raises the flag: the caller discards it.
            const bool bad = chol_diag_block(a, lane, pan, [&](int c, double v) {
                if (lane < nb && c <= lane) __hip_atomic_store(A + (size_t)(k0 + lane) + (size_t)(k0 + c) * n, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            });
            __builtin_amdgcn_s_waitcnt(0);  // the stores above are acknowledged
            if (lane == 0) __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (stamp) ts[2] = wall_clock64();
            if (bad && lane == 0) atomicOr(fail, 1);
            const int r = k0 + lane;
            if (lane >= CH_NB && r < n) {
#pragma unroll
                for (int c = 0; c < CH_NB; ++c)
                    if (c < nb) A[(size_t)r + (size_t)(k0 + c) * n] = a[c];
            }
            if (lane < CH_NB) {
#pragma unroll
                for (int c = 0; c < CH_NB; ++c) Lb[c][lane] = a[c];
            }
            if (stamp) ts[3] = wall_clock64();
        }
        __syncthreads();
        if (wave == 1) {  // right-hand side: y_k = L_kk^-1 b_k, lane = entry
            double v = (lane < nb) ? brow[lane] : 0.0;
            const int cl = min(lane, CH_NB - 1);
            for (int m = 0; m < nb; ++m) {
                const double ym = __shfl(v, m) / Lb[m][m];
                if (lane == m) v = ym;
                else if (lane > m && lane < nb) v -= Lb[m][cl] * ym;
            }
            if (lane < nb) b[k0 + lane] = v;
            else if (k0 + lane < n) b[k0 + lane] = brow[lane];
        }
        return;
    }
    // ---- tiles (i > 0, 0): 64 panel rows, x L_kk^T = p  (a partial panel is the last one: it has no rows below)
    if (!FULL || wave != 0) return;
    double x[CH_NB];
#pragma unroll
    for (int c = 0; c < CH_NB; ++c) x[c] = Pj[c][lane];  // columns >= nb of the stash are zero
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(1);
    if (stamp) ts[6] = wall_clock64();
    {   // L_kk: 16 coalesced loads per lane in flight, then into LDS (identity padding)
        double v[CH_NB * CH_NB / 64];
#pragma unroll
        for (int t = 0; t < CH_NB * CH_NB / 64; ++t) {
            const int idx = t * 64 + lane, r = idx % CH_NB, c = idx / CH_NB;
            v[t] = (r < nb && c <= r) ? __hip_atomic_load(A + (size_t)(k0 + r) + (size_t)(k0 + c) * n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                      : ((r == c) ? 1.0 : 0.0);
        }
#pragma unroll
        for (int t = 0; t < CH_NB * CH_NB / 64; ++t) (&Lb[0][0])[t * 64 + lane] = v[t];
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < CH_NB) lcol[0][lane] = 1.0 / Lb[lane][lane];
    __builtin_amdgcn_wave_barrier();
    const int r = r0 + lane;
    if (r >= n) return;  // before the arithmetic: stores under a condition would let the compiler sink all of it below
                         // the LDS reads (2.5 KB of spills)
    // row m + 1 of L^T is read from LDS while step m is computed; the barriers keep the compiler from hoisting all
    // 528 reads (spills) or sinking the arithmetic below them
    double cur[CH_NB], nxt[CH_NB];
#pragma unroll
    for (int c = 1; c < CH_NB; ++c) cur[c] = Lb[0][c];
    cur[0] = lcol[0][0];
#pragma unroll
    for (int m = 0; m < CH_NB; ++m) {
        if (m + 1 < CH_NB) {
#pragma unroll
            for (int c = m + 2; c < CH_NB; ++c) nxt[c] = Lb[m + 1][c];
            nxt[m + 1] = lcol[0][m + 1];
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
#pragma unroll
    for (int c = 0; c < CH_NB; ++c)
        if (FULL || c < nb) A[(size_t)r + (size_t)(k0 + c) * n] = x[c];
    if (stamp) ts[7] = wall_clock64();
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

// L^T z = y for n <= 64, one wave, straight from the factor's lower triangle (no mirror, no block inverses): lane c holds column c
// of L below the diagonal in registers (= row c of L^T) and its running right-hand side; step k (bottom up) broadcasts z_k and
// every lane in front of it takes its term off.  Few cameras are the common case of the reference's pipelines (2 .. 20 images x 3
// parameters); at 10 x 5 the mirror + inverses kernel and the multi-workgroup back-substitution were 19 us of a 165 us iteration,
// this is 5.  (Tried with it: the factorisation in the same single wave, lane = row -- 38 us against 21 + 5 for k_chol_dstep's
// four-wave diagonal blocks plus this kernel: dropped.)
constexpr int CH_SMALL = 64;
__global__ __launch_bounds__(64) void k_trsv_back_small(const double* __restrict__ L, int n, double* __restrict__ b, const int* gate) {
    SATBA_GATE(gate);
    const int c = threadIdx.x;
    const int cc = min(c, n - 1);
    double t[CH_SMALL];
#pragma unroll
    for (int k = 0; k < CH_SMALL; ++k) t[k] = (k < n && k > c) ? L[(size_t)k + (size_t)cc * n] : 0.0;
    const double inv = 1.0 / L[(size_t)cc + (size_t)cc * n];
    double y = (c < n) ? b[c] : 0.0;
#pragma unroll
    for (int k = CH_SMALL - 1; k >= 0; --k) {
        if (k < n) {  // (uniform)
            const double zk = readlane_f64(y * inv, k);
            y = (c == k) ? zk : fma(-t[k], zk, y);  // t[k] is zero at and behind the diagonal (lanes >= k)
        }
    }
    if (c < n) b[c] = y;
}

}  // namespace satba
#include "satba_chol2.h"
#include "satba_chol3.h"
namespace satba {

constexpr int CH_MAX_STEPS = 256;  // flag words behind the not-SPD flag (the small path's panel flags, then the backward substitution's)
constexpr int CH_TRSV_FLAGS = 64;  // k_trsv_back_mw's flags

// k_chol_dstep's tile column + scratch exceed the 64 KB a kernel gets without asking (called once per process)
inline void cholesky_init() {
    static const bool once = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chol_dstep), hipFuncAttributeMaxDynamicSharedMemorySize, (int)chol_dstep_lds());
        return true;
    }();
    (void)once;
    chol_tiles_init();
}

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
    if (n <= CH_SMALL) {
        // one tile: panel steps (two panels per launch while more than 32 columns remain, then one) + the one-wave backward substitution
        int k0 = 0, npend = 0;
        int* fl = flags;
        for (; n - k0 > CH_NB; k0 += 2 * CH_NB, fl += 2, npend = 2) {  // (the second panel may be partial)
            const int T = (n - k0 + 63) / 64;
            hipLaunchKernelGGL(k_chol_dstep, dim3(T * (T + 1) / 2), dim3(256), chol_dstep_lds(), stream, A, n, npend, k0, fail, fl, b, (long long*)nullptr, gate);
        }
        for (; k0 < n; k0 += CH_NB, ++fl, npend = 1) {
            const int T = (n - k0 + 63) / 64;
            if (n - k0 >= CH_NB)
                hipLaunchKernelGGL(k_chol_step<true>, dim3(T * (T + 1) / 2), dim3(256), 0, stream, A, n, npend, k0, fail, fl, b, (long long*)nullptr, gate);
            else
                hipLaunchKernelGGL(k_chol_step<false>, dim3(T * (T + 1) / 2), dim3(256), 0, stream, A, n, npend, k0, fail, fl, b, (long long*)nullptr, gate);
        }
        hipLaunchKernelGGL(k_trsv_back_small, dim3(1), dim3(64), 0, stream, A, n, b, gate);
        return false;
    }
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
