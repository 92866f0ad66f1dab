// satba_chol2.h -- double panel step of the dense Cholesky (included by satba_chol.h): two panels of 32 columns per launch.
//
// A single panel step costs ~20 us of chain that no amount of parallelism shortens: launch gap (3 us), trailing update of the
// tile that holds the diagonal block (2.3 us per pending panel: 64 x 64 x 32 fp64 multiply-adds on one CU, plus the cold
// loads), diagonal block (~6 us), panel solve behind a flag (4 us).  k_chol_dstep factorises TWO panels (k0 and k0 + 32) per
// launch: one gap and one cold start per 64 columns, and the panel solve of the first panel overlaps the second diagonal block.
//
// Tile column 0 of a launch = the 64-row tiles of columns k0 .. k0 + 63, one workgroup each, the tile resident in LDS
// (W[p] = its 32 columns of panel p, [c][r]).
//   all tiles        first apply the `npend` panels of the previous launch(es) (trailing update, v_mfma_f64_16x16x4; the operands
//                    of two panels are in flight in registers)
//   tile (0, 0)      the four waves factorise D0 (chol_diag_block4: 8 columns each, lanes 32..63 carry the rows of L21 along) and
//                    publish both behind flag[0]; three waves form A22 - L21 L21^T (MFMA); the four waves factorise D1 and
//                    publish it behind flag[1].  Wave 0 carries the right-hand side (forward substitution folded in) beside the
//                    rest of the second diagonal block -- off the chain.
//   tiles (i > 0, 0) wave 0 solves its 64 rows against D0 as soon as flag[0] is up while waves 1..3 fetch L21; all four waves
//                    subtract X0 L21^T from the second half; wave 0 solves against D1 behind flag[1].
// Tile (0, 0) is workgroup 0 of the 1-D grid and therefore resident before any waiter.  Requires n - k0 > 32 (the second panel
// may be partial: the last launch of a matrix whose size is not a multiple of 64).
//
// Tried and dropped in round 2: per-micro-panel flags for D1 with the waiting tiles' solve staged behind them (the steps of the
// first 24 columns before the block is complete, only an 8 x 8 fetch at the end): rows solved 0.8 us earlier by the stamps,
// nothing in the LM loop (649 it/s either way).
// Tried and dropped in round 2: handing the diagonal blocks over through NaN-initialised mailboxes that the waiting tiles poll
// instead of a flag behind the data (one memory trip less on paper): the solved rows arrived at the same time (21.0 vs 21.2 us
// into the launch) and the kernel ended 1.5 us later.
// Round 2 tried four panels per launch (row tile 1 takes the chain over after two diagonal blocks): 57 us per 128 columns
// against 2 x 27.5 here -- the trailing update of four pending panels on one CU costs what the saved launch gap gains.
#pragma once

namespace satba {

constexpr int CH_LS = 48;  // row stride of a fetched 32 x 32 factor block used as MFMA operand (96 dwords = 32 mod 64 banks)
constexpr int CH_TS = 24;  // time stamps per launch (tools/chol_times.py)

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

// acc[i] += sum_k Pc[k][16 c16 + u] Pr[k][16 (rb0 + i) + v]  for i < NRB, K = CH_NB  (lane mapping of chol_mfma_update: on
// return acc[i][reg] of a lane belongs to row 16 (rb0 + i) + (lane & 15), column 16 c16 + (lane >> 4) + 4 reg)
template <int NRB>
__device__ __forceinline__ void chol_mfma_acc(const double* Pr, int ldr, const double* Pc, int ldc, int c16, int rb0, int lane, chol_d4 (&acc)[NRB]) {
    const int kq = lane >> 4, e = lane & 15;
#pragma unroll
    for (int ks = 0; ks < CH_NB / 4; ++ks) {
        const double a = Pc[(4 * ks + kq) * ldc + 16 * c16 + e];
#pragma unroll
        for (int i = 0; i < NRB; ++i) {
            const double bb = Pr[(4 * ks + kq) * ldr + 16 * (rb0 + i) + e];
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc[i], 0, 0, 0);
        }
    }
}

__device__ __forceinline__ void chol_spin(const int* f) {
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(1);
}
__device__ __forceinline__ void chol_spin_lds(const volatile int* f) {  // hand-over between two waves of a workgroup
    while (*f == 0) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}

// y = D^-1 v for a 32 x 32 lower triangular D (Lb[c][r] = D[r][c] in LDS), lane = entry (lanes >= 32 idle)
__device__ __forceinline__ double chol_fwd32(double v, const double (*Lb)[CH_NB], int lane) {
    const int cl = min(lane, CH_NB - 1);
    for (int m = 0; m < CH_NB; ++m) {
        const double ym = __shfl(v, m) / Lb[m][m];
        if (lane == m) v = ym;
        else if (lane > m && lane < CH_NB) v -= Lb[m][cl] * ym;
    }
    return v;
}

// Diagonal block by FOUR waves: wave w owns columns 8 w .. 8 w + 7 (one micro-panel of chol_diag_block) of the 32 x 32 block for
// all 64 lanes (lanes 0..31: rows of the block, zero above the diagonal on entry; lanes 32..63: riders).  The chain moves from
// wave to wave: wave p factorises its micro-panel in registers exactly like chol_diag_block does, puts it into LDS (pan[p]) and
// raises lf[p]; the waves behind it apply it to their own columns (rank-8 update) as soon as it is there -- all but the update of
// the NEXT micro-panel is off the chain.  One wave doing everything issued ~65 instructions per column (the rank-8 updates of up
// to 24 columns behind every micro-panel): 5.5 us per block, instruction-bound; here the chain sees 8 columns + one update.
// pan: [4][64][CH_MP] doubles of LDS, lf: 4 ints of LDS, zero on entry.  pub(c, v): publish column c of the factor (own lanes).
// Returns true if a pivot of this wave's micro-panel was not positive and finite.
template <class PUB>
__device__ __forceinline__ bool chol_diag_block4(double (&a)[CH_MP], int w, int lane, double (*pan)[64][CH_MP], volatile int* lf, PUB&& pub) {
    bool bad = false;
#pragma unroll
    for (int p = 0; p < CH_NB / CH_MP; ++p) {
        if (w == p) {
            double d = readlane_f64(a[0], CH_MP * p);
            bad |= !(d > 1e-300) || !(d < 1e300);
            double h = half_rsqrt(d);
#pragma unroll
            for (int jj = 0; jj < CH_MP; ++jj) {
                const double a2 = a[jj] + a[jj];
                const double l = a2 * h;  // lane j: 2 d h = sqrt(d)
                a[jj] = l;
                if (jj + 1 < CH_MP) {
                    const double piv = fma(-l, l, a[jj + 1]);  // lane j + 1: its own l is L[j+1][j]
                    d = readlane_f64(piv, CH_MP * p + jj + 1);
                    bad |= !(d > 1e-300) || !(d < 1e300);
                    h = half_rsqrt(d);
#pragma unroll
                    for (int c = jj + 1; c < CH_MP; ++c) a[c] = fma(-l, readlane_f64(l, CH_MP * p + c), a[c]);
                }
            }
            double2* row = reinterpret_cast<double2*>(&pan[p][lane][0]);
#pragma unroll
            for (int m = 0; m < CH_MP / 2; ++m) row[m] = make_double2(a[2 * m], a[2 * m + 1]);
            asm volatile("" ::: "memory");
            if (lane == 0) lf[p] = 1;  // the LDS unit executes a wave's operations in order: the data are in place before the flag
#pragma unroll
            for (int jj = 0; jj < CH_MP; ++jj) pub(CH_MP * p + jj, a[jj]);
        } else if (w > p) {
            while (lf[p] == 0) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            const double2* row = reinterpret_cast<const double2*>(&pan[p][lane][0]);
            double r[CH_MP];
#pragma unroll
            for (int m = 0; m < CH_MP / 2; ++m) { const double2 t = row[m]; r[2 * m] = t.x; r[2 * m + 1] = t.y; }
#pragma unroll
            for (int jj = 0; jj < CH_MP; ++jj) {
                const double2* lc = reinterpret_cast<const double2*>(&pan[p][CH_MP * w + jj][0]);  // row c of the panel: the same address for every lane
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int m = 0; m < CH_MP / 2; ++m) {
                    const double2 t = lc[m];
                    s0 = fma(r[2 * m], t.x, s0);
                    s1 = fma(r[2 * m + 1], t.y, s1);
                }
                a[jj] -= s0 + s1;
            }
        }
    }
    return bad;
}

// dynamic LDS of k_chol_dstep (more than the 64 KB a kernel may declare statically)
constexpr size_t chol_dstep_lds() {
    return sizeof(double) * (2 * CH_NB * CH_LD + CH_NB * CH_LS + 2 * CH_NB * CH_NB + 4 * 64 * CH_MP + 64 + 64 + 2 * CH_NB + 8);
}

__global__ __launch_bounds__(256) void k_chol_dstep(double* __restrict__ A, int n, int npend, int k0, int* __restrict__ fail,
                                                    int* __restrict__ flag, double* __restrict__ b, long long* __restrict__ ts, const int* gate) {
    SATBA_GATE(gate);
    // npend: number of 32-column panels directly before k0 whose trailing update is still pending (0, 1 or 2)
    // flag: two ints per launch, zero on entry: [p] diagonal block p (and, p = 0, L21) is visible
    // ts (tools only, normally null): CH_TS wall-clock stamps of this launch, 8 per row tile 0..2 of tile column 0:
    //   0 start, 1 trailing update done, 2 + p: panel p done (tile 0: diagonal block published; others: rows solved)
    extern __shared__ double s_ds[];
    // W[2][32][CH_LD] (during the trailing update: its operands) | Lr[32][CH_LS] | Lb[2][32][32] | pan[4][64][CH_MP] | brow[64] | lcol[64]
    // | yv[2][32] | lf[16] (ints)           (chol_dstep_lds)
    double (*W)[CH_NB][CH_LD] = reinterpret_cast<double (*)[CH_NB][CH_LD]>(s_ds);
    double* Pi = &W[0][0][0];
    double* Pj = &W[1][0][0];
    double (*Lr)[CH_LS] = reinterpret_cast<double (*)[CH_LS]>(s_ds + 2 * CH_NB * CH_LD);
    double (*Lb)[CH_NB][CH_NB] = reinterpret_cast<double (*)[CH_NB][CH_NB]>(s_ds + 2 * CH_NB * CH_LD + CH_NB * CH_LS);
    double (*pan)[64][CH_MP] = reinterpret_cast<double (*)[64][CH_MP]>(&Lb[0][0][0] + 2 * CH_NB * CH_NB);
    double* brow = &pan[0][0][0] + 4 * 64 * CH_MP;
    double* lcol = brow + 64;
    double (*yv)[CH_NB] = reinterpret_cast<double (*)[CH_NB]>(lcol + 64);
    int* lf = reinterpret_cast<int*>(&yv[0][0] + 2 * CH_NB);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = (n - k0 + 63) / 64;
    int bi, bj;
    if ((int)blockIdx.x < T) { bi = blockIdx.x; bj = 0; }
    else {  // trailing tiles (bi >= bj >= 1), row by row
        int idx = blockIdx.x - T;
        bi = 1;
        while (idx >= bi) { idx -= bi; ++bi; }
        bj = idx + 1;
    }
    const bool col0 = bj == 0;
    if (!col0 && npend == 0) return;  // first launch: nothing to apply to the trailing tiles
    const int r0 = k0 + bi * 64, c0 = k0 + bj * 64;
    const int e16 = lane & 15, g4 = lane >> 4;
    const bool stamp = ts && col0 && bi < 3 && tid == 0;
    if (stamp) ts[bi * 8 + 0] = wall_clock64();
    if (col0 && tid < 16) lf[tid] = 0;

    // ------------------------------------------------------------------ trailing update with the pending panels
    // A[r0.., c0..] -= P_i P_j^T, b[r0..] -= P_i y_prev, 16 x 16 blocks by v_mfma_f64_16x16x4 (64 cycles each: the update of a
    // tile is bound by the MFMA rate of its CU).  Block (rb, cb) = rows 16 rb.., columns 16 cb.. of the tile; a lane holds rows
    // 16 rb + (lane & 15), columns 16 cb + (lane >> 4) + 4 reg.  Ordinary tiles: wave w takes the four blocks of column block w.
    // The diagonal tile (0, 0) -- the one on the chain -- only needs its lower ten blocks; they are dealt 3 / 3 / 2 / 2, and the
    // wave with the least takes the right-hand side, too.
    {
        const bool diag = col0 && bi == 0;
        int nblk = 4, rbs[4], cbs[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { rbs[t] = t; cbs[t] = wave; }
        if (diag) {
            if (wave == 0) { nblk = 3; }                                              // (0,0) (1,0) (2,0)
            else if (wave == 1) { nblk = 3; rbs[0] = 1; rbs[1] = 2; rbs[2] = 3; }      // (1,1) (2,1) (3,1)
            else if (wave == 2) { nblk = 2; rbs[0] = 2; rbs[1] = 3; }                  // (2,2) (3,2)
            else { nblk = 2; rbs[0] = 3; rbs[1] = 3; cbs[1] = 0; }                     // (3,3) (3,0)
        }
        const int bwave = diag ? 3 : 0;  // the wave that carries the right-hand side of the tile's rows
        chol_d4 old[4], acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc[t] = chol_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int r = r0 + 16 * rbs[t] + e16, c = c0 + 16 * cbs[t] + g4 + 4 * reg;
                old[t][reg] = (t < nblk && r < n && c < n && r >= c) ? A[(size_t)r + (size_t)c * n] : 0.0;
            }
        }
        if (col0 && tid < 64) brow[tid] = (r0 + tid < n) ? b[r0 + tid] : 0.0;
        // operands of one pending panel: my rows (Pi) and the rows of the tile's column range (Pj); both panels in flight in registers
        constexpr int PF = CH_NB * 64 / 256;
        struct Operands { double i[PF], j[PF], y; };
        auto fetch = [&](Operands& o, int kq) {
#pragma unroll
            for (int t = 0; t < PF; ++t) {
                const int idx = tid + t * 256, r = idx & 63, k = idx >> 6;
                const size_t col = (size_t)(kq + k) * n;
                o.i[t] = (r0 + r < n) ? A[(size_t)(r0 + r) + col] : 0.0;
                o.j[t] = (c0 + r < n) ? A[(size_t)(c0 + r) + col] : 0.0;
            }
            o.y = (col0 && tid >= 64 && tid < 64 + CH_NB) ? b[kq + tid - 64] : 0.0;  // y of that panel
        };
        auto apply = [&](Operands& o, int pass) {
            if (pass > 0) __syncthreads();  // the previous pass is done with the operand arrays
#pragma unroll
            for (int t = 0; t < PF; ++t) {
                const int idx = tid + t * 256, r = idx & 63, k = idx >> 6;
                Pi[k * CH_LD + r] = o.i[t]; Pj[k * CH_LD + r] = o.j[t];
            }
            if (col0 && tid >= 64 && tid < 64 + CH_NB) lcol[tid - 64] = o.y;
            __syncthreads();
            if (col0 && wave == bwave) {
                double s = 0.0;
#pragma unroll 8
                for (int k = 0; k < CH_NB; ++k) s += Pi[k * CH_LD + lane] * lcol[k];
                brow[lane] -= s;  // only this lane touches brow[lane] until the next barrier
            }
            const int kq = lane >> 4;
#pragma unroll
            for (int ks = 0; ks < CH_NB / 4; ++ks) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (t < nblk) {
                        const double av = Pj[(4 * ks + kq) * CH_LD + 16 * cbs[t] + e16];
                        const double bv = Pi[(4 * ks + kq) * CH_LD + 16 * rbs[t] + e16];
                        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[t], 0, 0, 0);
                    }
                }
            }
        };
        Operands oa, ob;
        if (npend > 0) fetch(oa, k0 - CH_NB * npend);
        if (npend > 1) fetch(ob, k0 - CH_NB);
        if (npend > 0) apply(oa, 0);
        if (npend > 1) apply(ob, 1);
        // rows below tile 0: their right-hand side is final for this launch (the panels of this launch are applied to it by the next
        // one); the rows of tile 0 stay in brow
        if (col0 && tid < 64 && bi > 0 && r0 + tid < n) b[r0 + tid] = brow[tid];
        if (col0) __syncthreads();  // everyone is done with the operand arrays before they become the tile
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = 16 * rbs[t] + e16, col = 16 * cbs[t] + g4 + 4 * reg;
                const double v = old[t][reg] - acc[t][reg];
                if (t < nblk) {
                    if (col0) {
                        W[col >> 5][col & 31][row] = v;  // the tile stays in LDS
                    } else {
                        const int r = r0 + row, c = c0 + col;
                        if (r < n && c < n && r >= c) __hip_atomic_store(A + (size_t)r + (size_t)c * n, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // write-through: nothing left to flush at the end of the kernel
                    }
                }
            }
    }
    if (!col0) return;
    __syncthreads();
    if (stamp) ts[bi * 8 + 1] = wall_clock64();

    if (bi == 0) {
        // ---------------------------------------------------------------- tile (0, 0): D0 + L21, A22 -= L21 L21^T, D1
        // lf: [0..3] micro-panels of D0, [4..7] of D1, [8] waves done with D0 (stores acknowledged), [9] the same for D1
        // nb2 < 32: the matrix ends inside the second panel (then this is its last launch and its only tile): the missing rows and
        // columns of D1 are padded with the identity, nothing is stored for them
        const int nb2 = min(CH_NB, n - k0 - CH_NB);
        volatile int* vlf = lf;
        double a[CH_MP];
        {
#pragma unroll
            for (int jj = 0; jj < CH_MP; ++jj) {
                const int c = CH_MP * wave + jj;
                double v = W[0][c][lane];
                if (lane < CH_NB && c > lane) v = 0.0;  // above the diagonal
                a[jj] = v;
            }
            // published as it is formed: agent-scope stores (write through to the coherence point), then the flag -- no release
            // fence (a fence writes back the whole L2 of this XCD, ~2.5 us, while the other tiles are still storing)
            const bool bad = chol_diag_block4(a, wave, lane, pan, vlf, [&](int c, double v) {
                if (lane >= CH_NB ? lane < CH_NB + nb2 : c <= lane)
                    __hip_atomic_store(A + (size_t)(k0 + lane) + (size_t)(k0 + c) * n, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            });
            if (bad && lane == 0) atomicOr(fail, 1);
            __builtin_amdgcn_s_waitcnt(0);  // my stores are acknowledged
            if (wave < 3) {
                if (lane == 0) __hip_atomic_fetch_add(&lf[8], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {  // the last micro-panel: the other three waves stored theirs long ago
                while (vlf[8] < 3) __builtin_amdgcn_s_sleep(1);
                if (lane == 0) __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (ts && lane == 0) ts[2] = wall_clock64();
            }
#pragma unroll
            for (int jj = 0; jj < CH_MP; ++jj) {
                const int c = CH_MP * wave + jj;
                if (lane >= CH_NB) W[0][c][lane] = a[jj];  // L21: operand of the update below and of the right-hand side
                else Lb[0][c][lane] = a[jj];
            }
        }
        if (stamp) ts[4] = wall_clock64();
        __syncthreads();
        if (wave != 1) {
            // A22 (rows / columns 32..63 of the tile: W[1][c][32 + r]) -= L21 L21^T: the three 16 x 16 blocks of its lower part, one
            // per wave (a v_mfma_f64_16x16x4 takes 64 cycles: one wave doing all three was slower than this with its two barriers)
            const int rb = wave == 0 ? 0 : 1, cb = wave == 3 ? 1 : 0;
            chol_d4 u[1] = {chol_d4{0.0, 0.0, 0.0, 0.0}};
            chol_mfma_acc<1>(&W[0][0][32], CH_LD, &W[0][0][32], CH_LD, cb, rb, lane, u);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) W[1][16 * cb + g4 + 4 * reg][32 + 16 * rb + e16] -= u[0][reg];
        }
        __syncthreads();
        {
#pragma unroll
            for (int jj = 0; jj < CH_MP; ++jj) {
                const int c = CH_MP * wave + jj;
                double v = 0.0;  // no riders
                if (lane < nb2 && c <= lane) v = W[1][c][CH_NB + lane];
                else if (lane < CH_NB && c == lane) v = 1.0;  // identity padding
                a[jj] = v;
            }
            const bool bad = chol_diag_block4(a, wave, lane, pan, vlf + 4, [&](int c, double v) {
                if (lane < nb2 && c <= lane)
                    __hip_atomic_store(A + (size_t)(k0 + CH_NB + lane) + (size_t)(k0 + CH_NB + c) * n, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            });
            if (bad && lane == 0) atomicOr(fail, 1);
            if (lane < CH_NB) {
#pragma unroll
                for (int jj = 0; jj < CH_MP; ++jj) Lb[1][CH_MP * wave + jj][lane] = a[jj];
            }
            __builtin_amdgcn_s_waitcnt(0);  // my stores are acknowledged, my columns of D1 are in LDS
            if (wave < 3) {
                if (lane == 0) __hip_atomic_fetch_add(&lf[9], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                while (vlf[9] < 3) __builtin_amdgcn_s_sleep(1);
                if (lane == 0) __hip_atomic_store(flag + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (ts && lane == 0) ts[3] = wall_clock64();
                if (lane == 0) __hip_atomic_fetch_add(&lf[9], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (wave > 0) return;
        }
        // wave 0 (its micro-panel of D1 was the first): the right-hand side, beside the rest of the second diagonal block:
        // y0 = D0^-1 b (rows 0..31), b (rows 32..63) -= L21 y0, then y1 = D1^-1 b (rows 32..63) once D1 is complete
        {
            double v = chol_fwd32((lane < CH_NB) ? brow[lane] : 0.0, Lb[0], lane);
            if (lane < CH_NB) { yv[0][lane] = v; b[k0 + lane] = v; }
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
            double bv = 0.0;
            if (lane >= CH_NB) {
                double s = 0.0;
#pragma unroll 8
                for (int k = 0; k < CH_NB; ++k) s += W[0][k][lane] * yv[0][k];
                bv = brow[lane] - s;
            }
            bv = __shfl(bv, CH_NB + (lane & (CH_NB - 1)));  // row 32 + l to lane l
            while (vlf[9] < 4) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            v = chol_fwd32((lane < CH_NB) ? bv : 0.0, Lb[1], lane);
            if (lane < nb2) b[k0 + CH_NB + lane] = v;
        }
        return;
    }

    // -------------------------------------------------------------------- tiles (i > 0, 0): 64 panel rows, 64 columns
    // lanes past the last row of the matrix (last row tile only) duplicate the last valid row: the same arithmetic, the same
    // stores to the same addresses -- no store sits under a condition (a conditional store lets the compiler sink the whole solve
    // below its LDS reads: kilobytes of spills)
    const int lane_c = min(lane, n - 1 - r0);
    const size_t row = (size_t)(r0 + lane_c);
    if (wave == 0) {
        double x[CH_NB];
#pragma unroll
        for (int c = 0; c < CH_NB; ++c) x[c] = W[0][c][lane_c];
        chol_spin(flag);
        chol_fetch_block(A, n, k0, k0, true, Lb[0], lane);
        __builtin_amdgcn_wave_barrier();
        if (lane < CH_NB) lcol[lane] = 1.0 / Lb[0][lane][lane];
        __builtin_amdgcn_wave_barrier();
        chol_panel_rows(x, Lb[0], lcol);
#pragma unroll
        for (int c = 0; c < CH_NB; ++c) {
            W[0][c][lane] = x[c];  // X0, operand of the update of the second half
            A[row + (size_t)(k0 + c) * n] = x[c];
        }
        if (stamp) ts[bi * 8 + 2] = wall_clock64();
    } else {
        // waves 1..3, while wave 0 solves: L21 (rows k0 + 32.., columns k0..) -> Lr[k][i]
        chol_spin(flag);
        constexpr int PER = (CH_NB * CH_NB + 191) / 192;
        double v[PER];
#pragma unroll
        for (int t = 0; t < PER; ++t) {
            const int idx = (tid - 64) + t * 192, i = idx % CH_NB, k = idx / CH_NB;
            v[t] = (idx < CH_NB * CH_NB)
                       ? __hip_atomic_load(A + (size_t)(k0 + CH_NB + i) + (size_t)(k0 + k) * n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                       : 0.0;
        }
#pragma unroll
        for (int t = 0; t < PER; ++t) {
            const int idx = (tid - 64) + t * 192, i = idx % CH_NB, k = idx / CH_NB;
            if (idx < CH_NB * CH_NB) Lr[k][i] = v[t];
        }
    }
    __syncthreads();
    {   // second half of the tile column: W[1] -= X0 L21^T; wave -> 16 columns x 32 rows
        chol_d4 u[2] = {chol_d4{0.0, 0.0, 0.0, 0.0}, chol_d4{0.0, 0.0, 0.0, 0.0}};
        const int c16 = wave & 1, rb0 = 2 * (wave >> 1);
        chol_mfma_acc<2>(&W[0][0][0], CH_LD, &Lr[0][0], CH_LS, c16, rb0, lane, u);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) W[1][16 * c16 + g4 + 4 * reg][16 * (rb0 + i) + e16] -= u[i][reg];
    }
    __syncthreads();
    if (wave != 0) return;
    {
        double x[CH_NB];
#pragma unroll
        for (int c = 0; c < CH_NB; ++c) x[c] = W[1][c][lane_c];
        chol_spin(flag + 1);
        chol_fetch_block(A, n, k0 + CH_NB, k0 + CH_NB, true, Lb[0], lane);
        __builtin_amdgcn_wave_barrier();
        if (lane < CH_NB) lcol[lane] = 1.0 / Lb[0][lane][lane];
        __builtin_amdgcn_wave_barrier();
        chol_panel_rows(x, Lb[0], lcol);
#pragma unroll
        for (int c = 0; c < CH_NB; ++c) A[row + (size_t)(k0 + CH_NB + c) * n] = x[c];
        if (stamp) ts[bi * 8 + 3] = wall_clock64();
    }
}

}  // namespace satba
