// satba_linearize.h -- K2 v2: fused residual + analytic Jacobian -> normal-equation blocks, without LDS atomics
// on float64.
//
// v1 (k_linearize in satba_kernels.h) accumulates the per-camera blocks with 20-27 ds_add_f64 per observation and
// reduces the per-point blocks with 108 ds_bpermute per tile; PMC showed the LDS pipeline busy ~65 % of the kernel
// (profiles/r1_pmc_linearize_schur.txt).  v2 routes both reductions through an LDS staging area instead:
//
//   phase A (per wave tile, lane = observation):
//     - residual, J_c (2 x NP), J_p (2 x 3) in registers, camera constants read from an LDS copy of the table;
//     - the 9 per-point products are written to the wave's own staging rows, then lane (q, v) sums value v over
//       the observations of the q-th point of the tile and stores V / g_p directly (no shuffles, no atomics);
//     - the CU = NP(NP+1)/2 + NP per-camera products are written to stage[v][slot] (conflict-free ds_write_b64)
//       and the slot is appended to its camera's bucket (one ds_add_rtn_u32 per observation);
//   phase B (after a workgroup barrier): thread (camera c, value v) owns that accumulator IN A REGISTER for the
//     whole kernel and adds stage[v][slot] for the slots in bucket c.  Every staged value is read exactly once.
//   At the end the owners store their accumulators to this workgroup's partial (plain coalesced stores);
//   k_lin_finish sums the partials.
//
// LDS traffic drops from ~(20 atomics + 108 permutes) to ~(29 writes + 29 reads) of plain 8-byte accesses per
// observation.  Buckets have a fixed capacity; an overflowing observation (rare) adds its products to a global
// overflow table with atomics.  Falls back to v1 when the owner registers would not suffice (M * CU > 12 * block).
#pragma once
#include "satba_kernels.h"

namespace satba {

constexpr int LIN2_MAXACC = 12;

template <int NP>
struct Lin2Cfg {
    static constexpr int CU = cam_acc_len(NP);
    static constexpr int BLOCK = (CU <= 20) ? 512 : 256;  // threads = staging slots per super-tile
    static constexpr int WAVES = BLOCK / 64;
};

struct Lin2Args {
    double2* __restrict__ f;
    double* __restrict__ V;
    double* __restrict__ gp;
    double* __restrict__ part;      // grid x M x CU
    double* __restrict__ overflow;  // M x CU, zeroed before the launch
    double* __restrict__ hdr_cost;
    double* __restrict__ hdr_gpmax;
    int cap;                        // bucket capacity (slots per camera per super-tile)
    int camc_in_lds;
};

template <int MODEL, int NP, bool ROBUST>
__global__ __launch_bounds__(Lin2Cfg<NP>::BLOCK) void k_linearize2(ObsArgs a, Lin2Args s) {
    using Cfg = Lin2Cfg<NP>;
    constexpr int CU = Cfg::CU, BLOCK = Cfg::BLOCK, WAVES = Cfg::WAVES;
    extern __shared__ double s_lds[];
    double* stage = s_lds;                                   // [CU][BLOCK]
    double* s_camc = stage + (size_t)CU * BLOCK;            // [M][CAMC] (if camc_in_lds)
    unsigned* s_cnt = reinterpret_cast<unsigned*>(s_camc + (s.camc_in_lds ? (size_t)a.M * CAMC : 0));  // [2][M]
    unsigned short* s_list = reinterpret_cast<unsigned short*>(s_cnt + 2 * a.M);                        // [M][cap]
    __shared__ unsigned char s_seg[WAVES][66];  // per wave: lane index of the start of each point run, then the end
    __shared__ double s_red[2][WAVES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (s.camc_in_lds)
        for (int i = tid; i < a.M * CAMC; i += BLOCK) s_camc[i] = a.camc[i];
    for (int i = tid; i < 2 * a.M; i += BLOCK) s_cnt[i] = 0;
    __syncthreads();
    const double* cbase = s.camc_in_lds ? s_camc : a.camc;
    const int n_acc = a.M * CU;  // accumulator a = c * CU + v, owned by thread a % BLOCK at position a / BLOCK

    double acc[LIN2_MAXACC];
#pragma unroll
    for (int k = 0; k < LIN2_MAXACC; ++k) acc[k] = 0.0;
    double cost = 0.0, gmax = 0.0;

    const int n_super = (a.n_tiles + WAVES - 1) / WAVES;
    int parity = 0;
    for (int st = blockIdx.x; st < n_super; st += gridDim.x, parity ^= 1) {
        // ------------------------------------------------------------------ phase A
        const int tile = st * WAVES + wave;
        if (tile < a.n_tiles) {
            const int o0 = a.tile_start[tile], o1 = a.tile_start[tile + 1];
            const long long o = (long long)o0 + lane;
            const bool active = o < o1;
            const int slot = wave * 64 + lane;
            int pt = -1 - lane, cam = 0;
            ObsEval<MODEL, NP, true, ROBUST> e;
            if (active) {
                cam = a.cam[o];
                pt = a.pt[o];
                e.eval(a, o, cam, pt, cbase + (size_t)cam * CAMC);
                s.f[o] = make_double2(e.ftrue[0], e.ftrue[1]);
                cost += e.rho;
                double* row = stage + slot;
                row[0 * BLOCK] = e.Jp[0][0] * e.Jp[0][0] + e.Jp[1][0] * e.Jp[1][0];
                row[1 * BLOCK] = e.Jp[0][0] * e.Jp[0][1] + e.Jp[1][0] * e.Jp[1][1];
                row[2 * BLOCK] = e.Jp[0][0] * e.Jp[0][2] + e.Jp[1][0] * e.Jp[1][2];
                row[3 * BLOCK] = e.Jp[0][1] * e.Jp[0][1] + e.Jp[1][1] * e.Jp[1][1];
                row[4 * BLOCK] = e.Jp[0][1] * e.Jp[0][2] + e.Jp[1][1] * e.Jp[1][2];
                row[5 * BLOCK] = e.Jp[0][2] * e.Jp[0][2] + e.Jp[1][2] * e.Jp[1][2];
                row[6 * BLOCK] = e.Jp[0][0] * e.fs[0] + e.Jp[1][0] * e.fs[1];
                row[7 * BLOCK] = e.Jp[0][1] * e.fs[0] + e.Jp[1][1] * e.fs[1];
                row[8 * BLOCK] = e.Jp[0][2] * e.fs[0] + e.Jp[1][2] * e.fs[1];
            }
            // runs of equal point index: table of run starts (wave-local LDS, in-order with the writes above)
            const int prev = __shfl_up(pt, 1);
            const bool head = active && (lane == 0 || prev != pt);
            const unsigned long long heads = __ballot(head);
            const int n_runs = __popcll(heads);
            const int n_active = o1 - o0;
            if (head) s_seg[wave][__popcll(heads & ((1ull << lane) - 1ull))] = (unsigned char)lane;
            if (lane == 0) s_seg[wave][n_runs] = (unsigned char)n_active;
            // lane (q, v): sum value v over run q; 7 runs per pass
            for (int q0 = 0; q0 < n_runs; q0 += 7) {
                const int q = q0 + lane / 9, v = lane % 9;
                if (lane < 63 && q < n_runs) {
                    const int b = s_seg[wave][q], en = s_seg[wave][q + 1];
                    const double* col = stage + (size_t)v * BLOCK + wave * 64;
                    double sum = 0.0;
                    for (int l = b; l < en; ++l) sum += col[l];
                    const int ptq = a.pt[o0 + b];
                    double* dst = (v < 6) ? s.V + 6 * (size_t)ptq + v : s.gp + 3 * (size_t)ptq + (v - 6);
                    if (a.tile_split[tile]) atomicAdd(dst, sum);
                    else *dst = sum;
                    if (v >= 6) gmax = fmax(gmax, fabs(sum));
                }
            }
            // camera products into the same staging rows (program order keeps them behind the reads above)
            if (active) {
                double* row = stage + slot;
                int k = 0;
#pragma unroll
                for (int i = 0; i < NP; ++i)
#pragma unroll
                    for (int j = i; j < NP; ++j) row[(size_t)(k++) * BLOCK] = e.Jc[0][i] * e.Jc[0][j] + e.Jc[1][i] * e.Jc[1][j];
#pragma unroll
                for (int i = 0; i < NP; ++i) row[(size_t)(k++) * BLOCK] = e.Jc[0][i] * e.fs[0] + e.Jc[1][i] * e.fs[1];
                const unsigned pos = atomicAdd(&s_cnt[parity * a.M + cam], 1u);
                if ((int)pos < s.cap) {
                    s_list[(size_t)cam * s.cap + pos] = (unsigned short)slot;
                } else {  // bucket full (rare): global overflow table
                    double* ov = s.overflow + (size_t)cam * CU;
                    for (int kk = 0; kk < CU; ++kk) atomicAdd(ov + kk, row[(size_t)kk * BLOCK]);
                }
            }
        }
        __syncthreads();
        // ------------------------------------------------------------------ phase B
#pragma unroll
        for (int k = 0; k < LIN2_MAXACC; ++k) {
            const int aidx = tid + k * BLOCK;
            if (aidx < n_acc) {
                const int c = aidx / CU, v = aidx - c * CU;
                const int n = min((int)s_cnt[parity * a.M + c], s.cap);
                const double* col = stage + (size_t)v * BLOCK;
                const unsigned short* lst = s_list + (size_t)c * s.cap;
                double t = 0.0;
                for (int i = 0; i < n; ++i) t += col[lst[i]];
                acc[k] += t;
            }
        }
        for (int i = tid; i < a.M; i += BLOCK) s_cnt[(parity ^ 1) * a.M + i] = 0;
        __syncthreads();
    }
    // ---------------------------------------------------------------------- epilogue
    cost = wave_sum(cost);
    gmax = wave_max(gmax);
    if (lane == 0) { s_red[0][wave] = cost; s_red[1][wave] = gmax; }
    __syncthreads();
    if (tid == 0) {
        double c = 0.0, g = 0.0;
        for (int i = 0; i < WAVES; ++i) { c += s_red[0][i]; g = fmax(g, s_red[1][i]); }
        atomicAdd(s.hdr_cost, 0.5 * c);
        atomic_max_pos(s.hdr_gpmax, g);
    }
    double* out = s.part + (size_t)blockIdx.x * n_acc;
#pragma unroll
    for (int k = 0; k < LIN2_MAXACC; ++k) {
        const int aidx = tid + k * BLOCK;
        if (aidx < n_acc) out[aidx] = acc[k];
    }
}

}  // namespace satba
