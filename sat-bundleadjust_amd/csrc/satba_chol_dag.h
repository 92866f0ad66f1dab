// satba_chol_dag.h -- dense Cholesky + forward substitution of the reduced camera system as ONE persistent
// dataflow kernel.
//
// The blocked multi-launch version (satba_chol.h) spends its time in the latency of 64 dependent kernel
// launches (~27 us + ~13 us per 32-column panel at n = 1000: 1.5 ms, constant in the number of GPUs and therefore
// the term that caps multi-GPU scaling).  Here the factorisation is a task graph over 64 x 64 tiles executed by a
// small, fully resident grid of workgroups:
//
//   F(k)      factorise diagonal tile (k, k)                     after all its updates
//   T(i, k)   tile(i, k) <- tile(i, k) L_kk^-T                   after F(k) and the tile's updates
//   U(i,j,k)  tile(i, j) -= L(i, k) L(j, k)^T                    after T(i, k), T(j, k) and U(i, j, k-1)
//   Tb(k) / Ub(j, k)  the same two steps for the right-hand side (forward substitution folded in)
//
// Tasks sit in a host-built list in a topological (right-looking) order; a workgroup draws the next index from
// an atomic counter, waits until the task's inputs are published, executes, publishes.  Because tasks are drawn in
// topological order and every drawn task runs to completion, the earliest unfinished task can always run: no
// deadlock as long as the whole grid is resident (64 workgroups on 256 CUs).  Hand-offs follow the agent-scope
// release / acquire recipe of cdna_hip_programming.md (Guideline 16): stores -> s_waitcnt vmcnt(0) -> barrier ->
// lane 0 release fence -> s_waitcnt -> relaxed flag store; consumers poll with relaxed agent-scope loads, then one
// acquire fence + barrier before plain loads.  Every spin is bounded; on timeout an abort word is raised and all
// workgroups leave (the host reports an error instead of hanging the GPU).
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "satba_chol.h"

namespace satba {

constexpr int DG_T = 64;
constexpr int DG_THREADS = 256;
constexpr int DG_GRID = 64;
constexpr long long DG_SPIN_LIMIT = 1ll << 22;

enum { DG_F = 0, DG_T_ = 1, DG_U = 2, DG_TB = 3, DG_UB = 4 };

struct DagTask { short type, i, j, k; };

struct DagFlags {  // all zeroed before each launch
    int* f_done;   // NT
    int* t_done;   // NT x NT
    int* upd;      // NT x NT: number of U updates applied to tile (i, j)
    int* tb_done;  // NT
    int* updb;     // NT: number of Ub updates applied to block j of the right-hand side
    int* ctr;      // [0] task counter, [1] abort, [2] not-positive-definite
    long long* times;  // optional (tools): per task draw / ready / done timestamps (s_memtime)
};

inline std::vector<DagTask> dag_task_list(int n) {
    const int NT = (n + DG_T - 1) / DG_T;
    std::vector<DagTask> t;
    for (short k = 0; k < NT; ++k) {
        t.push_back({DG_F, k, k, k});
        for (short i = k + 1; i < NT; ++i) t.push_back({DG_T_, i, k, k});
        t.push_back({DG_TB, k, k, k});
        for (short j = k + 1; j < NT; ++j) {  // column k+1 first: it feeds the next panel
            for (short i = j; i < NT; ++i) t.push_back({DG_U, i, j, k});
            t.push_back({DG_UB, j, j, k});
        }
    }
    return t;
}

__device__ inline int dg_load_flag(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// wait until *p >= want (lane 0 polls); returns false on abort / timeout
__device__ inline bool dg_wait(const int* p, int want, int* abort_word) {
    long long spins = 0;
    while (dg_load_flag(p) < want) {
        __builtin_amdgcn_s_sleep(2);
        if ((++spins & 1023) == 0) {
            if (dg_load_flag(abort_word)) return false;
            if (spins > DG_SPIN_LIMIT) {
                __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
    return true;
}

__global__ __launch_bounds__(DG_THREADS) void k_chol_dag(double* __restrict__ A, int n, double* __restrict__ b,
                                                         const DagTask* __restrict__ tasks, int n_tasks, DagFlags fl) {
    __shared__ double sA[DG_T][DG_T + 1];  // F: the tile; T: L_kk transposed; U: L(i,k) transposed
    __shared__ double sB[DG_T][DG_T + 1];  // U: L(j,k) transposed
    __shared__ double s_vec[DG_T];
    __shared__ int s_task, s_ok;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NT = (n + DG_T - 1) / DG_T;

    for (;;) {
        if (tid == 0) s_task = __hip_atomic_fetch_add(fl.ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int t = s_task;
        if (t >= n_tasks) return;
        const long long t_draw = fl.times ? (long long)__builtin_readcyclecounter() : 0;
        const DagTask tk = tasks[t];
        const int i = tk.i, j = tk.j, k = tk.k;
        // ---------------------------------------------------------------- wait for the inputs
        if (tid == 0) {
            bool ok = true;
            switch (tk.type) {
                case DG_F:  ok = dg_wait(fl.upd + k * NT + k, k, fl.ctr + 1); break;
                case DG_T_: ok = dg_wait(fl.f_done + k, 1, fl.ctr + 1) && dg_wait(fl.upd + i * NT + k, k, fl.ctr + 1); break;
                case DG_U:  ok = dg_wait(fl.t_done + i * NT + k, 1, fl.ctr + 1) && dg_wait(fl.t_done + j * NT + k, 1, fl.ctr + 1) &&
                                 dg_wait(fl.upd + i * NT + j, k, fl.ctr + 1); break;
                case DG_TB: ok = dg_wait(fl.f_done + k, 1, fl.ctr + 1) && dg_wait(fl.updb + k, k, fl.ctr + 1); break;
                default:    ok = dg_wait(fl.t_done + j * NT + k, 1, fl.ctr + 1) && dg_wait(fl.tb_done + k, 1, fl.ctr + 1) &&
                                 dg_wait(fl.updb + j, k, fl.ctr + 1); break;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            s_ok = ok ? 1 : 0;
        }
        __syncthreads();
        if (!s_ok) return;
        const long long t_ready = fl.times ? (long long)__builtin_readcyclecounter() : 0;

        const int r0 = i * DG_T, c0 = j * DG_T, k0 = k * DG_T;
        int* publish = nullptr;
        int publish_value = 1;
        if (tk.type == DG_F) {
            // ------------------------------------------------------------ diagonal tile: unblocked Cholesky in LDS
            for (int idx = tid; idx < DG_T * DG_T; idx += DG_THREADS) {
                const int r = idx % DG_T, c = idx / DG_T;
                const bool in = (k0 + r < n) && (k0 + c < n);
                sA[r][c] = (in && r >= c) ? A[(size_t)(k0 + r) + (size_t)(k0 + c) * n] : ((r == c && !in) ? 1.0 : 0.0);
            }
            __syncthreads();
            for (int jj = 0; jj < DG_T; ++jj) {
                if (tid == 0) {
                    double d = sA[jj][jj];
                    if (!(d > 0.0)) { __hip_atomic_store(fl.ctr + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); d = 1.0; }
                    sA[jj][jj] = sqrt(d);
                }
                __syncthreads();
                if (tid > jj && tid < DG_T) sA[tid][jj] /= sA[jj][jj];
                __syncthreads();
                // trailing update: thread = (row r, column group): rows r > jj, columns jj < c <= r in steps of 4
                {
                    const int r = tid & 63;
                    if (r > jj) {
                        const double lr = sA[r][jj];
                        for (int c = jj + 1 + (tid >> 6); c <= r; c += 4) sA[r][c] -= lr * sA[c][jj];
                    }
                }
                __syncthreads();
            }
            for (int idx = tid; idx < DG_T * DG_T; idx += DG_THREADS) {
                const int r = idx % DG_T, c = idx / DG_T;
                if (r >= c && k0 + r < n) A[(size_t)(k0 + r) + (size_t)(k0 + c) * n] = sA[r][c];
            }
            publish = fl.f_done + k;
        } else if (tk.type == DG_T_) {
            // ------------------------------------------------------------ panel tile: X L_kk^T = P, 4 threads per row
            for (int idx = tid; idx < DG_T * DG_T; idx += DG_THREADS) {
                const int c = idx % DG_T, m = idx / DG_T;  // sA[m][c] = L[c][m] (c >= m), diagonal holds 1 / L[m][m]
                double v = 0.0;
                if (k0 + c < n && c >= m) v = A[(size_t)(k0 + c) + (size_t)(k0 + m) * n];
                if (c == m) v = (k0 + m < n) ? 1.0 / v : 1.0;
                sA[m][c] = v;
            }
            __syncthreads();
            const int row = tid >> 2, part = tid & 3;  // thread owns columns part, part + 4, ...
            const int r = r0 + row;
            double x[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int c = part + 4 * q;
                x[q] = (r < n && k0 + c < n) ? A[(size_t)r + (size_t)(k0 + c) * n] : 0.0;
            }
#pragma unroll
            for (int m = 0; m < DG_T; ++m) {
                double xm = x[m >> 2] * sA[m][m];
                xm = __shfl(xm, (lane & ~3) | (m & 3));  // the owner of column m broadcasts to its row's 4 threads
                if (part == (m & 3)) x[m >> 2] = xm;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int c = part + 4 * q;
                    if (c > m) x[q] -= xm * sA[m][c];
                }
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int c = part + 4 * q;
                if (r < n && k0 + c < n) A[(size_t)r + (size_t)(k0 + c) * n] = x[q];
            }
            publish = fl.t_done + i * NT + k;
        } else if (tk.type == DG_U) {
            // ------------------------------------------------------------ tile(i, j) -= L(i, k) L(j, k)^T
            for (int idx = tid; idx < DG_T * DG_T; idx += DG_THREADS) {
                const int r = idx % DG_T, kk = idx / DG_T;
                sA[kk][r] = (r0 + r < n && k0 + kk < n) ? A[(size_t)(r0 + r) + (size_t)(k0 + kk) * n] : 0.0;
                sB[kk][r] = (c0 + r < n && k0 + kk < n) ? A[(size_t)(c0 + r) + (size_t)(k0 + kk) * n] : 0.0;
            }
            __syncthreads();
            const int tr = (tid & 15) * 4, tc = (tid >> 4) * 4;
            double acc[4][4] = {};
#pragma unroll 8
            for (int kk = 0; kk < DG_T; ++kk) {
                double av[4], bv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { av[q] = sA[kk][tr + q]; bv[q] = sB[kk][tc + q]; }
#pragma unroll
                for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) acc[cc][rr] += av[rr] * bv[cc];
            }
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int r = r0 + tr + rr, c = c0 + tc + cc;
                    if (r < n && c < n && r >= c) A[(size_t)r + (size_t)c * n] -= acc[cc][rr];
                }
            publish = fl.upd + i * NT + j;
            publish_value = k + 1;
        } else if (tk.type == DG_TB) {
            // ------------------------------------------------------------ y_k = L_kk^-1 b_k (wave 0, lane = row)
            if (wave == 0) {
                const int r = k0 + lane;
                double v = (r < n) ? b[r] : 0.0;
                for (int m = 0; m < DG_T; ++m) {
                    const double lmm = (k0 + m < n) ? A[(size_t)(k0 + m) + (size_t)(k0 + m) * n] : 1.0;
                    const double ym = __shfl(v, m) / lmm;
                    const double lrm = (lane > m && r < n && k0 + m < n) ? A[(size_t)r + (size_t)(k0 + m) * n] : 0.0;
                    if (lane == m) v = ym;
                    else if (lane > m) v -= lrm * ym;
                }
                if (r < n) b[r] = v;
            }
            publish = fl.tb_done + k;
        } else {
            // ------------------------------------------------------------ b_j -= L(j, k) y_k
            if (tid < DG_T) s_vec[tid] = (k0 + tid < n) ? b[k0 + tid] : 0.0;
            __syncthreads();
            if (tid < DG_T && c0 + tid < n) {
                const int r = c0 + tid;
                double s = 0.0;
                for (int m = 0; m < DG_T; ++m)
                    if (k0 + m < n) s += A[(size_t)r + (size_t)(k0 + m) * n] * s_vec[m];
                b[r] -= s;
            }
            publish = fl.updb + j;
            publish_value = k + 1;
        }
        // ---------------------------------------------------------------- publish
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(publish, publish_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (fl.times) {
                fl.times[4 * (size_t)t + 0] = t_draw;
                fl.times[4 * (size_t)t + 1] = t_ready;
                fl.times[4 * (size_t)t + 2] = (long long)__builtin_readcyclecounter();
                fl.times[4 * (size_t)t + 3] = ((long long)tk.type << 48) | ((long long)tk.i << 32) | ((long long)tk.j << 16) | tk.k;
            }
        }
    }
}

// fold the kernel's status words into the solver's failure flag: bit 0 not positive definite, bit 1 aborted
__global__ void k_dag_status(const int* __restrict__ ctr, int* __restrict__ fail) {
    if (ctr[2]) atomicOr(fail, 1);
    if (ctr[1]) atomicOr(fail, 2);
}

struct DagWorkspace {
    DagTask* d_tasks = nullptr;
    int n_tasks = 0;
    int* d_flags = nullptr;  // one allocation, carved into DagFlags
    size_t flag_ints = 0;
    int NT = 0;
    long long* d_times = nullptr;  // only with SATBA_DAG_TIMES (tools/dag_times.py)
};

inline DagFlags dag_flags(const DagWorkspace& w) {
    DagFlags f;
    int* p = w.d_flags;
    f.f_done = p; p += w.NT;
    f.t_done = p; p += w.NT * w.NT;
    f.upd = p; p += w.NT * w.NT;
    f.tb_done = p; p += w.NT;
    f.updb = p; p += w.NT;
    f.ctr = p;
    f.times = w.d_times;
    return f;
}

inline size_t dag_flag_ints(int NT) { return (size_t)2 * NT * NT + 3 * NT + 4; }

}  // namespace satba
