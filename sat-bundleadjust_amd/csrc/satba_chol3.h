// satba_chol3.h -- the dense Cholesky of the reduced camera system as ONE persistent launch (round 4).
//
// Replaces the launch-per-double-step factorisation of rounds 1-3 (16 launches of 23.7 us at 1 000 unknowns: launch gap, cold
// start, trailing update of the diagonal tile, two 32-column diagonal blocks and a panel solve behind a flag -- all on the
// chain).  Here the matrix is cut into 64 x 64 tiles and every workgroup is one of two things:
//
//   the CHAIN workgroup (ticket 0)   walks down the diagonal.  At step k it holds the diagonal tile D = A'(k,k) and the tile
//       below it R = A'(k+1,k), both with every earlier panel applied, one row per lane in registers.  Twelve waves -- three
//       row sets (D, R and the identity, whose rows end up as L_kk^-T) x four column quarters -- run a right-looking
//       factorisation in micro-panels of 8 columns: the owner of a micro-panel factorises (D) or solves (R, identity) its 8
//       columns in registers, puts them into LDS and raises an LDS flag; the waves to its right apply the rank-8 update.  The
//       serial chain sees 8 dependent columns + one hand-over + one rank-8 update per micro-panel and nothing else: no launch
//       gap, no global memory.  Four more waves form the next diagonal tile A'(k+1,k+1) - R R^T by fp64 MFMA while R is being
//       solved (rank-8 updates as its micro-panels appear) and hand it over through LDS, carry the right-hand side
//       (y_k = L_kk^-1 b_k: the forward substitution is folded in) and publish.
//   OWNER workgroups (tickets 1..)   hold ONE tile (i, j) in MFMA accumulators for its whole life: apply panels m = 0 .. as the
//       tiles L(i,m), L(j,m) are published, then (i >= j + 2) multiply by L_jj^-T -- the panel solve is a GEMM -- and publish
//       L(i,j) and its share L(i,j) y_j of the forward substitution; the tiles on and directly below the diagonal are handed
//       to the chain with all panels but the last applied.
//
// Everything a workgroup reads from another one is published with write-through (sc1) stores, a drain and a flag word
// (guide: Guideline 16, form R1 with 8-byte agent-scope accesses on both sides); flags hold 4 x epoch + stage, the epoch counts
// launches, so nothing has to be cleared between launches.  Tasks are handed out by a ticket counter in column-major tile order:
// a workgroup only ever waits for tiles with smaller tickets (or for the chain, ticket 0), so the kernel makes progress whatever
// part of the grid is resident.  Every spin is bounded (fail |= 2 after ~1 s).
#pragma once
#include <type_traits>

namespace satba {

constexpr int C3_RS = 10;                  // row stride (doubles) of a micro-panel row in LDS: 20 banks, conflict-free b128 row reads
constexpr int C3_BLK = 64 * C3_RS + 2;     // one (row set, micro-panel) block; + 2: the 8 blocks of a set start 4 banks apart
constexpr int C3_TBS = 65;                 // column stride of the hand-over buffer of the next diagonal tile
constexpr int C3_LD = 80;                  // row stride of an owner's operand tiles [k][r] (MFMA operand loads: disjoint bank ranges)
constexpr int C3_S1 = 1, C3_S2 = 2;        // stages of a tile flag: 1 = A' (all panels but the chain's) published, 2 = L published
constexpr int C3_SPIN_LIMIT = 1 << 22;
constexpr long long C3_ARRIVE_TIMEOUT = 1000000;  // default wait for the producers' counters, in ticks of the 100 MHz wall clock (10 ms; C3Args::arr_timeout)
constexpr int C3_TS = 32;                  // time stamps per step (tools): 0..5 phases, 8 + 8 set + p: micro-panel p of a row set flagged

struct C3Args {
    double* A;        // n x n, column-major, lower triangle valid; L in place on return (and L^T in the strict upper triangle if mirror)
    int n;
    double* b;        // right-hand side; y = L^-1 b on return
    int* fail;        // |= 1 not positive definite, |= 2 a wait timed out, |= 4 (with 2) it was a wait for the kernel that produces the matrix (c3_wait_arrive)
    int* flags;       // T x T tile flags (never cleared: epochs)
    int epoch;        // > 0, larger at every launch
    double* Linv;     // T x 64 x 64: column-major inverses of the diagonal blocks
    double* Cc;       // T x T x 64: L(i,j) y_j
    int* ctr;         // [0] ticket counter, [1] workgroups done; zero on entry, zero on exit
    double* dinv;     // 32 x 32 inverses of the diagonal blocks, [blk][r][c], for k_trsv_back_mw (or null)
    long long* ts;    // tools: C3_TS wall-clock stamps per step (or null)
    int mirror;       // also store L^T into the strict upper triangle (k_trsv_back_mw reads it)
    // The launch runs BESIDE the kernel that produces the matrix (k_schur_pairs, SchurArgs::arrive) when `arrive` is set: a tile is
    // read only when the producers of its columns have counted themselves in -- word SCHUR_ARRIVE_STRIDE * c of `arrive` reaches
    // arr_M - 1 - c for every camera c (np unknowns each) with a column in the tile, word SCHUR_ARRIVE_STRIDE * arr_M holds
    // arr_epoch -- and is scaled as it is read, A[r][c] / (si[r] si[c]), b = rhs / si (k_scale_system's arithmetic); the last
    // workgroup leaves the counters at zero.  Producers publish with write-through stores; everything that is input is read with
    // agent-scope loads.
    int* arrive = nullptr;
    int arr_M = 0, np = 1, arr_epoch = 0, nap = 1;  // nap: length of a pause between two looks at the counters, in units of ~0.5 us
    // how long a tile waits for its producers before it gives up (fail |= 2: the two kernels are not running at the same time), in
    // ticks of the 100 MHz wall clock.  Round 4 counted spins (~0.3 - 1 s); the caller now passes a few milliseconds plus a
    // multiple of the pair kernel's expected duration, so that a fall-back to the sequential front costs ~10 ms
    long long arr_timeout = C3_ARRIVE_TIMEOUT;
    // arr_extra = 1: the producer also writes the DIAGONAL block of camera c and its entries of the right-hand side, and counts that in
    // (M - c counts for camera c instead of M - 1 - c): the right-hand side of a tile row is complete when the producers of the
    // tile's own columns have counted in, and the chain takes it tile by tile (c3_chain_aux) instead of all at once at its start
    int arr_extra = 0;
    const double* si = nullptr;
    const double* rhs = nullptr;
};
constexpr int C3_ARRIVE_STRIDE = 32;  // (= SCHUR_ARRIVE_STRIDE)

// The owner and mirror tasks are functions of their own (their registers are allocated apart from the chain's roles) and need most of the
// argument block.  Passed by value it was copied to the stack in front of every call -- 576 of the kernel's 744 bytes of scratch per lane
// in round 5 --; they re-read it from the kernel-argument segment instead (scalar loads: C3Args is k_chol_tiles' FIRST parameter, offset 0).
typedef const __attribute__((address_space(4))) C3Args* c3_kargs;
// (q: __builtin_amdgcn_kernarg_segment_ptr() taken IN THE KERNEL -- inside a callee the intrinsic folds to a null pointer -- and handed down;
// the callee makes it uniform again, so that the reads are scalar loads)
__device__ __forceinline__ C3Args c3_args_from_kernarg(c3_kargs q) {
    {
        const unsigned long long v = (unsigned long long)q;
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
        q = (c3_kargs)(((unsigned long long)hi << 32) | lo);
    }
    C3Args g;
    g.A = q->A; g.n = q->n; g.b = q->b; g.fail = q->fail; g.flags = q->flags; g.epoch = q->epoch; g.Linv = q->Linv; g.Cc = q->Cc; g.ctr = q->ctr;
    g.dinv = q->dinv; g.ts = q->ts; g.mirror = q->mirror; g.arrive = q->arrive; g.arr_M = q->arr_M; g.np = q->np; g.arr_epoch = q->arr_epoch;
    g.nap = q->nap; g.arr_timeout = q->arr_timeout; g.arr_extra = q->arr_extra; g.si = q->si; g.rhs = q->rhs;
    return g;
}

struct C3Arrive { const int* arrive; int M, np, epoch, nap; long long timeout; int extra; };

constexpr size_t c3_lds_bytes() {
    return sizeof(double) * (3 * 8 * C3_BLK + 64 * C3_TBS + 64 + 64 + 128 + 512) + sizeof(int) * 64;
}

// time stamps inside the kernel are compiled in for tools/chol only (-DC3_STAMPS): their pointers are loop invariants the
// compiler keeps in registers, which pushed LDS addresses of the riders' loop into scratch
#ifdef C3_STAMPS
#define C3_STAMP(ts, idx, cond) do { if ((ts) && (cond)) { long long* t_ = (ts); asm volatile("" : "+v"(t_)); *(c3_gll*)(t_ + (idx)) = wall_clock64(); } } while (0)
#else
#define C3_STAMP(ts, idx, cond) do { } while (0)
#endif

#ifdef C3_STAMPS_IWAVE  // (experiment: the time line of wave 3 of the identity at the end of a step, in the slots of the diagonal tile's micro-panels)
#define C3_STAMPI(ts, i, cond) do { if ((ts) && (cond)) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); long long* t_ = (ts); asm volatile("" : "+v"(t_)); *(c3_gll*)(t_ + (i)) = wall_clock64(); } } while (0)
#else
#define C3_STAMPI(ts, i, cond) do { } while (0)
#endif

// nothing moves across: neither memory operations nor arithmetic (an empty asm with a memory clobber alone lets the scheduler sink
// the multiply-adds below it: every broadcast of an update was then in flight at once -- kilobytes of spills)
#define C3_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
// ... and register-only instructions are ordered against an asm statement only through its operands: the values named here are
// complete before the fence, the loads behind it start after it
#define C3_PIN4(w, x, y, z) do { asm volatile("" : "+v"(w), "+v"(x), "+v"(y), "+v"(z) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define C3_PIN2(w, x) do { asm volatile("" : "+v"(w), "+v"(x) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define C3_PIN1(w) do { asm volatile("" : "+v"(w) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

// Global memory is addressed through address-space-1 pointers: the role functions are not inlined, their pointer arguments are
// generic, and generic accesses become flat_* instructions, which count on lgkmcnt as well -- every wait for an LDS read then also
// waits for the outstanding global accesses (the identity's waves took 15 us for 32 LDS read -> store pairs that way).
typedef __attribute__((address_space(1))) double c3_gdouble;
typedef __attribute__((address_space(1))) int c3_gint;
typedef __attribute__((address_space(1))) long long c3_gll;
__device__ __forceinline__ double c3_ld(const double* p) { return __hip_atomic_load((const c3_gdouble*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double c3_gld(const double* p) { return *(const c3_gdouble*)p; }       // plain global load
__device__ __forceinline__ void c3_gst(double* p, double v) { *(c3_gdouble*)p = v; }               // plain global store
__device__ __forceinline__ int c3_ld_flag(const int* p) { return __hip_atomic_load((const c3_gint*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void c3_st_flag(int* p, int v) { __hip_atomic_store((c3_gint*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Element idx of a (uniform) array as base + zero-extended 32-bit byte offset: the form a global access takes its base from scalar
// registers in (one vector register of address instead of two, and no 64-bit vector arithmetic); the reduced system has at most
// 6 000 x 6 000 entries, 288 MB: byte offsets fit 32 bits
template <class T>
__device__ __forceinline__ T* c3_at(T* base, unsigned idx) {
    return reinterpret_cast<T*>(reinterpret_cast<char*>(const_cast<typename std::remove_const<T>::type*>(base)) + (size_t)(idx * (unsigned)sizeof(T)));
}
// element (row, col) of the n x n matrix if `ok`, else `other`: the load itself is unconditional (clamped address) -- a load under a
// condition gets its own s_waitcnt, and sixteen of them in a row were ten microseconds on the chain's critical hand-over
__device__ __forceinline__ double c3_ld_at(const double* A, int n, int row, int col, bool ok, double other = 0.0) {
    const int r = row < n ? row : n - 1, c = col < n ? col : n - 1;
    const double v = c3_ld(c3_at(A, (unsigned)(r + c * n)));
    return ok ? v : other;
}
__device__ __forceinline__ void c3_st(double* p, double v) { __hip_atomic_store((c3_gdouble*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// status word: a global (not flat) atomic -- a flat instruction in a role function makes every later write-through store of that function wait
// for all outstanding memory operations (round 6: sixteen serialised stores in the diagonal tile's publication)
__device__ __forceinline__ void c3_or_fail(int* fail, int bits) { __hip_atomic_fetch_or((c3_gint*)fail, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Arguments of the (non-inlined) role functions arrive in vector registers, whatever the caller knows about them: a value that is the
// same in every lane is moved to scalar registers once, and the address arithmetic built on it stays off the vector file (round 6: the
// riders kept 18 registers of uniform arguments and 24 spilled LDS addresses -- two scratch reloads in front of every operand read)
__device__ __forceinline__ int c3_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <class T>
__device__ __forceinline__ T* c3_uni(T* q) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(q);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}

// How long a wait inside k_chol_tiles may last, in polls: slots 62 (waits on global flag words, ~0.3 us per poll) and 63 (waits on LDS
// words, ~0.06 us per poll) of the kernel's LDS words, written by thread 0 at the kernel's start -- about a second, plus the arrival
// time-out when the launch waits for its input (C3Args::arrive): every wait for another tile or role is then, transitively, a wait
// for a message or a producer (round 6: the fixed poll counts of round 4, ~1 s, cut a factorisation short whose messages came
// through a slow collective).  (Counted in polls, not on the wall clock: a 64-bit start time alive across every inlined wait loop
// cost the chain's roles 160 - 210 bytes of scratch.)
constexpr int C3_LDS_INTS_AT = 3 * 8 * C3_BLK + 64 * C3_TBS + 64 + 64 + 128 + 512;  // (doubles in front of the LDS words: c3_carve, c3_lds_bytes)
__device__ __forceinline__ int c3_poll_limit(int which) {
    extern __shared__ __attribute__((aligned(16))) double c3_lds[];
    return __hip_atomic_load(reinterpret_cast<const int*>(c3_lds + C3_LDS_INTS_AT) + 62 + which, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// wait until a global flag word reaches `want` (wrap-safe); false after the time-out or when another wait has timed out
__device__ __forceinline__ bool c3_wait(const int* f, int want, int* fail) {
    int spins = 0;
    while ((int)(c3_ld_flag(f) - want) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 1023) == 0) {
            if (spins > c3_poll_limit(0)) { c3_or_fail(fail, 2); return false; }
            if (c3_ld_flag(fail) & 2) return false;
        }
    }
    return true;
}
// (LDS words: workgroup-scope atomics on plain pointers -- a volatile access through a pointer the compiler cannot prove to be LDS
// becomes a flat load)
__device__ __forceinline__ int c3_lds_get(const int* f) { return __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void c3_lds_set(int* f, int v) { __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void c3_lds_inc(int* f) { __hip_atomic_fetch_add(f, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// kernel_limit = false: k_solve_small (no slot of its own: the fixed count)
__device__ __forceinline__ bool c3_wait_lds(const int* f, int want, int* fail, bool kernel_limit = true) {
    int spins = 0;
    while (c3_lds_get(f) < want) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 4095) == 0) {
            if (spins > (kernel_limit ? c3_poll_limit(1) : 4 * C3_SPIN_LIMIT)) { c3_or_fail(fail, 2); return false; }
            if (c3_ld_flag(fail) & 2) return false;
        }
    }
    asm volatile("" ::: "memory");
    return true;
}

// one wave (all 64 lanes call): until the producers of columns col_lo .. col_hi of the matrix have all counted themselves in
__device__ __forceinline__ bool c3_wait_arrive(const C3Arrive& r, int col_lo, int col_hi, int* fail) {
    const int lane = threadIdx.x & 63;
    const int c_lo = col_lo / r.np, c_hi = col_hi / r.np;
    int spins = 0;
    const long long t0 = wall_clock64();
    for (;;) {
        bool ok = true;
        // lane 0: the word behind the last camera; lanes 1 ..: one camera each (64-column tiles: at most 22 with 3 unknowns per camera)
        for (int c = c_lo + lane - 1; c <= c_hi; c += 63) {
            if (lane == 0) { ok = (int)(c3_ld_flag(r.arrive + (size_t)C3_ARRIVE_STRIDE * r.M) - r.epoch) >= 0; break; }
            if (c3_ld_flag(r.arrive + (size_t)C3_ARRIVE_STRIDE * c) < r.M - 1 - c + r.extra) ok = false;
        }
        if (__all(ok)) return true;
        for (int t = 0; t < r.nap; ++t) __builtin_amdgcn_s_sleep(20);
        if ((++spins & 31) == 0) {
            if (wall_clock64() - t0 > r.timeout) { if (lane == 0) c3_or_fail(fail, 2 | 4); return false; }  // bit 2: the wait was for the PRODUCING kernel
            if (c3_ld_flag(fail) & 2) return false;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- chain
// One step of a wave of the diagonal tile (column quarter q): a[16] = its 16 columns of its row (lane), see the file comment.
// pan: the LDS blocks of the diagonal tile's micro-panels ([p][row][C3_RS]); lf their flags (value: step + 1).
// The loop over the pairs of micro-panels is a run-time loop on purpose: unrolled over all eight micro-panels the row sets were
// 110 KB of straight-line code executed once per step -- more than the instruction cache holds -- and a step took 54 us.
// Tried and dropped: the wave's 16 columns factorised entirely in registers (column j's multipliers by v_readlane applied to all
// 15 - j later columns, the last foreign micro-panel's share of columns 8 .. 15 issued inside the factorisation): the registers
// ran out, five spills per column sat on the chain -- 7.6 us per pair of micro-panels against 3.7.
// n_mp: micro-panels to factorise (8: the whole tile; k_solve_small: only those that hold columns of the system)
__device__ __forceinline__ bool c3_panel(double (&a)[16], int q, int lane, double* pan, double* pinv, int* lf, int step1, int* fail, long long* tsp, int n_mp = 8,
                                         bool kernel_limit = true) {
    bool bad = false;
    if (2 * q >= n_mp) return false;
    for (int pp = 0; pp < q; ++pp) {
        // ---- the two micro-panels of the waves to my left: rank-8 updates of my 16 columns, as they appear
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int p = 2 * pp + h;
            if (!c3_wait_lds(lf + p, step1, fail, kernel_limit)) return true;
            const double2* row = reinterpret_cast<const double2*>(pan + p * C3_BLK + lane * C3_RS);
            double r[8];
#pragma unroll
            for (int m = 0; m < 4; ++m) { const double2 t = row[m]; r[2 * m] = t.x; r[2 * m + 1] = t.y; }
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const double2* lc = reinterpret_cast<const double2*>(pan + p * C3_BLK + (16 * q + c) * C3_RS);
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const double2 t = lc[m];
                    s0 = fma(r[2 * m], t.x, s0);
                    s1 = fma(r[2 * m + 1], t.y, s1);
                }
                a[c] -= s0 + s1;
                if ((c & 3) == 3) C3_PIN4(a[c - 3], a[c - 2], a[c - 1], a[c]);  // at most four columns' broadcasts in flight (registers)
            }
        }
    }
    // ---- my own two micro-panels: the pivot chain.  The other three waves of the tile share this SIMD and are busy with the updates
    // of their own columns; with a higher priority the chain's dependent operations are issued as soon as they are ready
    __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int p = 2 * q + h;
        if (p >= n_mp) break;
        double inv_[8];  // 1 / L_jj (wave-uniform; written once per micro-panel: a store under `lane == 0` per column cost the chain 60 %)
        double d = readlane_f64(a[8 * h], 8 * p);
        bad |= !(d > 1e-300) || !(d < 1e300);
        double hh = half_rsqrt(d);
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const double a2 = a[8 * h + jj] + a[8 * h + jj];
            const double l = a2 * hh;  // lane of the pivot: 2 d h = sqrt(d)
            a[8 * h + jj] = l;
            inv_[jj] = hh + hh;
            if (jj + 1 < 8) {
                const double piv = fma(-l, l, a[8 * h + jj + 1]);
                d = readlane_f64(piv, 8 * p + jj + 1);
                bad |= !(d > 1e-300) || !(d < 1e300);
                hh = half_rsqrt(d);
#pragma unroll
                for (int c = jj + 1; c < 8; ++c) a[8 * h + c] = fma(-l, readlane_f64(l, 8 * p + c), a[8 * h + c]);
            }
        }
        double2* row = reinterpret_cast<double2*>(pan + p * C3_BLK + lane * C3_RS);
#pragma unroll
        for (int m = 0; m < 4; ++m) row[m] = make_double2(a[8 * h + 2 * m], a[8 * h + 2 * m + 1]);
        if (lane == 0) {
            double2* pi = reinterpret_cast<double2*>(pinv + 8 * p);
#pragma unroll
            for (int m = 0; m < 4; ++m) pi[m] = make_double2(inv_[2 * m], inv_[2 * m + 1]);
        }
        asm volatile("" ::: "memory");
        if (lane == 0) c3_lds_set(lf + p, step1);  // the LDS unit executes a wave's operations in order: the data are in place before the flag
        C3_STAMP(tsp, p, lane == 0);
        if (h == 0 && p + 1 < n_mp) {  // the micro-panel's share of my columns 8 .. 15
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const double2* lc = reinterpret_cast<const double2*>(pan + p * C3_BLK + (16 * q + 8 + c) * C3_RS);
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const double2 t = lc[m];
                    s0 = fma(a[2 * m], t.x, s0);
                    s1 = fma(a[2 * m + 1], t.y, s1);
                }
                a[8 + c] -= s0 + s1;
                if ((c & 3) == 3) C3_PIN4(a[8 + c - 3], a[8 + c - 2], a[8 + c - 1], a[8 + c]);
            }
        }
    }
    __builtin_amdgcn_s_setprio(0);
    return bad;
}

// LDS of the chain workgroup
struct C3Lds {
    double* pan;            // [3][8][C3_BLK]: D | R | identity, micro-panel blocks [row][C3_RS]
    double* Tb;             // [64][C3_TBS]: next diagonal tile, [column][row]
    double* pinv;           // 1 / L_jj of the diagonal tile
    double* yv;             // y_k
    double* bcur;           // [2][64]: right-hand side of the current / next diagonal tile, all panels applied
    double* inv8;           // [8][8][8]: inverses of the 8 x 8 diagonal blocks of the micro-panels, [p][row][column]
    int* lf;                // [0..7] micro-panel p of D published (value: step + 1); [8..15], [16..23]: waves of R / the identity
                            // that have published micro-panel p (4 per step)
    int* tb_cnt;            // += 1 per wave of the next diagonal tile (two) and step: its blocks are in Tb
    int* pub_cnt;           // += 1 per storing wave of L_kk^-1, y_k and step (4 + 1)
    int* pubD_cnt;          // += 1 per storing wave of L_kk and step (4): a counter of its own -- waves of different roles reach a step's
                            // publication at different times, and a count shared between roles could be completed by the early ones of the next step
    int* pubR_cnt;          // += 1 per storing wave of R and step (4)
    int* y_done;            // = step + 1 when y of the step is formed (the identity's rows have been read by its wave)
    int* prod_cnt;          // += 1 per wave of R and step: its share of the next input is formed (R's rows in LDS may go)
    int* b_rdy;             // = step + 1 when bcur[step & 1] is ready
    int* inv_flag;          // [8]: inv8[p] is ready (value: step + 1)
};
__device__ __forceinline__ C3Lds c3_carve(double* lds) {
    C3Lds l;
    l.pan = lds;
    l.Tb = l.pan + 3 * 8 * C3_BLK;
    l.pinv = l.Tb + 64 * C3_TBS;
    l.yv = l.pinv + 64;
    l.bcur = l.yv + 64;
    l.inv8 = l.bcur + 128;
    int* li = reinterpret_cast<int*>(l.inv8 + 512);
    l.lf = li; l.tb_cnt = li + 24; l.pub_cnt = li + 25; l.b_rdy = li + 26; l.pubR_cnt = li + 27; l.prod_cnt = li + 28; l.y_done = li + 29; l.pubD_cnt = li + 30; l.inv_flag = li + 32;
    return l;
}

// L^T of a step's tiles into the strict upper triangle (for the back-substitution kernel): row r of L is one wave store of 64
// consecutive addresses, read from the micro-panel blocks in LDS (lane = column); rows 16 w .. 16 w + 15 of the diagonal tile and of R
// per call, by the two waves that form the next diagonal tile, behind its hand-over.  All rows are read before the first store and
// counted: the next step may reuse the blocks while the stores go out.  (History: as 8-byte stores scattered over 64 lines by the
// waves that hold the tiles, in front of their drains: +25 us per factorisation; in the riders' function, an LDS address reloaded from
// scratch per row, whose s_waitcnt vmcnt(0) waited for the previous store: 11 us per step.)
// Next diagonal tile, A'(k+1,k+1) - R R^T: its ten lower 16 x 16 blocks by two of the helper waves, five blocks each, as rank-8
// updates (two MFMAs per block) when a micro-panel of R appears in LDS -- ten MFMAs per micro-panel and wave, well inside the 1.6 us
// a micro-panel takes, so that after R's last micro-panel only ten more are left.  The accumulators start from the base -- the tile
// with every earlier panel applied, from its owner, which is about a step ahead.
// (History: in one function with other roles, or without scheduling fences, the compiler had every operand of a step in flight and
// spilled around each MFMA -- the tile was ready 6 to 12 us after R; one shot after R over the idle waves of the diagonal tile:
// their share arrived 4 us after the helpers'.)
template <int Q>
__device__ __forceinline__ bool c3_dnext_blocks(const C3Lds& l, const double* A, int n, int T, int k, int* fail, const int* flags, int want1, int lane) {
    constexpr int RB[2][5] = {{0, 1, 1, 2, 2}, {2, 3, 3, 3, 3}};
    constexpr int CB[2][5] = {{0, 0, 1, 0, 1}, {2, 0, 1, 2, 3}};
    const int e16 = lane & 15, g4 = lane >> 4, r0 = 64 * k, step1 = k + 1;
    const double* panR = l.pan + 8 * C3_BLK;
    if (k + 1 >= 2 && !c3_wait(flags + (k + 1) * T + (k + 1), want1, fail)) return false;
    chol_d4 dn[5];  // (the base in the accumulators: held beside them it did not fit the 128 registers of a 16-wave workgroup)
#pragma unroll
    for (int t = 0; t < 5; ++t) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int rr = 16 * RB[Q][t] + e16, cc = 16 * CB[Q][t] + g4 + 4 * reg;
            dn[t][reg] = c3_ld_at(A, n, r0 + 64 + rr, r0 + 64 + cc, r0 + 64 + rr < n && cc <= rr, (rr == cc) ? 1.0 : 0.0);  // identity padding
        }
    }
    for (int p = 0; p < 8; ++p) {
        if (!c3_wait_lds(l.lf + 8 + p, 4 * step1, fail)) return false;
        // the micro-panel's rows of the four row blocks, as MFMA operands (k = 4 ks + lane / 16)
        double op[4][2];
#pragma unroll
        for (int b4 = 0; b4 < 4; ++b4)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) op[b4][ks] = panR[p * C3_BLK + (16 * b4 + e16) * C3_RS + 4 * ks + g4];
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            dn[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[CB[Q][t]][0], -op[RB[Q][t]][0], dn[t], 0, 0, 0);
            dn[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[CB[Q][t]][1], -op[RB[Q][t]][1], dn[t], 0, 0, 0);
        }
        C3_FENCE();
    }
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) l.Tb[(16 * CB[Q][t] + g4 + 4 * reg) * C3_TBS + 16 * RB[Q][t] + e16] = dn[t][reg];
    asm volatile("" ::: "memory");
    if (lane == 0) c3_lds_inc(l.tb_cnt);
    return true;
}

// the four waves of the diagonal tile: one row per lane, column quarter q (c3_panel)
__device__ __noinline__ void c3_chain_diag(int q, double* A, int n, int T, int* fail, int mirror, long long* ts) {
    extern __shared__ __attribute__((aligned(16))) double c3_lds[];  // (declared here, not passed: the pointers stay in the LDS address space)
    // (q stays in a vector register: with the column quarter in scalar registers every write-through store of L_kk below got an
    // s_waitcnt vmcnt(0) lgkmcnt(0) of its own -- sixteen serialised stores in front of the step's publication)
    A = c3_uni(A); n = c3_uni(n); T = c3_uni(T); fail = c3_uni(fail); mirror = c3_uni(mirror); ts = c3_uni(ts);
    const int tid = threadIdx.x, lane = tid & 63;
    const C3Lds l = c3_carve(c3_lds);
    double a[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int col = 16 * q + c;
        double v = (col == lane) ? 1.0 : 0.0;
        if (lane < n && col <= lane) v = c3_ld(A + (size_t)lane + (size_t)col * n);
        a[c] = v;
    }
    for (int k = 0; k < T; ++k) {
        const int step1 = k + 1;
        const int r0 = 64 * k;
        const bool has_r = k + 1 < T;
        // the identity's riders still read the micro-panels of the previous step (R's are done: the hand-over buffer is complete)
        if (k > 0 && !c3_wait_lds(l.lf + 16 + 7, 4 * k, fail)) return;
        C3_STAMP(ts, k * C3_TS + 0, tid == 0);
        const bool bad = c3_panel(a, q, lane, l.pan, l.pinv, l.lf, step1, fail, ts ? ts + k * C3_TS + 8 : nullptr);
        if (bad && lane == 0) c3_or_fail(fail, 1);
        C3_STAMP(ts, k * C3_TS + 1, lane == 0 && q == 3);
        // ---- my 16 columns of L_kk, write-through and counted for the step's publication: the only reader in this launch is the
        // workgroup that writes L^T of the step's tiles for the back-substitution kernel (c3_mirror_task); the drain falls into the
        // wait for the next diagonal tile
        asm volatile("" : "+v"(n), "+v"(A));  // (keeps the address arithmetic of the stores out of the registers of the factorisation)
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int row = r0 + lane, col = r0 + 16 * q + c;
            if (row < n && col <= row) c3_st(c3_at(A, (unsigned)(row + col * n)), a[c]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // (counted for THIS step only when every wave has counted the previous one: the counters are cumulative, and a wave that has
        // run ahead into step k + 1 must not complete step k's count while a slower one is still storing -- round 6: with the stores
        // of this function serialised by a flat instruction, the mirror task read L_kk's last columns before they had landed)
        if (k > 0 && !c3_wait_lds(l.pubD_cnt, 4 * k, fail)) return;
        if (lane == 0) c3_lds_inc(l.pubD_cnt);
        // ---- the next diagonal tile, formed by two of the helper waves (c3_dnext_blocks), from LDS, one row per lane
        if (has_r) {
            if (!c3_wait_lds(l.tb_cnt, 2 * step1, fail)) return;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int col = 16 * q + c;
                a[c] = (col <= lane) ? l.Tb[col * C3_TBS + lane] : 0.0;
            }
            // (the hand-over buffer is rewritten at the end of the next step, long after these reads)
            C3_STAMP(ts, k * C3_TS + 6, tid == 0);
        }
    }
}

// The riders -- R = A'(k+1, k) and the identity -- on the matrix cores: wave w of a set holds rows 16 w .. 16 w + 15 of its tile as
// four MFMA accumulator blocks (block cb: row 16 w + (lane & 15), column 16 cb + (lane >> 4) + 4 reg).  The accumulator layout of
// the 8 columns of a micro-panel (registers 2 h, 2 h + 1 of block p >> 1) IS the layout of an MFMA operand with k = column, so
//   solve     x = cur L_pp^-T      two MFMAs against the inverted 8 x 8 diagonal block (inv8, by the inverter wave), in place
//   trailing  acc -= x L[c, p]^T   two MFMAs per 16-column block to the right, operands: the L rows of the micro-panel from LDS
// need no transposition and no LDS round trip; only x goes to LDS (for the next diagonal tile A'(k+1,k+1) - R R^T, which two of
// the other waves accumulate -- in R's own waves its accumulators were spilled around every MFMA -- and for the right-hand side).  Round 4 first ran the riders like the
// diagonal tile (one row per lane, LDS broadcasts on the vector ALU): 3.5 us per micro-panel against 1.6 for the diagonal tile
// they follow, twelve waves on four SIMDs.
__device__ __noinline__ void c3_chain_rider(int set, int w, double* A, int n, int T, int* fail, const int* flags, int want1, double* Linv, double* dinv,
                                            int mirror, long long* ts) {
    extern __shared__ __attribute__((aligned(16))) double c3_lds[];
    set = c3_uni(set); w = c3_uni(w); A = c3_uni(A); n = c3_uni(n); T = c3_uni(T); fail = c3_uni(fail); flags = c3_uni(flags); want1 = c3_uni(want1);
    Linv = c3_uni(Linv); dinv = c3_uni(dinv); mirror = c3_uni(mirror); ts = c3_uni(ts);
    const int tid = threadIdx.x, lane = tid & 63;
    const int e16 = lane & 15, g4 = lane >> 4;
    const C3Lds l = c3_carve(c3_lds);
    const bool isR = set == 1;
    double* panS = l.pan + set * 8 * C3_BLK;
    int* cntS = l.lf + 8 * set;
    chol_d4 acc[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) acc[cb] = chol_d4{0.0, 0.0, 0.0, 0.0};
    if (isR && T > 1) {  // step 0: tile (1, 0) of the input
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int cc = 16 * cb + g4 + 4 * reg, row = 64 + 16 * w + e16;
                acc[cb][reg] = c3_ld_at(A, n, row, cc, row < n);
            }
    }
    for (int k = 0; k < T; ++k) {
        const int step1 = k + 1;
        const int r0 = 64 * k;
        const bool has_r = k + 1 < T;
        if (!isR) {
            int wi = w;
            asm volatile("" : "+v"(wi));  // (the identity is set up again at every step: hoisted out of the loop its 16 values sat in scratch)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) acc[cb][reg] = (cb == wi && e16 == g4 + 4 * reg) ? 1.0 : 0.0;
        }
        if (!isR || has_r) {
            // the right-hand side's wave is done with this set's rows of the previous step, and so are the products of R's four waves
            if (k > 0 && !c3_wait_lds(isR ? l.b_rdy : l.y_done, isR ? step1 : k, fail)) return;
            if (isR && k > 0 && !c3_wait_lds(l.prod_cnt, 4 * k, fail)) return;
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int cb = p >> 1, h = p & 1;
                if (!c3_wait_lds(l.inv_flag + p, step1, fail)) return;
                // ---- solve: x[row][jj] = sum_m cur[row][m] Linv8[jj][m]  (first operand: indexed by the output column)
                const double i0 = (e16 < 8) ? l.inv8[p * 64 + e16 * 8 + g4] : 0.0;
                const double i1 = (e16 < 8) ? l.inv8[p * 64 + e16 * 8 + 4 + g4] : 0.0;
                chol_d4 x = chol_d4{0.0, 0.0, 0.0, 0.0};
                // (row r of I L_kk^-T is column r of L_kk^-1: zero left of column r -- rows 16 w .. stay zero in the column blocks
                // before w, which are the ones with the most blocks to their right: more than half of the identity's MFMAs)
                const bool zero = !isR && cb < w;
                if (!zero) {
                    x = __builtin_amdgcn_mfma_f64_16x16x4f64(i0, acc[cb][2 * h], x, 0, 0, 0);
                    x = __builtin_amdgcn_mfma_f64_16x16x4f64(i1, acc[cb][2 * h + 1], x, 0, 0, 0);
                }
                acc[cb][2 * h] = x[0]; acc[cb][2 * h + 1] = x[1];
                panS[p * C3_BLK + (16 * w + e16) * C3_RS + g4] = x[0];
                panS[p * C3_BLK + (16 * w + e16) * C3_RS + 4 + g4] = x[1];
                asm volatile("" ::: "memory");
                if (lane == 0) c3_lds_inc(cntS + p);  // (a wave's LDS operations execute in order: the rows are in place before the count)
                C3_STAMP(ts, k * C3_TS + 8 + 8 * set + p, lane == 0 && w == 3);
                const double xn0 = -x[0], xn1 = -x[1];
                // ---- trailing update of the columns to the right (h == 0: the second half of this block, too)
#pragma unroll
                for (int cb2 = 0; cb2 < 4; ++cb2) {
                    if ((cb2 > cb || (cb2 == cb && h == 0)) && !zero) {
                        double v0 = l.pan[p * C3_BLK + (16 * cb2 + e16) * C3_RS + g4];
                        double v1 = l.pan[p * C3_BLK + (16 * cb2 + e16) * C3_RS + 4 + g4];
                        if (cb2 == cb && e16 < 8) { v0 = 0.0; v1 = 0.0; }  // the solved columns stay
                        acc[cb2] = __builtin_amdgcn_mfma_f64_16x16x4f64(v0, xn0, acc[cb2], 0, 0, 0);
                        acc[cb2] = __builtin_amdgcn_mfma_f64_16x16x4f64(v1, xn1, acc[cb2], 0, 0, 0);
                    }
                }
            }
        }
        C3_STAMP(ts, k * C3_TS + 1 + set, lane == 0 && w == 3);
        // (the address arithmetic of the stores and loads below must not be hoisted out of the step loop: held in registers across
        // the micro-panels it pushed the loop's own LDS addresses into scratch -- four reloads per micro-panel on the riders' path)
        asm volatile("" : "+v"(n), "+v"(A), "+v"(Linv), "+v"(dinv));
        // ---- publish (write-through); the mirror and the 32 x 32 block inverses are plain stores: read by later kernels only
        if (isR) {
            if (has_r) {
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int row = r0 + 64 + 16 * w + e16, col = r0 + 16 * cb + g4 + 4 * reg;
                        if (row < n) c3_st(c3_at(A, (unsigned)(row + col * n)), acc[cb][reg]);
                    }
            }
        } else {
            const int m = 16 * w + e16;  // row m of L^-T = column m of L^-1
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int c = 16 * cb + g4 + 4 * reg;
                    c3_st(c3_at(Linv, (unsigned)(k * 4096 + m * 64 + c)), acc[cb][reg]);
                }
        }
        C3_STAMPI(ts, k * C3_TS + 8 + 8, !isR && lane == 0 && w == 3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my stores are acknowledged
        // (R's waves count only the steps in which they store: in the last step they have nothing to do, and a wave that ran ahead
        // into it would count twice before a slower one has drained the step before -- the last R would be flagged too early, which
        // its only reader, the workgroup that writes its transpose, showed as one wrong solve in a hundred at three tile rows)
        if (!isR || has_r) {
            // (as for the diagonal tile: this step's count only behind the complete count of the previous step -- R: 4 k, the identity and
            // the right-hand side's wave: 5 k)
            if (k > 0 && !c3_wait_lds(isR ? l.pubR_cnt : l.pub_cnt, (isR ? 4 : 5) * k, fail)) return;
            if (lane == 0) c3_lds_inc(isR ? l.pubR_cnt : l.pub_cnt);
        }
        C3_STAMPI(ts, k * C3_TS + 8 + 9, !isR && lane == 0 && w == 3);
        if (!isR) {
            // ---- plain stores for the back-substitution kernel: the 32 x 32 block inverses
            const int hb = w >> 1;
            if (dinv && 32 * (2 * k + hb) < n) {
                const int m = 16 * w + e16;
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int c = 16 * cb + g4 + 4 * reg;  // [blk][row of the inverse][column]; identity padding inverts to itself
                        if ((cb >> 1) == hb) c3_gst(c3_at(dinv, (unsigned)(((2 * k + hb) * 32 + (c - 32 * hb)) * 32 + (m - 32 * hb))), acc[cb][reg]);
                    }
            }
        }
        // ---- R: input of the next step.  Tile (k+2, k+1) comes from its owner with every panel but this step's applied (published
        // a step ago); this step's panel is applied here: - L(k+2, k) R^T, the rows of L(k+2, k) straight from memory in the layout of
        // an MFMA operand (row 16 w + (lane & 15), column 4 ks + (lane >> 4)), R from LDS
        if (isR && k + 2 < T) {
            if (k + 1 >= 2 && !c3_wait(flags + (k + 2) * T + (k + 1), want1, fail)) return;
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int cc = 16 * cb + g4 + 4 * reg, row = r0 + 128 + 16 * w + e16;
                    acc[cb][reg] = c3_ld_at(A, n, row, r0 + 64 + cc, row < n);
                }
            if (!c3_wait(flags + (k + 2) * T + k, 4 * (want1 >> 2) + C3_S2, fail)) return;
            double lv[16];
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) lv[ks] = c3_ld_at(A, n, r0 + 128 + 16 * w + e16, r0 + 4 * ks + g4, r0 + 128 + 16 * w + e16 < n);
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) lv[ks] = -lv[ks];
            if (!c3_wait_lds(cntS + 7, 4 * step1, fail)) return;  // every row of R is in LDS
#pragma unroll
            for (int p = 0; p < 8; ++p) {
#pragma unroll
                for (int ks2 = 0; ks2 < 2; ++ks2) {
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) {
                        const double av = panS[p * C3_BLK + (16 * cb + e16) * C3_RS + 4 * ks2 + g4];
                        acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, lv[2 * p + ks2], acc[cb], 0, 0, 0);
                    }
                }
            }
            asm volatile("" ::: "memory");
            if (lane == 0) c3_lds_inc(l.prod_cnt);  // this wave is done with R's rows in LDS
        }
    }
}

// the two helper waves that form the next diagonal tile (their own function: compiled together with the inverter and the right-hand
// side's wave the five bases were spilled -- the tile was ready 12 us after R instead of 2)
template <int Q>
__device__ __noinline__ void c3_chain_dnext(double* A, int n, int T, int* fail, const int* flags, int want1, int mirror, long long* ts) {
    extern __shared__ __attribute__((aligned(16))) double c3_lds[];
    A = c3_uni(A); n = c3_uni(n); T = c3_uni(T); fail = c3_uni(fail); flags = c3_uni(flags); want1 = c3_uni(want1); ts = c3_uni(ts);
    const int lane = threadIdx.x & 63;
    const C3Lds l = c3_carve(c3_lds);
    for (int k = 0; k < T; ++k) {
        const bool has_r = k + 1 < T;
        if (has_r) {
            if (!c3_dnext_blocks<Q>(l, A, n, T, k, fail, flags, want1, lane)) return;
            C3_STAMP(ts, k * C3_TS + 4, lane == 0 && Q == 1);
        }
    }
}

// The other four waves of the chain workgroup, each in a function of its own (compiled together their registers went to scratch):
// the inverter (the 8 x 8 diagonal blocks of the micro-panels for the riders), the right-hand side's wave, which also publishes
// the step, and the two that form the next diagonal tile (c3_chain_dnext).
__device__ __noinline__ void c3_chain_inv(int T, int* fail) {
    extern __shared__ __attribute__((aligned(16))) double c3_lds[];
    T = c3_uni(T); fail = c3_uni(fail);
    const int lane = threadIdx.x & 63;
    const C3Lds l = c3_carve(c3_lds);
    for (int k = 0; k < T; ++k) {
        const int step1 = k + 1;
        // ---- X = L_pp^-1, lane c (< 8) solves column c by forward substitution; the L entries come as LDS broadcasts
        // (inv8[p] of the previous step has been read by every rider: D's micro-panel p of this step exists)
        const int c = lane & 7;
        for (int p = 0; p < 8; ++p) {
            if (!c3_wait_lds(l.lf + p, step1, fail)) return;
            const double* Lp = l.pan + p * C3_BLK + 8 * p * C3_RS;
            double x[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                double s0 = (r == c) ? 1.0 : 0.0, s1 = 0.0;
#pragma unroll
                for (int m = 0; m < r; ++m) {
                    if (m & 1) s1 = fma(-Lp[r * C3_RS + m], x[m], s1);
                    else s0 = fma(-Lp[r * C3_RS + m], x[m], s0);
                }
                x[r] = (s0 + s1) * l.pinv[8 * p + r];
            }
            if (lane < 8) {
#pragma unroll
                for (int r = 0; r < 8; ++r) l.inv8[p * 64 + r * 8 + c] = x[r];
            }
            asm volatile("" ::: "memory");
            if (lane == 0) c3_lds_set(l.inv_flag + p, step1);
        }
    }
}

// the right-hand side's wave (also publishes the step)
// arr (arr->arrive != null and arr->extra): the right-hand side of tile row k + 1 >= 2 is read from rhs, scaled by si, once the producers
// of that tile's columns have counted in (C3Args::arr_extra)
// (the arrival parameters come as scalars: behind a pointer to the caller's local struct they were ten flat loads from scratch per step)
__device__ __noinline__ void c3_chain_aux(double* A, int n, int T, int* fail, int* flags, int want1, int want2, double* b, const double* Cc, long long* ts,
                                          const int* arr_arrive, int arr_M, int arr_np, int arr_epoch, int arr_nap, long long arr_timeout, int arr_extra,
                                          const double* rhs, const double* si) {
    extern __shared__ __attribute__((aligned(16))) double c3_lds[];
    A = c3_uni(A); n = c3_uni(n); T = c3_uni(T); fail = c3_uni(fail); flags = c3_uni(flags); want1 = c3_uni(want1); want2 = c3_uni(want2);
    b = c3_uni(b); Cc = c3_uni(Cc); ts = c3_uni(ts); rhs = c3_uni(rhs); si = c3_uni(si);
    arr_arrive = c3_uni(arr_arrive); arr_M = c3_uni(arr_M); arr_np = c3_uni(arr_np); arr_epoch = c3_uni(arr_epoch); arr_nap = c3_uni(arr_nap); arr_extra = c3_uni(arr_extra);
    arr_timeout = ((long long)c3_uni((int)(arr_timeout >> 32)) << 32) | (unsigned)c3_uni((int)arr_timeout);
    const C3Arrive arr_v{arr_arrive, arr_M, arr_np, arr_epoch, arr_nap, arr_timeout, arr_extra};
    const C3Arrive* arr = &arr_v;
    const int lane = threadIdx.x & 63;
    const C3Lds l = c3_carve(c3_lds);

    for (int k = 0; k < T; ++k) {
        const int step1 = k + 1;
        const int r0 = 64 * k;
        const bool has_r = k + 1 < T;
        // ---- b_{k+1} - sum_{m < k} L(k+1, m) y_m first, in this fixed order: everything in it is at least a step old (the last
        // term comes with tile (k+1, k-1), whose owner had L_(k-1)(k-1)^-1 a step ago; R's waves of the last step waited for that
        // tile, but not necessarily before this point), and fetched behind R it cost 4 - 5 us of latency at the end of the step,
        // which the riders of the next step waited for
        double bn = 0.0;
        if (has_r) {
            if (k >= 1 && !c3_wait(flags + (k + 1) * T + (k - 1), want2, fail)) return;
            const int row = r0 + 64 + lane;
            if (arr && arr->arrive && arr->extra && k >= 1) {
                if (!c3_wait_arrive(*arr, r0 + 64, (r0 + 127 < n ? r0 + 127 : n - 1), fail)) return;
                const int rc = row < n ? row : n - 1;
                const double v = c3_ld(rhs + rc) / si[rc];
                bn = (row < n) ? v : 0.0;
            } else
            bn = (row < n) ? c3_ld(b + row) : 0.0;
            for (int m0 = 0; m0 < k; m0 += 8) {  // eight loads in flight
                double cv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const double cc_ = c3_ld(Cc + ((size_t)(k + 1) * T + (m0 + u < k ? m0 + u : k - 1)) * 64 + lane);
                    cv[u] = (m0 + u < k) ? cc_ : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) bn -= cv[u];
            }
        }
        // ---- right-hand side: y_k = L_kk^-1 b'_k (lane = row: sum_c (L^-T)[c][r] b'[c]), then b'_{k+1}
        if (!c3_wait_lds(l.lf + 16 + 7, 4 * step1, fail)) return;
        const double* panI = l.pan + 16 * C3_BLK + (lane >> 3) * C3_BLK + (lane & 7);
        const double* bk = l.bcur + 64 * (k & 1);
        double y0 = 0.0, y1 = 0.0;
#pragma unroll
        for (int c = 0; c < 64; c += 2) {
            y0 = fma(panI[c * C3_RS], bk[c], y0);
            y1 = fma(panI[(c + 1) * C3_RS], bk[c + 1], y1);
            if ((c & 14) == 14) C3_PIN2(y0, y1);  // sixteen entries' operands in flight
        }
        const double y = y0 + y1;
        l.yv[lane] = y;
        asm volatile("" ::: "memory");
        if (lane == 0) c3_lds_set(l.y_done, step1);  // the identity's rows of this step have been read
        if (r0 + lane < n) c3_st(b + r0 + lane, y);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (k > 0 && !c3_wait_lds(l.pub_cnt, 5 * k, fail)) return;  // (this step's count behind the previous step's complete one)
        if (lane == 0) c3_lds_inc(l.pub_cnt);
        // ---- publish L_kk, L_kk^-1 and y_k as soon as their nine storing waves (4 + 4 + 1, counted per role) have drained: the panel solves of the tiles
        // below start while R is still on its way
        if (!c3_wait_lds(l.pub_cnt, 5 * step1, fail) || !c3_wait_lds(l.pubD_cnt, 4 * step1, fail)) return;
        if (lane == 0) c3_st_flag(flags + k * T + k, want2);
        if (has_r) {
            // ---- publish R = L(k+1, k) as soon as its four storing waves have drained
            if (!c3_wait_lds(l.pubR_cnt, 4 * step1, fail)) return;
            if (lane == 0) c3_st_flag(flags + (k + 1) * T + k, want2);
            // ---- b'_{k+1} = (b_{k+1} - sum_{m < k} L(k+1, m) y_m) - R y_k
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const double2* row_ = reinterpret_cast<const double2*>(l.pan + 8 * C3_BLK + p * C3_BLK + lane * C3_RS);
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const double2 t = row_[m];
                    s0 = fma(t.x, l.yv[8 * p + 2 * m], s0);
                    s1 = fma(t.y, l.yv[8 * p + 2 * m + 1], s1);
                }
                if (p & 1) C3_PIN2(s0, s1);
            }
            bn -= s0 + s1;
            l.bcur[64 * ((k + 1) & 1) + lane] = bn;
            asm volatile("" ::: "memory");
            if (lane == 0) c3_lds_set(l.b_rdy, step1 + 1);  // also: this wave is done with the riders' rows of this step
        }
        C3_STAMP(ts, k * C3_TS + 5, lane == 0);
    }
}

__device__ __forceinline__ void c3_chain(const C3Args& g, int T) {
    extern __shared__ __attribute__((aligned(16))) double c3_lds[];
    const int tid = threadIdx.x;
    const C3Lds l = c3_carve(c3_lds);
    if (tid < 60) l.lf[tid] = 0;
    if (g.arrive) {
        // the chain's own input -- tiles (0,0), (1,0), (1,1), (2,1) and the right-hand side -- scaled in place once the producers of
        // the first two tile columns are done
        int* s_ok = l.lf + 61;
        const int n = g.n;
        if (tid < 64) {
            const C3Arrive r{g.arrive, g.arr_M, g.np, g.arr_epoch, g.nap, g.arr_timeout, g.arr_extra};
            C3_STAMP(g.ts, T * C3_TS + 0, tid == 0);
            const bool ok = c3_wait_arrive(r, 0, (n < 128 ? n : 128) - 1, g.fail);
            C3_STAMP(g.ts, T * C3_TS + 1, tid == 0);
            if (tid == 0) c3_lds_set(s_ok, ok ? 1 : 0);
        }
        __syncthreads();
        if (!c3_lds_get(s_ok)) return;
        const int rows = n < 192 ? n : 192, cols = n < 128 ? n : 128;
        for (int idx = tid; idx < rows * cols; idx += 1024) {
            const int row = idx % rows, col = idx / rows;
            if (row >= col && (col >= 64 || row < 128)) {  // (tile (2,0) has an owner)
                double* pa = g.A + (size_t)row + (size_t)col * n;
                c3_st(pa, c3_ld(pa) / (g.si[row] * g.si[col]));
            }
        }
        // (arr_extra: the entries of the later tile rows are not there yet -- c3_chain_aux takes them when their producers have counted in)
        for (int i = tid; i < (g.arr_extra ? (n < 128 ? n : 128) : n); i += 1024) c3_st(g.b + i, c3_ld(g.rhs + i) / g.si[i]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (tid < 64) l.bcur[tid] = (tid < g.n) ? c3_ld(g.b + tid) : 0.0;
    __syncthreads();
    if (tid == 0) c3_lds_set(l.b_rdy, 1);
    const int want2 = 4 * g.epoch + C3_S2, want1 = 4 * g.epoch + C3_S1;
    // Roles by wave.  The four waves of the diagonal tile are waves 0, 4, 8, 12: with waves dealt to the four SIMDs in turn they share
    // ONE SIMD, and every wave that issues MFMAs sits on the other three -- a 64-cycle fp64 MFMA in front of it delays a dependent
    // operation of the pivot chain (with one wave of every role per SIMD the diagonal tile took 16 us per step instead of 12.5).
    const int wave = tid >> 6;
    const int oth = (wave >> 2) * 3 + (wave & 3) - 1;  // 0 .. 11 over the waves that are not of the diagonal tile
    if ((wave & 3) == 0) c3_chain_diag(wave >> 2, g.A, g.n, T, g.fail, g.mirror, g.ts);
    else if (oth < 8) c3_chain_rider(1 + (oth >> 2), oth & 3, g.A, g.n, T, g.fail, g.flags, want1, g.Linv, g.dinv, g.mirror, g.ts);
    else if (oth == 10) c3_chain_dnext<0>(g.A, g.n, T, g.fail, g.flags, want1, g.mirror, g.ts);
    else if (oth == 11) c3_chain_dnext<1>(g.A, g.n, T, g.fail, g.flags, want1, g.mirror, g.ts);
    else if (oth == 8) c3_chain_inv(T, g.fail);
    else {
        c3_chain_aux(g.A, g.n, T, g.fail, g.flags, want1, want2, g.b, g.Cc, g.ts, g.arrive, g.arr_M, g.np, g.arr_epoch, g.nap, g.arr_timeout, g.arr_extra, g.rhs, g.si);
    }
}

// ---------------------------------------------------------------------------------------------------------------- owners
// Tasks behind the chain (ticket 0), column by column j: the diagonal tile (j, j) and the tile below it (j+1, j) (j >= 2) get all
// panels but the LAST one applied and are handed to the chain, which applies the last panel itself -- it has just produced it
// (R R^T for the diagonal tile, L(j+1, j-1) R^T for the tile below: c3_chain_rider); the other tiles (i, j), i >= j + 2, get all
// panels, are multiplied by L_jj^-T and published.  Of everything the chain reads only L(k+2, k) is younger than a step.
// (Round 4 on the way here: (j+1, j) with all panels applied by its owner -- 13 us from the chain's publication to its next
// input, through four hand-overs; one workgroup for (j+2, j) and (j+2, j+1) -- 10 us.)
// kind: 0 diagonal, 1 below the diagonal, 2 ordinary
// kind 3 (mirror != 0): L^T of the chain's tiles of step j -- (j, j) and (j+1, j) -- into the strict upper triangle, behind column j's
// other tasks (it waits for the step's publication only: every ticket it depends on is lower)
__host__ __device__ inline int c3_task_count(int T, int mirror) {
    int cnt = 1;
    for (int j = 0; j < T; ++j) {
        if (j >= 2) ++cnt;
        if (j >= 2 && j + 1 <= T - 1) ++cnt;
        if (T - 1 >= j + 2) cnt += T - 1 - (j + 2) + 1;
        if (mirror) ++cnt;
    }
    return cnt;
}
__device__ inline void c3_task(int T, int mirror, int idx, int& i, int& j, int& kind) {  // idx >= 1
    --idx;
    for (j = 0; j < T; ++j) {
        if (j >= 2) { if (idx == 0) { i = j; kind = 0; return; } --idx; }
        if (j >= 2 && j + 1 <= T - 1) { if (idx == 0) { i = j + 1; kind = 1; return; } --idx; }
        const int cnt = T - 1 >= j + 2 ? T - 1 - (j + 2) + 1 : 0;
        if (idx < cnt) { i = j + 2 + idx; kind = 2; return; }
        idx -= cnt;
        if (mirror) { if (idx == 0) { i = j; kind = 3; return; } --idx; }
    }
    i = j = 0; kind = -1;
}

// L^T of the chain's two tiles of step k for the back-substitution kernel: through LDS ([column][row], stride 65), so that a wave
// stores 64 consecutive addresses.  (Inside the chain's workgroup -- by the waves that form the next diagonal tile, by the diagonal
// tile's own, by the inverter's -- it sat in front of the next step, which reuses the micro-panel blocks it was read from: 2 - 5 us
// per step; a wave there gets an issue slot every ~13 cycles beside the MFMAs.)
__device__ __noinline__ void c3_mirror_task(c3_kargs kargs, int T, int k) {
    const C3Args g = c3_args_from_kernarg(kargs);
    extern __shared__ __attribute__((aligned(16))) double c3_lds[];
    double* Xt = c3_lds;
    int* s_ok = reinterpret_cast<int*>(c3_lds + 3 * 8 * C3_BLK + 64 * C3_TBS + 64 + 64 + 128 + 512) + 61;
    const int tid = threadIdx.x, n = g.n, r0 = 64 * k;
    const int want2 = 4 * g.epoch + C3_S2;
    double* A = g.A;
    const int n_tiles = (k + 1 < T) ? 2 : 1;
    for (int t2 = 0; t2 < n_tiles; ++t2) {
        if (tid == 0) c3_lds_set(s_ok, c3_wait(g.flags + (k + t2) * T + k, want2, g.fail) ? 1 : 0);
        __syncthreads();  // also: the stores of the first tile have read Xt
        if (!c3_lds_get(s_ok)) return;
        const int rt = r0 + 64 * t2;  // first row of the tile
        double v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int idx = tid + 1024 * t, r = idx & 63, c = idx >> 6;
            v[t] = c3_ld_at(A, n, rt + r, r0 + c, rt + r < n && r0 + c < n && (t2 == 1 || c < r));
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) { const int idx = tid + 1024 * t, r = idx & 63, c = idx >> 6; Xt[c * 65 + r] = v[t]; }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int idx = tid + 1024 * t, c = idx & 63, r = idx >> 6;
            if (rt + r < n && r0 + c < n && (t2 == 1 || c < r)) c3_gst(A + (size_t)(r0 + c) + (size_t)(rt + r) * n, Xt[c * 65 + r]);
        }
    }
}

__device__ __noinline__ void c3_owner(c3_kargs kargs, int T, int i, int j, int kind) {
    const C3Args g = c3_args_from_kernarg(kargs);
    extern __shared__ __attribute__((aligned(16))) double c3_lds[];  // (declared here, not passed: the pointers stay in the LDS address space)
    double* lds = c3_lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rb = wave & 3, cb = wave >> 2, e16 = lane & 15, g4 = lane >> 4;
    const int n = g.n;
    double* A = g.A;
    double* Li = lds;                       // [64][C3_LD]: [k][row] of L(i, m)
    double* Lj = Li + 64 * C3_LD;           // L(j, m)
    double* part = Lj + 64 * C3_LD;         // [4][64]
    double* yv = part + 256;                // [64]
    int* s_ok = reinterpret_cast<int*>(lds + 3 * 8 * C3_BLK + 64 * C3_TBS + 64 + 64 + 128 + 512) + 61;
    const int want2 = 4 * g.epoch + C3_S2, want1 = 4 * g.epoch + C3_S1;
    const int r0 = 64 * i, c0 = 64 * j;
    const bool diag = kind == 0;
    const int rr = 16 * rb + e16;
    chol_d4 old = chol_d4{0.0, 0.0, 0.0, 0.0}, acc = chol_d4{0.0, 0.0, 0.0, 0.0};
    // beside the producing kernel the tile itself is read LAST: the sum of the panel products does not need it, and the late tile
    // columns arrive long after their first panels (read first, the diagonal tiles reached the chain 20 - 30 us after they had arrived)
    double sprod[4] = {1.0, 1.0, 1.0, 1.0};  // scale_inv[row] scale_inv[col] of the thread's four entries
    if (g.arrive) {
        const double si_r = g.si[r0 + rr < n ? r0 + rr : n - 1];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) { const int col = c0 + 16 * cb + g4 + 4 * reg; sprod[reg] = si_r * g.si[col < n ? col : n - 1]; }
    }
    auto load_arrived = [&]() -> bool {
        if (wave == 0) {
            const C3Arrive r{g.arrive, g.arr_M, g.np, g.arr_epoch, g.nap, g.arr_timeout, g.arr_extra};
            const bool ok = c3_wait_arrive(r, c0, (c0 + 63 < n ? c0 + 63 : n - 1), g.fail);
            if (tid == 0) c3_lds_set(s_ok, ok ? 1 : 0);
        }
        __syncthreads();
        if (!c3_lds_get(s_ok)) return false;
        C3_STAMP(g.ts, T * C3_TS + j, tid == 0 && kind == 0);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int cc = 16 * cb + g4 + 4 * reg;
            const int row = r0 + rr, col = c0 + cc;
            const double v = c3_ld_at(A, n, row, col, row < n && col < n && (!diag || cc <= rr));
            old[reg] = v / sprod[reg];
        }
        __syncthreads();  // (s_ok is used again)
        return true;
    };
    if (!g.arrive) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int cc = 16 * cb + g4 + 4 * reg;
            const int row = r0 + rr, col = c0 + cc;
            double v = 0.0;
            if (row < n && col < n && (!diag || cc <= rr)) v = c3_gld(A + (size_t)row + (size_t)col * n);
            old[reg] = v;
        }
    }
    const int n_upd = kind == 2 ? j : j - 1;
    for (int m = 0; m < n_upd; ++m) {
        if (tid == 0) {
            bool ok = c3_wait(g.flags + i * T + m, want2, g.fail);
            if (ok && !diag) ok = c3_wait(g.flags + j * T + m, want2, g.fail);
            c3_lds_set(s_ok, ok ? 1 : 0);
        }
        __syncthreads();  // also: the previous product is done with the operand tiles
        if (!c3_lds_get(s_ok)) return;
        double vi[4], vj[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int idx = tid + 1024 * t, r = idx & 63, kk = idx >> 6;
            vi[t] = c3_ld_at(A, n, r0 + r, 64 * m + kk, r0 + r < n);
            vj[t] = c3_ld_at(A, n, c0 + r, 64 * m + kk, !diag && c0 + r < n);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int idx = tid + 1024 * t, r = idx & 63, kk = idx >> 6;
            Li[kk * C3_LD + r] = vi[t];
            if (!diag) Lj[kk * C3_LD + r] = vj[t];
        }
        __syncthreads();
        const double* Pj = diag ? Li : Lj;
        if (!diag || cb <= rb) {
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const double av = Pj[(4 * ks + g4) * C3_LD + 16 * cb + e16];
                const double bv = Li[(4 * ks + g4) * C3_LD + 16 * rb + e16];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
            }
        }
    }
    if (g.arrive && !load_arrived()) return;
    if (kind != 2) {
        // ---- hand the tile to the chain: all panels but the last applied
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int cc = 16 * cb + g4 + 4 * reg;
            const int row = r0 + rr, col = c0 + cc;
            if (row < n && col < n && (!diag || cc <= rr)) c3_st(A + (size_t)row + (size_t)col * n, old[reg] - acc[reg]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) c3_st_flag(g.flags + i * T + j, want1);
        C3_STAMP(g.ts, (T + 1) * C3_TS + j, tid == 0 && kind == 0);
        return;
    }
    // ---- panel solve as a product: L(i,j) = A' L_jj^-T, X[r][c] = sum_m A'[r][m] Linv[c][m]
    __syncthreads();  // the last product is done with the operand tiles
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) Li[(16 * cb + g4 + 4 * reg) * C3_LD + 16 * rb + e16] = old[reg] - acc[reg];  // [m][r]
    if (tid == 0) c3_lds_set(s_ok, c3_wait(g.flags + j * T + j, want2, g.fail) ? 1 : 0);
    __syncthreads();
    if (!c3_lds_get(s_ok)) return;
    {
        double v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = c3_ld(g.Linv + (size_t)j * 4096 + tid + 1024 * t);  // [m][c]: column m of L^-1
#pragma unroll
        for (int t = 0; t < 4; ++t) { const int idx = tid + 1024 * t; Lj[(idx >> 6) * C3_LD + (idx & 63)] = v[t]; }
        if (tid < 64) { const double yy = c3_ld(g.b + (c0 + tid < n ? c0 + tid : n - 1)); yv[tid] = (c0 + tid < n) ? yy : 0.0; }
    }
    __syncthreads();
    chol_d4 x = chol_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        const double av = Lj[(4 * ks + g4) * C3_LD + 16 * cb + e16];
        const double bv = Li[(4 * ks + g4) * C3_LD + 16 * rb + e16];
        x = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, x, 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int cc = 16 * cb + g4 + 4 * reg;
        const int row = r0 + rr, col = c0 + cc;
        if (row < n && col < n) c3_st(A + (size_t)row + (size_t)col * n, x[reg]);
        s = fma(x[reg], yv[cc], s);
    }
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (g4 == 0) part[cb * 64 + 16 * rb + e16] = s;
    __syncthreads();
    if (tid < 64) c3_st(g.Cc + ((size_t)i * T + j) * 64 + tid, (part[tid] + part[64 + tid]) + (part[128 + tid] + part[192 + tid]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) c3_st_flag(g.flags + i * T + j, want2);
    if (g.mirror) {
        // ---- L^T for the back-substitution kernel, behind the flag: the tile goes through LDS ([column][row], stride 65) so that a
        // wave stores 64 consecutive addresses (as 8-byte stores scattered over 64 lines they sat in front of the drain: +25 us)
        double* Xt = Lj;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) Xt[(16 * cb + g4 + 4 * reg) * 65 + rr] = x[reg];
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int idx = tid + 1024 * t, c = idx & 63, r = idx >> 6;
            if (r0 + r < n && c0 + c < n) c3_gst(A + (size_t)(c0 + c) + (size_t)(r0 + r) * n, Xt[c * 65 + r]);
        }
    }
}

__global__ __launch_bounds__(1024) void k_chol_tiles(C3Args g, const int* gate) {
    SATBA_GATE(gate);
    extern __shared__ __attribute__((aligned(16))) double c3_lds[];
    int* s_task = reinterpret_cast<int*>(c3_lds + 3 * 8 * C3_BLK + 64 * C3_TBS + 64 + 64 + 128 + 512) + 60;  // (all LDS in the dynamic region: 16-byte aligned base)
    if (threadIdx.x == 0) {
        const long long extra = g.arrive ? g.arr_timeout : 0ll;  // ticks of 10 ns
        const long long lg = (long long)C3_SPIN_LIMIT + extra / 20, ll = 4ll * C3_SPIN_LIMIT + extra / 4;
        int* lim = reinterpret_cast<int*>(c3_lds + C3_LDS_INTS_AT) + 62;
        c3_lds_set(lim, (int)(lg < 0x7fffffffll ? lg : 0x7fffffffll));
        c3_lds_set(lim + 1, (int)(ll < 0x7fffffffll ? ll : 0x7fffffffll));
    }
    const int T = (g.n + 63) / 64;
    const c3_kargs kargs = (c3_kargs)__builtin_amdgcn_kernarg_segment_ptr();  // (C3Args is the first parameter: offset 0)
    const int n_tasks = c3_task_count(T, g.mirror);
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) c3_lds_set(s_task, atomicAdd(g.ctr, 1));
        __syncthreads();
        const int task = c3_lds_get(s_task);
        if (task >= n_tasks) break;
        if (task == 0) {
            c3_chain(g, T);
        } else {
            int i, j, kind;
            c3_task(T, g.mirror, task, i, j, kind);
            if (kind == 3) c3_mirror_task(kargs, T, j);
            else c3_owner(kargs, T, i, j, kind);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int done = atomicAdd(g.ctr + 1, 1);
        if (done == (int)gridDim.x - 1) { g.ctr[0] = 0; g.ctr[1] = 0; }  // the last workgroup leaves the counters clean
        if (g.arrive) c3_lds_set(s_task, done == (int)gridDim.x - 1 ? 1 : 0);
    }
    if (g.arrive) {
        __syncthreads();
        if (c3_lds_get(s_task))
            for (int c = threadIdx.x; c < g.arr_M; c += 1024) g.arrive[(size_t)C3_ARRIVE_STRIDE * c] = 0;
    }
}

// ---------------------------------------------------------------------------------------------------------------- small systems
// n <= 63 unknowns (the reference's usual case: a dozen cameras): the whole solve phase in ONE launch of one workgroup -- scaling of
// the reduced system, factorisation, forward and backward substitution, the step back in unscaled variables and the header of the
// phase (k_scale_system, k_chol_dstep, k_trsv_back_small, k_unscale before: 35 us of a 150 us iteration at 10 cameras x 5).  The
// diagonal-tile code of the chain above (c3_panel: four waves, lane = row, 8-column micro-panels) factorises the tile
//     [ S'   .  ]      S' = D S D (D = 1 / scale_inv), identity padding to 63,
//     [ b'^T big ]     b' = D rhs in row 63
// whose last row comes out as y^T = (L^-1 b')^T: the forward substitution rides along as a row.  Then one wave substitutes backwards
// from the micro-panels in LDS.  hdr: the solve phase's header as k_unscale writes it.
__global__ __launch_bounds__(256) void k_solve_small(int n, const double* __restrict__ scale_inv, const double* __restrict__ S, const double* __restrict__ rhs,
                                                     double* __restrict__ dch, double* __restrict__ dc, int* __restrict__ fail, int n_clear, int hdr_len,
                                                     double* __restrict__ hdr, double lead, const double* __restrict__ keep, int keep_at, int keep_len,
                                                     const int* gate) {
    SATBA_GATE(gate);
    __shared__ __attribute__((aligned(16))) double s_pan[8 * C3_BLK];
    __shared__ __attribute__((aligned(16))) double s_pinv[64];
    __shared__ int s_lf[8];
    __shared__ int s_bad;
    const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
    for (int i = tid; i < n_clear; i += 256) fail[i] = 0;  // the dense solver's status word and the flag words behind it
    if (tid < 8) s_lf[tid] = 0;
    if (tid == 0) s_bad = 0;
    const double si_row = scale_inv[lane < n ? lane : 0];
    double a[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int col = 16 * q + c;
        double v = (col == lane) ? 1.0 : 0.0;  // identity padding
        if (col < n) {
            const double si_col = scale_inv[col];
            if (lane < n && col <= lane) v = S[(size_t)lane + (size_t)col * n] / (si_row * si_col);  // (k_scale_system's arithmetic)
            else if (lane == 63) v = rhs[col] / si_col;
        } else if (lane == 63 && col == 63) {
            v = 1e200;  // any value above |y|^2: the last pivot is not used
        }
        a[c] = v;
    }
    __syncthreads();
    const bool bad = c3_panel(a, q, lane, s_pan, s_pinv, s_lf, 1, fail, nullptr, (n + 7) >> 3, false);  // (padding columns and the corner are not factorised)
    if (bad && lane == 0) atomicOr(&s_bad, 1);
    __syncthreads();  // every micro-panel is in LDS
    if (q == 0) {
        // z = L^-T y: lane = column c of L (k_trsv_back_small's loop, the factor read from the micro-panel blocks: entry (k, c) at
        // [c / 8][k][c % 8])
        const int c = lane, cc = min(c, n - 1);
        const double* colp = s_pan + (cc >> 3) * C3_BLK + (cc & 7);
        double t[63];
#pragma unroll
        for (int k = 0; k < 63; ++k) t[k] = (k < n && k > c) ? colp[k * C3_RS] : 0.0;
        const double inv = s_pinv[cc];
        double y = (c < n) ? colp[63 * C3_RS] : 0.0;
#pragma unroll
        for (int k = 62; k >= 0; --k) {
            if (k < n) {  // (uniform)
                const double zk = readlane_f64(y * inv, k);
                y = (c == k) ? zk : fma(-t[k], zk, y);  // t[k] is zero at and behind the diagonal (lanes >= k)
            }
        }
        if (c < n) { dch[c] = y; dc[c] = y / scale_inv[c]; }
    }
    const int failed = s_bad;
    if (tid == 0 && failed) fail[0] = 1;
    for (int i = tid; i < hdr_len; i += 256) {
        double v = 0.0;
        if (i == 4) v = failed ? lead : 0.0;
        if (i >= keep_at && i < keep_at + keep_len) v = lead * keep[i - keep_at];
        hdr[i] = v;
    }
}
constexpr int CH_ONE_LAUNCH = 63;  // largest system k_solve_small takes

inline int chol_tiles_grid(int n, int mirror) {
    const int n_tasks = c3_task_count((n + 63) / 64, mirror);
    return n_tasks < 256 ? n_tasks : 256;
}

inline void chol_tiles_init() {
    static const bool once = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chol_tiles), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c3_lds_bytes());
        return true;
    }();
    (void)once;
}

}  // namespace satba
