// satba_lmdev.h -- the trust-region loop's DECISIONS on the device (round 3): what satba_solve_lm did on the host between two
// header reads -- scipy's top-of-loop tests, the 2-D trust-region subproblem on span{g_h, gn_h}, the radius update, the accept /
// reject decision and the termination tests (scipy:optimize/_lsq/trf.py:450-551, common.py:171-245, 705-717; the scalar code is
// satba_lm.h, compiled for both sides) -- runs in two one-thread kernels, and the host no longer waits for the device inside an
// iteration.
//
// A TICK is one trial evaluation.  The host queues the same pattern of launches for every tick without knowing the outcome of
// the previous one:
//
//     [linearize -> prepare]                                                                    gate: st.run_lin
//     [Schur complement -> dense solve -> back-substitution]                                    gate: st.run_solve
//     k_lm_decide1a  (bookkeeping, top-of-loop tests, damping escalation, model)                (always runs, one thread)
//     [explicit subspace vectors]                                                               gate: st.run_sub    (degenerate case)
//     k_lm_decide1b
//     [explicit products J_h q1, J_h w]                                                         gate: st.run_prod   (degenerate case)
//     k_lm_decide1c
//     [trial: camera constants at the trial point, residual kernel with the step folded in]     gate: st.run_trial
//     k_lm_decide2   (radius update, accept / reject, termination)                              (always runs, one thread)
//     k_lm_accept: x <- x_new, camera constants, cost bound                                     gate: st.accept
//
// Every kernel of the pattern starts by reading its gate word from the loop's state `LmDev` in device memory (SATBA_GATE) and
// returns when it is 0.  After a REJECTED trial the next tick's front is switched off (same x: scipy keeps the quadratic model and
// only shrinks the radius, trf.py:493-521) and decide1a solves the subproblem again with the new radius; after an accepted one the
// point is copied (not swapped: the launch arguments of the next tick are already queued) and the front runs.  A failed Cholesky
// (gauge freedom, flat valleys) switches only the second half of the front on again, with the damping escalated x 100, up to ten
// times like the host loop.  When the loop terminates -- or the fixed-point camera sums overflow, which needs the host to switch the
// summation route -- the state switches every later tick off; the host notices through `LmSummary` (pinned host memory the
// device posts to) and stops queueing.  A handful of empty ticks may have been queued by then (at most LM_RUN_AHEAD).
//
// The pattern is data independent; capturing it into a hipGraph was measured in round 3 (profiles/r3_graph_gaps.txt: the replay leaves
// 2 - 9 us between its nodes where back-to-back launches leave none, and the host, which queues LM_RUN_AHEAD ticks ahead, is never the
// bottleneck) and removed in round 4: the ticks are direct launches.
#pragma once
#include <hip/hip_runtime.h>

#include "satba_lm.h"

namespace satba {

enum { LM_RUN = 0, LM_DONE = 1, LM_NEED_HOST = 2, LM_NEED_SUB = 3 };
enum { LM_HOST_NONE = 0, LM_HOST_FX = 1, LM_HOST_CHOL = 2, LM_HOST_NONFINITE = 3, LM_HOST_BESIDE = 4 };
constexpr int LM_RUN_AHEAD = 3;  // ticks the host may queue beyond the last one the device has reported

struct LmDev {
    // gates (SATBA_GATE) and control
    int run_lin, run_solve, run_sub, run_prod, run_trial, accept, restore;
    int phase, status, first, one_dim, have_actual, host_reason, never_stop, attempts;
    int resume_lin;  // LM_HOST_BESIDE: run_lin at the hand-over (the linearisation of the repeated front is not booked yet iff 1)
    int accepted_total, interior_total, cycle_len, cycle_it;
    long long nfev, njev, iterations, max_nfev, tick;
    long long max_iterations;  // fixed-work runs (never_stop): stop after this many iterations (0: never)
    long long sub_requests;
    long long sub_tick, end_tick;  // the tick (1-based) of the latest pause for the subspace pattern / at which the loop left LM_RUN (0: not yet)
    // trust region and the scalars of scipy's loop
    double Delta, cost, g_norm, initial_cost, step_norm, actual;
    // quadratic model on the orthonormal basis of span{g_h, gn_h} (kept across rejected trials)
    double Ba, Bb, Bc, gS0, gS1, sa, alpha, nw;
    double ga, gb, gc, jg_sq, reg;
    double ftol, xtol, gtol;
    double lam_force;    // > 0: the damping of the next Schur complement (escalation after a failed Cholesky) instead of the Cauchy-step value
    double sub_args[2];  // alpha, 1 / |g_h|: arguments of the explicit subspace vectors (degenerate case)
    // the trial step: coefficients of (g_h, gn_h) [read by the trial kernels], predicted reduction, |step| in scaled variables
    double coef[2], predicted, step_h_norm, cost_new;
};

// what the host polls, in pinned host memory: written by k_lm_decide2 with relaxed system-scope stores (uncached, straight over
// PCIe).  No release fence: at system scope it writes the whole L2 back (tens of microseconds per tick, measured).  Every word is
// self-contained instead: `word` = executed patterns << 8 | phase, and each of the other three carries the tick it was written at
// in its upper half (stamp << 32 | value).  A reader that finds a stamp older than the tick in `word` has caught the stores half
// way and polls again (satba_lm_poll, lm_summary_read): with several ranks every rank must act on values at least as new as the
// tick it acts on, or the ranks queue different patterns and their collectives no longer match (round-4 advisor finding).
struct LmSummary {
    unsigned long long word;
    unsigned long long sub_requests;  // how often the loop has paused for the subspace pattern
    // for several ranks (satba_lm_part): every rank must queue the SAME number of patterns, so the host does not act on "the latest
    // report" but on these two tick numbers, which are functions of all-reduced scalars only
    unsigned long long sub_tick, end_tick;
};
constexpr long long LM_MAX_TICKS = (1ll << 31) - 8;  // ticks travel in 32 bits of the stamped words
__host__ __device__ inline long long lm_stamp_tick(unsigned long long w) { return (long long)(w >> 32); }
__host__ __device__ inline long long lm_stamp_value(unsigned long long w) { return (long long)(w & 0xffffffffull); }
__host__ __device__ inline long long lm_summary_tick(unsigned long long w) { return (long long)(w >> 8); }
__host__ __device__ inline int lm_summary_phase(unsigned long long w) { return (int)(w & 0xff); }

// header slots (satba/trf.py)
enum { LMH_GRAM_A = 1, LMH_GRAM_B = 2, LMH_GRAM_C = 3, LMH_CHOL_FAIL = 4, LMH_COST_NEW = 1, LMH_STEP_SQ = 2, LMH_X_SQ = 3,
       LMH_WW = 1, LMH_B11 = 3, LMH_B12 = 4, LMH_B22 = 5, LMH_GHW = 6,
       LMH_K_COST = 8, LMH_K_GINF = 9, LMH_K_JG_SQ = 11, LMH_K_LAM = 13, LMH_K_DELTA = 14, LMH_FX_BAD = 15 };

// start of a solve: the state comes by value (queued like everything else: no host staging buffer to wait for)
__global__ void k_lm_reset(LmDev* __restrict__ st, LmDev init, int keep_counters) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (keep_counters) { init.accepted_total = st->accepted_total; init.interior_total = st->interior_total; }
    *st = init;
}

__device__ inline void lm_skip_rest(LmDev* st) { st->run_sub = 0; st->run_prod = 0; st->run_trial = 0; st->accept = 0; st->restore = 0; }

// the trust-region subproblem for the current radius on the kept model, and the coefficients of the trial step on (g_h, gn_h)
__device__ inline void lm_trial_step(LmDev* st) {
#pragma clang fp contract(off)
    double p0, p1;
    const bool newton = satba_lm::solve_trust_region_2d(st->Ba, st->Bb, st->Bc, st->gS0, st->gS1, st->Delta, p0, p1);
    st->interior_total += newton ? 1 : 0;
    st->predicted = -(0.5 * (p0 * (st->Ba * p0 + st->Bb * p1) + p1 * (st->Bb * p0 + st->Bc * p1)) + st->gS0 * p0 + st->gS1 * p1);
    st->coef[0] = st->one_dim ? p0 / st->sa : p0 / st->sa - p1 * st->alpha / st->nw;
    st->coef[1] = st->one_dim ? 0.0 : p1 / st->nw;
    st->step_h_norm = sqrt(p0 * p0 + p1 * p1);
    st->run_trial = 1;
}

// Hand-over to the host loop: every gate goes down with the phase, so that the ticks already queued behind this one (up to
// LM_RUN_AHEAD) pass through empty instead of repeating the front at the same x (the host loop starts from x, not from the gates)
__device__ inline void lm_to_host(LmDev* st, int reason) {
    st->phase = LM_NEED_HOST; st->host_reason = reason;
    st->run_lin = 0; st->run_solve = 0;
}

// After the front (the parts of it that ran): bookkeeping of a new linearisation and scipy's top-of-loop tests, the damping
// escalation after a failed factorisation, the quadratic model from the normal equations -- or the request for the explicit
// subspace vectors when the two directions are parallel to 1e-6 (satba/trf.py: subspace_model).  h: the solve header.
__device__ inline void lm_decide1a(LmDev* __restrict__ gst, const double* __restrict__ h) {
#pragma clang fp contract(off)
    if (gst->phase != LM_RUN) return;  // finished, or paused for the subspace pattern (whose gates must stay as they are)
    // the state is read in one sweep, worked on in registers and written back in one sweep (field-by-field accesses through the
    // pointer serialise on each other's latency: 6 us for this one-thread kernel instead of 3)
    LmDev local = *gst;
    LmDev* st = &local;
    struct WriteBack { LmDev* g; LmDev* l; __device__ ~WriteBack() { *g = *l; } } wb{gst, st};
    lm_skip_rest(st);
    if (!st->run_solve) {  // the previous trial was rejected: same model, new radius
        lm_trial_step(st);
        return;
    }
    // fixed-point overflow of the camera sums: the host switches the summation route and repeats (nothing is booked)
    if (h[LMH_FX_BAD] != 0.0) { lm_to_host(st, LM_HOST_FX); return; }
    // the factorisation beside the pair kernel timed out waiting for it (status bit 1: the two kernels did not run at the same time --
    // launches serialised by a tool, two streams on one hardware queue): not a failed factorisation.  Nothing is booked, the damping
    // stays; the host switches the handle to one kernel after the other and resumes the loop at this front (lm_drive)
    // (round-5 advisor: only a wait BETWEEN the two kernels -- status bit 2, raised by c3_wait_arrive and k_unscale's spin -- is handed back; a
    // time-out inside a sequential front, or with several ranks, is a factorisation that did not finish: the escalation branch below)
    if (h[LMH_CHOL_FAIL] >= 4.0) { st->resume_lin = st->run_lin; lm_to_host(st, LM_HOST_BESIDE); return; }
    if (st->run_lin) {
        // bookkeeping of the new linearisation (scipy trf.py:536-546 after an accepted step; :405-426 before the loop)
        const double cost = h[LMH_K_COST];
        if (st->first && !isfinite(cost)) { lm_to_host(st, LM_HOST_NONFINITE); return; }
        st->cost = cost;
        st->g_norm = h[LMH_K_GINF];
        if (st->first) {
            st->Delta = h[LMH_K_DELTA];
            st->initial_cost = cost;
            st->nfev = 1; st->njev = 1;
            st->first = 0;
        } else {
            st->njev += 1;
        }
        st->run_lin = 0;  // booked (a repeated factorisation of the same linearisation must not book it again)
        st->attempts = 0;
        st->reg = h[LMH_K_LAM];
        st->jg_sq = h[LMH_K_JG_SQ];
        // top of the loop (trf.py:450-458)
        if (!st->never_stop) {
            if (st->g_norm < st->gtol) st->status = 1;
            if (st->status != -1 || st->nfev >= st->max_nfev) { st->phase = LM_DONE; st->run_solve = 0; return; }
        }
    }
    // a Cholesky needs a floor where LSMR copes with a numerically singular system: escalate the damping and factorise again
    if (!(h[LMH_CHOL_FAIL] == 0.0) || !isfinite(h[LMH_GRAM_C])) {
        if (++st->attempts > 10) { lm_to_host(st, LM_HOST_CHOL); return; }
        st->reg = fmax(st->reg, 1e-16) * 100.0;
        st->lam_force = st->reg;
        return;  // run_solve stays up: the next tick forms the Schur complement with the forced damping and solves again
    }
    st->run_solve = 0;
    st->lam_force = 0.0;
    const double ga = h[LMH_GRAM_A], gb = h[LMH_GRAM_B], gc = h[LMH_GRAM_C];
    st->ga = ga; st->gb = gb; st->gc = gc;
    const double sa = sqrt(ga), alpha = gb / ga;
    st->sa = sa; st->alpha = alpha;
    const double ww = gc - gb * alpha;
    if (ww > 1e-6 * gc) {
        const double reg = st->reg, nw = sqrt(ww);
        const double m11 = st->jg_sq, m12 = ga - reg * gb, m22 = gb - reg * gc;
        const double b11 = m11 / ga, b12 = (m12 - alpha * m11) / sa, b22 = m22 - 2.0 * alpha * m12 + alpha * alpha * m11;
        st->Ba = b11; st->Bb = b12 / nw; st->Bc = b22 / ww;
        st->gS0 = sa; st->gS1 = 0.0;
        st->nw = nw; st->one_dim = 0;
        lm_trial_step(st);
    } else {
        // degenerate case: q1 = g_h / |g_h|, w = gn_h - alpha g_h explicitly.  Those kernels are not part of the normal pattern: the
        // loop pauses (every queued tick is switched off), the host sees LM_NEED_SUB in the summary and queues the subspace pattern
        st->sub_args[0] = alpha; st->sub_args[1] = 1.0 / sa;
        st->run_sub = 1;
        st->phase = LM_NEED_SUB;
        st->sub_requests += 1;
        st->sub_tick = st->tick + 1;
    }
}

__global__ void k_lm_decide1a(LmDev* __restrict__ gst, const double* __restrict__ h) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    lm_decide1a(gst, h);
}

// The same decision and, in the same launch, what follows it in the normal pattern: the trial point of the cameras and their
// constants there (k_trial_cams; one workgroup: a few hundred cameras at most per thread loop), the exchange header cleared for the
// residual kernel behind.  One launch less per tick -- a tick of a 10-camera problem is 21 launches of 4 - 23 us.
__global__ __launch_bounds__(256) void k_lm_decide1a_trial_cams(LmDev* __restrict__ gst, const double* h, int model, int M, int n_p, int c_p,
                                                                const double* __restrict__ x, const double* __restrict__ v0,
                                                                const double* __restrict__ v1, const double* __restrict__ scale_inv,
                                                                const double* __restrict__ cam_static, double* __restrict__ x_new,
                                                                double* __restrict__ camc_new, double* xb, int hdr_len) {
    if (threadIdx.x == 0) {
        lm_decide1a(gst, h);
        __threadfence();  // the state is in memory before the other waves of the workgroup look at it
    }
    __syncthreads();  // (also: thread 0 is done with the header that is cleared below)
    if (__hip_atomic_load(&gst->run_trial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;  // k_trial_cams' gate
    const double c0 = __hip_atomic_load(&gst->coef[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double c1 = __hip_atomic_load(&gst->coef[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int top = M > hdr_len ? M : hdr_len;
    for (int c = threadIdx.x; c < top; c += blockDim.x) {
        if (c < hdr_len) xb[c] = 0.0;
        if (c >= M) continue;
        double full[11];
        for (int i = 0; i < c_p; ++i) full[i] = cam_static[(size_t)c * c_p + i];
        for (int i = 0; i < n_p; ++i) {
            const size_t k = (size_t)c * n_p + i;
            const double v = x[k] + (c0 * v0[k] + c1 * v1[k]) / scale_inv[k];
            x_new[k] = v;
            full[i] = v;
        }
        cam_constants(model, full, camc_new + (size_t)c * CAMC);
    }
}

// degenerate case, after the explicit subspace vectors: one-dimensional model, or the request for the three products
__global__ void k_lm_decide1b(LmDev* __restrict__ st, const double* __restrict__ h) {
#pragma clang fp contract(off)
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (st->phase != LM_NEED_SUB || !st->run_sub) return;
    st->phase = LM_RUN;  // the subspace pattern is running: the loop carries on behind it
    st->run_sub = 0;
    const double ww = h[LMH_WW];
    if (!(ww > 1e-24 * st->gc && ww > 0)) {  // gn_h parallel to g_h
        st->Ba = st->jg_sq / st->ga; st->Bb = 0.0; st->Bc = 1.0;
        st->gS0 = st->sa; st->gS1 = 0.0;
        st->nw = 1.0; st->one_dim = 1;
        lm_trial_step(st);
    } else {
        st->nw = sqrt(ww);
        st->gS1 = h[LMH_GHW] / st->nw;
        st->Bc = ww;  // parked: |w|^2 for decide1c
        st->run_prod = 1;
    }
}

// degenerate case, after the explicit products |J_h q1|^2, (J_h q1).(J_h w), |J_h w|^2
__global__ void k_lm_decide1c(LmDev* __restrict__ st, const double* __restrict__ h) {
#pragma clang fp contract(off)
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (st->phase != LM_RUN || !st->run_prod) return;
    st->run_prod = 0;
    const double ww = st->Bc;
    st->Ba = h[LMH_B11]; st->Bb = h[LMH_B12] / st->nw; st->Bc = h[LMH_B22] / ww;
    st->gS0 = st->sa;
    st->one_dim = 0;
    lm_trial_step(st);
}

// After the trial evaluation: radius update, accept / reject, termination (trf.py:497-534), and what the next tick does.
// (one thread; cost_new, step_sq, x_sq: the trial header's totals -- not read when no trial ran.  A function of its own: it rides in the
// trial's residual kernel on one rank, TrialArgs::lm_st, and must not take part in that kernel's register allocation)
__device__ __noinline__ void lm_decide2_body(LmDev* gst, double cost_new_in, double step_sq, double x_sq, LmSummary* sum) {
#pragma clang fp contract(off)
    LmDev local = *gst;
    LmDev* st = &local;
    struct WriteBack { LmDev* g; LmDev* l; __device__ ~WriteBack() { *g = *l; } } wb{gst, st};
    if (st->phase != LM_RUN) { st->accept = 0; st->restore = 0; }  // a tick queued behind the end of the loop (or behind a pause)
    if (st->phase == LM_RUN && st->run_trial) {
        const double cost_new = cost_new_in;
        st->cost_new = cost_new;
        st->nfev += 1;
        bool end_inner = false;  // leave scipy's inner loop (while actual_reduction <= 0 and nfev < max_nfev)
        double actual = -1.0;
        if (!isfinite(cost_new)) {
            st->Delta = 0.25 * st->step_h_norm;
        } else {
            actual = st->cost - cost_new;
            double ratio;
            const double Delta_new = satba_lm::update_tr_radius(st->Delta, actual, st->predicted, st->step_h_norm,
                                                                st->step_h_norm > 0.95 * st->Delta, ratio);
            st->step_norm = sqrt(step_sq);
            const int term = st->never_stop ? 0 : satba_lm::check_termination(actual, st->cost, st->step_norm, sqrt(x_sq), ratio, st->ftol, st->xtol);
            if (term) { st->status = term; end_inner = true; }
            else st->Delta = Delta_new;
        }
        if (actual > 0) end_inner = true;
        if (!st->never_stop && st->nfev >= st->max_nfev) end_inner = true;
        if (st->never_stop) end_inner = true;  // fixed work: every tick is a full iteration (bench.py, satba_lm_step's semantics)
        if (end_inner) {
            st->have_actual = 1;
            if (actual > 0) {
                st->actual = actual;
                st->accept = 1;          // x <- x_new, then a new linearisation (which scipy also does before it looks at the status)
                st->run_lin = 1; st->run_solve = 1;
                st->accepted_total += 1;
            } else {
                st->step_norm = 0.0; st->actual = 0.0;
                if (st->never_stop) { st->run_lin = 1; st->run_solve = 1; }  // same x, new radius: a new damped step (satba_lm_step's semantics)
                else st->phase = LM_DONE;  // status set, or max_nfev reached (trf.py:450-458 at the next top of the loop)
            }
            st->iterations += 1;
            if (st->never_stop) {
                // fixed-work runs (bench.py): stop after max_iterations; every cycle_len iterations the solve starts again from the
                // kept point x0 (satba_snapshot_x), the way bench.py restarts the solve that the shipped tolerances end there
                if (st->max_iterations > 0 && st->iterations >= st->max_iterations) {
                    st->phase = LM_DONE;
                    st->run_lin = 0; st->run_solve = 0;  // (the ticks queued behind the end pass empty: they ran up to three more fronts before round 5)
                }
                else if (st->cycle_len > 0 && ++st->cycle_it >= st->cycle_len) {
                    st->cycle_it = 0;
                    st->restore = 1; st->accept = 0;
                    st->first = 1; st->Delta = -1.0; st->run_lin = 1; st->run_solve = 1; st->attempts = 0; st->lam_force = 0.0;
                }
            }
        }
    }
    st->run_trial = 0;
    st->tick += 1;
    if (st->end_tick == 0 && st->phase != LM_RUN && st->phase != LM_NEED_SUB) st->end_tick = st->tick;
    const unsigned long long stamp = (unsigned long long)st->tick << 32;
    __hip_atomic_store(&sum->sub_tick, stamp | ((unsigned long long)st->sub_tick & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&sum->end_tick, stamp | ((unsigned long long)st->end_tick & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&sum->sub_requests, stamp | ((unsigned long long)st->sub_requests & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&sum->word, ((unsigned long long)st->tick << 8) | (unsigned long long)st->phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void k_lm_decide2(LmDev* __restrict__ gst, const double* __restrict__ h, LmSummary* __restrict__ sum) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    lm_decide2_body(gst, h[LMH_COST_NEW], h[LMH_STEP_SQ], h[LMH_X_SQ], sum);
}

// After LM_HOST_BESIDE the host has switched the handle to sequential fronts (and waited for the stream: nothing else touches the
// state): the loop carries on at the front that timed out.
__global__ void k_lm_resume_front(LmDev* __restrict__ st) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    st->phase = LM_RUN; st->host_reason = LM_HOST_NONE; st->end_tick = 0;
    st->run_lin = st->resume_lin; st->run_solve = 1; st->resume_lin = 0;
}

// x <- x_new (and the camera constants, and the cost that bounds the fixed-point camera sums) after an accepted trial; or, at the
// end of a cycle of a fixed-work run, x <- x0, the point kept by satba_snapshot_x (x0: x | bounding box of its points; camc0:
// its camera constants; fxcost0: its cost)
__global__ __launch_bounds__(256) void k_lm_accept(const LmDev* __restrict__ st, long long n, double* __restrict__ x, const double* __restrict__ x_new,
                                                   int n_camc, double* __restrict__ camc, const double* __restrict__ camc_new,
                                                   double* __restrict__ fxcost, const double* __restrict__ fxcost_new,
                                                   const double* __restrict__ x0, const double* __restrict__ camc0,
                                                   const double* __restrict__ fxcost0, double* __restrict__ bbox) {
    if (st->restore != 0 && x0 != nullptr) {
        x_new = x0; camc_new = camc0; fxcost_new = fxcost0;
        if (blockIdx.x == 0 && threadIdx.x < 6) bbox[threadIdx.x] = x0[n + threadIdx.x];
    } else if (st->accept == 0) return;
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
    const long long n2 = n >> 1;  // x and x_new come from hipMalloc: 16-byte aligned
    double2* x2 = reinterpret_cast<double2*>(x);
    const double2* xn2 = reinterpret_cast<const double2*>(x_new);
    for (long long i = i0; i < n2; i += stride) x2[i] = xn2[i];
    if (i0 == 0 && (n & 1)) x[n - 1] = x_new[n - 1];
    for (long long i = i0; i < n_camc; i += stride) camc[i] = camc_new[i];
    if (i0 == 0) *fxcost = *fxcost_new;
}

// k_lm_accept with one more workgroup (the last) that does k_lin_scales' work for the NEXT tick's linearisation (device-resident loop on one
// rank, round 6: one launch per tick less).  The extra workgroup reads the point the loop ends up at from its SOURCE -- x0's record at a
// restart, the trial's at an accepted step, the kept one otherwise -- while the other workgroups copy it, and works only when the decision
// has asked for a linearisation (st->run_lin); the first tick of a run, and every tick behind a hand-over to the host, launch k_lin_scales
// themselves (satba_capi.hip: lm_launch_tick).
struct LinScalesArgs {
    int M, loss, n_clear;
    double w_max, f_scale, n_max, shrink;
    const double* rpc;
    double* fx;
    int* fxe;
    int* fx_flag;
    double* clear;
};
template <int MODEL, int NP>
__global__ __launch_bounds__(256) void k_lm_accept_scales(const LmDev* __restrict__ st, long long n, double* __restrict__ x, const double* __restrict__ x_new,
                                                          int n_camc, double* __restrict__ camc, const double* __restrict__ camc_new,
                                                          double* __restrict__ fxcost, const double* __restrict__ fxcost_new,
                                                          const double* __restrict__ x0, const double* __restrict__ camc0,
                                                          const double* __restrict__ fxcost0, double* __restrict__ bbox, LinScalesArgs q) {
    const bool restore = st->restore != 0 && x0 != nullptr;
    if (blockIdx.x + 1 == gridDim.x) {  // the scales of the next linearisation
        if (st->run_lin == 0) return;
        const bool acc = st->accept != 0;
        lin_scales_body<MODEL, NP>(q.M, restore ? camc0 : (acc ? camc_new : camc), q.rpc, restore ? x0 + n : bbox, q.w_max, q.loss, q.f_scale,
                                   restore ? fxcost0 : (acc ? fxcost_new : fxcost), q.n_max, q.shrink, q.fx, q.fxe, q.fx_flag, q.clear, q.n_clear);
        return;
    }
    if (restore) {
        x_new = x0; camc_new = camc0; fxcost_new = fxcost0;
        if (blockIdx.x == 0 && threadIdx.x < 6) bbox[threadIdx.x] = x0[n + threadIdx.x];
    } else if (st->accept == 0) return;
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)(gridDim.x - 1) * blockDim.x;
    const long long n2 = n >> 1;
    double2* x2 = reinterpret_cast<double2*>(x);
    const double2* xn2 = reinterpret_cast<const double2*>(x_new);
    for (long long i = i0; i < n2; i += stride) x2[i] = xn2[i];
    if (i0 == 0 && (n & 1)) x[n - 1] = x_new[n - 1];
    for (long long i = i0; i < n_camc; i += stride) camc[i] = camc_new[i];
    if (i0 == 0) *fxcost = *fxcost_new;
}

}  // namespace satba
