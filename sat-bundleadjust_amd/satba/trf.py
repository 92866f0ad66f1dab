"""
Trust-region / Levenberg-Marquardt outer loop with scipy's semantics, driving a device engine.

This replaces `scipy.optimize.least_squares(method="trf", tr_solver="lsmr", x_scale="jac")` as the reference
calls it (ref:bundle_adjust/ba_core.py:284-297).  The control flow, trust-radius update, termination tests, nfev
accounting and status codes restate scipy:optimize/_lsq/trf.py:401-560 (`trf_no_bounds`) and
scipy:optimize/_lsq/common.py:171-245, 302-322, 705-717; what changes is where the arithmetic happens:

  * every vector and matrix lives on the device inside an *engine* (satba/engine_hip.py); the host sees a
    handful of scalars per iteration;
  * the Jacobian is analytic and never materialised: the engine forms the normal-equation blocks directly;
  * scipy's inexact LSMR solve of the damped Gauss-Newton system  (J_h^T J_h + reg I) p = g_h  (trf.py:473-480)
    is replaced by an exact one: per-point 3x3 elimination (Schur complement), a dense Cholesky of the reduced
    camera system and back-substitution.  reg is scipy's own Cauchy-step regulariser, so in unscaled variables
    this is the Levenberg-Marquardt system (J^T J + reg diag(scale_inv^2)) x = g;
  * the step is then chosen in span{g_h, gn_h} under the trust radius exactly as scipy does (trf.py:481-496).

Multi-GPU: one engine per rank holds a contiguous shard of points; camera-side sums are combined by an
all-reduce of the engine's exchange buffer after each phase (`comm.allreduce`); every rank runs this same loop
on identical scalars and therefore takes identical decisions.

Engine contract (implemented by engine_hip.HipEngine; tests use a CPU stand-in built on the oracle):
    xb                  exchange buffer, float64 torch tensor: [header (hdr) | payload] (None without torch, 1 GPU)
    hdr, len_lin, len_schur, HDR_FIXED
    configure(loss, f_scale); set_x(x); get_x(); read_header() -> host copy of xb[:hdr]; residuals()
    linearize()         normal blocks at x          -> header[COST], slot[rank] = |g_p|_inf, payload U | g_c
    prepare(first)      scaling, g_h, |J_h g_h|^2   -> header[GH_SQ, JG_SQ, XS_SQ, GC_INF]
    schur(lam)          local reduced system        -> payload S | rhs
    schur_auto(Delta, floor)  the same, lam = scipy's Cauchy-step regulariser computed by the engine from the
                        prepare header (Delta <= 0: scipy's initial radius); no host round trip before it
    solve()             Cholesky + back-substitute  -> header[A, B, C, FAIL]   (Gram matrix of g_h, gn_h) and
                        header[K_COST .. K_DELTA]: what the earlier phases of this iteration reported
    subspace(alpha, s)  q1 = s g_h, w = gn_h - alpha g_h            -> header[WW, WQ1, GHW]
    subspace_products() |J_h q1|^2, (J_h q1).(J_h w), |J_h w|^2      -> header[B11, B12, B22]  (fallback / tests)
    trial(p0, p1)       x_new = x + scale (p0 q1 + p1 w); cost there -> header[COST_NEW, STEP_SQ, X_SQ]
    trial_gn(ca, cb)    x_new = x + scale (ca g_h + cb gn_h): the same step without the subspace phase
    accept()            x <- x_new
"""
import math
import os

import numpy as np

# header slots, per phase (the header is zeroed by each phase before it writes)
COST = 0
GH_SQ, JG_SQ, XS_SQ, GC_INF = 1, 2, 3, 4
GRAM_A, GRAM_B, GRAM_C, CHOL_FAIL = 1, 2, 3, 4
WW, WQ1, B11, B12, B22, GHW = 1, 2, 3, 4, 5, 6
COST_NEW, STEP_SQ, X_SQ = 1, 2, 3
# after solve(): scalars kept from the linearize / prepare / schur_auto phases of the same iteration
K_COST, K_GINF, K_GH_SQ, K_JG_SQ, K_XS_SQ, K_LAM, K_DELTA, K_FX_BAD = 8, 9, 10, 11, 12, 13, 14, 15

TERMINATION_MESSAGES = {
    -1: "Improper input parameters status returned from `leastsq`",
    0: "The maximum number of function evaluations is exceeded.",
    1: "`gtol` termination condition is satisfied.",
    2: "`ftol` termination condition is satisfied.",
    3: "`xtol` termination condition is satisfied.",
    4: "Both `ftol` and `xtol` termination conditions are satisfied.",
}


class SingleComm:
    """No exchange: one engine holds the whole problem."""
    rank, world = 0, 1

    def allreduce(self, engine, n):
        pass

    def allreduce_schur(self, engine):
        pass

    def sum_array(self, a):
        return a

    def gather_array(self, a):
        return a


class TorchComm:
    """Sum all-reduce over torch.distributed (backend nccl == RCCL on ROCm, gloo in the CPU tests)."""

    def __init__(self, group=None, always=False, ordered=False):
        import torch.distributed as dist

        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.always = always  # issue the collectives even for a single rank (plumbing tests)
        self.pack = not os.environ.get("SATBA_NO_PACK")  # packed lower triangle for the Schur all-reduce
        self.pipeline = self.pack and os.environ.get("SATBA_PIPELINE", "1") != "0"  # ... in messages, the factorisation beside it (solve_in_messages)
        self.pipeline_min = int(os.environ.get("SATBA_PIPELINE_MIN", str(500 * 501 // 2)))  # ... from this packed length on (~500 camera unknowns)
        # SATBA_ORDERED_REDUCE=1: sums in RANK ORDER, ((r0 + r1) + r2) + ..., formed by every rank from an all-gather -- the same
        # bits whatever algorithm, topology or channel count the collective library picks (a rank-ordered reduction
        # option).  world x the bytes of the all-reduce it replaces: 32 MB per rank for the packed Schur payload at 8 ranks.
        self.ordered = ordered or bool(os.environ.get("SATBA_ORDERED_REDUCE"))
        self._gather = {}

    def _sum(self, t):
        """In-place sum of tensor t over the ranks (stream ordered like the collective itself)."""
        if not self.ordered:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            return
        import torch

        key = (t.numel(), t.device, t.dtype)
        buf = self._gather.get(key)
        if buf is None:
            buf = self._gather[key] = torch.empty((self.world, t.numel()), dtype=t.dtype, device=t.device)
        self.dist.all_gather(list(buf.unbind(0)), t.reshape(-1), group=self.group)
        flat = t.reshape(-1)
        flat.copy_(buf[0])
        for r in range(1, self.world):
            flat.add_(buf[r])

    def allreduce(self, engine, n):
        """Sum the first n doubles of the engine's exchange buffer over all ranks, in place."""
        if self.world > 1 or self.always:
            self._sum(engine.xb[:n])

    def _messages(self, engine):
        if not (self.pipeline and (self.world > 1 or self.always) and getattr(engine, "schur_messages", None) is not None):
            return []
        msgs = engine.schur_messages()
        # below ~500 camera unknowns the protocol costs more than the overlap can return (one rank, zero network latency, ms per front:
        # 50 cameras x 5 0.369 against 0.309, 200 x 5 1.287 against 1.247 with the launch behind the Schur phase -- profiles/r6_messages.txt)
        return msgs if (msgs and msgs[-1][1] >= self.pipeline_min) else []

    def messages_overlap(self):
        """RCCL queues a collective on the stream and returns: the factorisation can be waiting on the engine's other stream while the
        messages are summed.  gloo (the two-process tests on one GPU) stages device tensors through the host and synchronises the
        DEVICE on the way -- it would wait for the waiting factorisation until that times out (measured: a one-rank gloo group fails
        with an arrival time-out where the one-rank RCCL group is bit-identical to the sequential front)."""
        return self.dist.get_backend(self.group) == "nccl" and not self.ordered

    def solve_in_messages(self, engine, lm_part=None):
        """
        The Schur exchange AND the dense solve of a front, pipelined (round 6): the packed payload is all-reduced in the few messages
        engine.schur_messages() names -- tile columns of the reduced system from the left, the first one with header and right-hand
        side -- while the factorisation, launched first on a stream of the engine's own, takes every tile column when its message has
        landed (csrc: satba_solve_messages_*).  (Launching it even earlier, in front of the Schur kernels as on one rank, was measured and
        dropped: the 64 CUs it holds cost the Schur phase more than the ~25 us of start-up it hides -- profiles/r6_messages.txt.)  Replaces allreduce_schur + engine.solve(): same sums, same
        bits, but the all-reduce (60 - 100 us of a 4 MB payload over xGMI at 8 ranks) and the first tile columns of the factorisation
        no longer wait for each other.  lm_part: the device-resident loop's gated forms (engine.lm_part: parts 10 / 11 / 12).  Returns
        False when the engine solves in one piece (the caller falls back to allreduce_schur + solve).  Every rank issues the same
        collectives in the same order: the message table is a function of (n_cam, n_params) alone.  With a host-blocking backend
        (messages_overlap) the payload is packed and every message summed first, and the same protocol runs back to back behind them:
        same kernels, same bits, no overlap.
        """
        msgs = self._messages(engine)
        if not msgs:
            return False
        overlap = self.messages_overlap()
        engine.pack_schur()
        if not overlap:
            for a, b in msgs:
                self._sum(engine.xp[a:b])
        if lm_part is None:
            engine.solve_messages_begin(packed_already=True)
        else:
            lm_part(10, 1.0)
        for m, (a, b) in enumerate(msgs):
            if overlap:
                self._sum(engine.xp[a:b])
            if lm_part is None:
                engine.solve_messages_arrived(m)
            else:
                lm_part(11, float(m))
        if lm_part is None:
            engine.solve_messages_end()
        else:
            lm_part(12)
        return True

    def allreduce_schur(self, engine):
        """
        Sum the Schur payload (header, reduced camera system, right-hand side) over all ranks.  Engines that can pack the
        lower triangle of the symmetric system (HipEngine) send n_c (n_c + 1) / 2 instead of n_c^2 doubles: the 8 MB
        all-reduce of the headline shape is the largest message of an iteration.
        """
        if not (self.world > 1 or self.always):
            return
        if getattr(engine, "pack_schur", None) is not None and getattr(engine, "xb", None) is not None and self.pack:
            xp = engine.pack_schur()
            self._sum(xp)
            engine.unpack_schur()
        else:
            self.allreduce(engine, engine.len_schur)

    def sum_array(self, a):
        """Element-wise sum of a host float64 array over all ranks (used to assemble sharded results)."""
        import torch

        if self.world == 1:
            return a
        dev = "cuda" if self.dist.get_backend(self.group) == "nccl" else "cpu"
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
        self._sum(t)
        return t.cpu().numpy()

    def gather_array(self, a):
        """Concatenation, in rank order, of the host float64 arrays of all ranks (their lengths may differ): one all-gather of the
        lengths and one of the padded pieces (used to assemble sharded results: every rank moves its own piece, not the whole)."""
        import torch

        if self.world == 1:
            return a
        dev = "cuda" if self.dist.get_backend(self.group) == "nccl" else "cpu"
        a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1)
        n = torch.tensor([a.size], dtype=torch.int64, device=dev)
        sizes = [torch.zeros_like(n) for _ in range(self.world)]
        self.dist.all_gather(sizes, n, group=self.group)
        sizes = [int(s.item()) for s in sizes]
        m = max(sizes)
        piece = torch.zeros(m, dtype=torch.float64, device=dev)
        piece[: a.size] = torch.from_numpy(a).to(dev)
        out = [torch.empty_like(piece) for _ in range(self.world)]
        self.dist.all_gather(out, piece, group=self.group)
        return np.concatenate([o[:k].cpu().numpy() for o, k in zip(out, sizes)])


class Result(dict):
    """Subset of scipy's OptimizeResult: x, cost, fun, optimality, nfev, njev, status, message, success."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


# ----------------------------------------------------------------------------- scalar helpers (scipy semantics)

def solve_trust_region_2d(B, g, Delta):
    """
    min 0.5 p^T B p + g^T p  s.t. |p| <= Delta in two dimensions (scipy:optimize/_lsq/common.py:171-219): the Newton
    step when B is positive definite and the step fits, otherwise the minimiser on the boundary.  scipy finds the
    latter among the real roots of a quartic (numpy.roots, ~60 us per call, with the device idle); the same point is
    the solution of the secular equation |(B + mu I)^-1 g| = Delta, mu >= max(0, -lambda_min(B)), solved here in the
    eigenbasis of B with Newton's method on 1/|p(mu)| - 1/Delta (monotone and concave on that interval), plain
    Python floats.
    """
    a, b, c = float(B[0, 0]), float(B[0, 1]), float(B[1, 1])
    g0, g1 = float(g[0]), float(g[1])
    det = a * c - b * b
    if a > 0 and det > 0:  # positive definite: try the Newton step
        p0, p1 = -(c * g0 - b * g1) / det, -(a * g1 - b * g0) / det
        if p0 * p0 + p1 * p1 <= Delta * Delta:
            return np.array([p0, p1]), True
    # eigen-decomposition B = l1 v1 v1^T + l2 v2 v2^T, l1 <= l2
    h, d = 0.5 * (a + c), 0.5 * (a - c)
    r = math.hypot(d, b)
    l1, l2 = h - r, h + r
    if r == 0.0:
        v1x, v1y = 1.0, 0.0
    elif d > 0:  # v2 = (d + r, b) is well conditioned, v1 is its rotation
        n = math.hypot(d + r, b)
        v1x, v1y = -b / n, (d + r) / n
    else:
        n = math.hypot(d - r, b)
        v1x, v1y = (d - r) / n, b / n
    v2x, v2y = -v1y, v1x
    c1, c2 = g0 * v1x + g1 * v1y, g0 * v2x + g1 * v2y  # g in the eigenbasis
    gn = math.hypot(c1, c2)
    if gn == 0.0:  # no gradient: along the eigenvector of the smallest eigenvalue
        return np.array([Delta * v1x, Delta * v1y]), False
    lo = max(0.0, -l1)
    # hard case: no component along v1 and the interior solution of the remaining direction is short
    tiny = 1e-14 * gn
    if abs(c1) <= tiny and l2 + lo > 0 and abs(c2) / (l2 + lo) < Delta:
        q2 = -c2 / (l2 + lo)
        q1 = math.sqrt(max(Delta * Delta - q2 * q2, 0.0))
        return np.array([q1 * v1x + q2 * v2x, q1 * v1y + q2 * v2y]), False
    # |p(mu)| <= |g| / (l1 + mu): the root lies in [lo, |g| / Delta - l1].  Newton on phi = 1/|p| - 1/Delta from the
    # upper end, safeguarded by the bracket
    mu_lo, mu_hi = lo, max(lo, gn / Delta - l1)
    mu = mu_hi
    for _ in range(100):
        d1, d2 = l1 + mu, l2 + mu
        if d1 <= 0.0:
            mu = 0.5 * (mu_lo + mu_hi)
            continue
        q1, q2 = c1 / d1, c2 / d2
        n2 = q1 * q1 + q2 * q2
        nrm = math.sqrt(n2)
        if nrm > Delta:
            mu_lo = mu
        else:
            mu_hi = mu
        if abs(nrm - Delta) <= 4e-16 * Delta:
            break
        # phi(mu) = 1/nrm - 1/Delta, phi' = (q1^2/d1 + q2^2/d2) / nrm^3
        dphi = (q1 * q1 / d1 + q2 * q2 / d2) / (n2 * nrm)
        step = (1.0 / nrm - 1.0 / Delta) / dphi
        new = mu - step
        if not (mu_lo <= new <= mu_hi) or new == mu:
            new = 0.5 * (mu_lo + mu_hi)
            if new == mu_lo or new == mu_hi:
                break
        mu = new
    d1, d2 = l1 + mu, l2 + mu
    q1, q2 = -c1 / d1, -c2 / d2
    s = Delta / math.hypot(q1, q2)  # exactly on the boundary
    q1, q2 = q1 * s, q2 * s
    return np.array([q1 * v1x + q2 * v2x, q1 * v1y + q2 * v2y]), False


def update_tr_radius(Delta, actual_reduction, predicted_reduction, step_norm, bound_hit):
    """scipy:optimize/_lsq/common.py:222-245."""
    if predicted_reduction > 0:
        ratio = actual_reduction / predicted_reduction
    elif predicted_reduction == actual_reduction == 0:
        ratio = 1
    else:
        ratio = 0
    if ratio < 0.25:
        Delta = 0.25 * step_norm
    elif ratio > 0.75 and bound_hit:
        Delta *= 2.0
    return Delta, ratio


def minimize_quadratic_1d(a, b, lb, ub):
    """Minimum of a t^2 + b t on [lb, ub] (scipy:optimize/_lsq/common.py:302-322)."""
    t = [lb, ub]
    if a != 0:
        ext = -0.5 * b / a
        if lb < ext < ub:
            t.append(ext)
    t = np.asarray(t)
    y = t * (a * t + b)
    i = np.argmin(y)
    return t[i], y[i]


def check_termination(dF, F, dx_norm, x_norm, ratio, ftol, xtol):
    """scipy:optimize/_lsq/common.py:705-717."""
    ftol_ok = dF < ftol * F and ratio > 0.25
    xtol_ok = dx_norm < xtol * (xtol + x_norm)
    if ftol_ok and xtol_ok:
        return 4
    if ftol_ok:
        return 2
    if xtol_ok:
        return 3
    return None


def _print_header():
    print("{:^15}{:^15}{:^15}{:^15}{:^15}{:^15}".format(
        "Iteration", "Total nfev", "Cost", "Cost reduction", "Step norm", "Optimality"))


def _print_iteration(iteration, nfev, cost, cost_reduction, step_norm, optimality):
    cr = " " * 15 if cost_reduction is None else f"{cost_reduction:^15.2e}"
    sn = " " * 15 if step_norm is None else f"{step_norm:^15.2e}"
    print(f"{iteration:^15}{nfev:^15}{cost:^15.4e}{cr}{sn}{optimality:^15.2e}")


def subspace_model(engine, exchange, ga, gb, gc, jg_sq, reg):
    """
    Quadratic model on the orthonormal basis {q1, q2} = {g_h / |g_h|, w / |w|}, w = gn_h - (b / a) g_h, of
    span{g_h, gn_h} (scipy trf.py:481-485): returns B_S (2 x 2), g_S and (ca, cb) -> coefficients, i.e. a function
    mapping a step p_S on that basis to its coefficients on (g_h, gn_h).
    a, b, c = Gram matrix of (g_h, gn_h).  Because gn_h solves the damped normal equations exactly,
    J_h^T J_h gn_h = g_h - reg gn_h, the Gram matrix of (J_h g_h, J_h gn_h) is known without touching the
    observations again:  |J_h g_h|^2 = jg_sq (from the prepare phase), (J_h g_h).(J_h gn_h) = a - reg b,
    |J_h gn_h|^2 = b - reg c; and |w|^2 = c - b^2 / a, g_h.w = 0.  Only when w is a tiny remainder of gn_h (the two
    directions nearly parallel: cancellation in these formulas) are w and the products formed explicitly on the
    device (subspace, subspace_products: two more passes and round trips).
    """
    sa = np.sqrt(ga)
    alpha = gb / ga
    ww = gc - gb * alpha
    if ww > 1e-6 * gc:
        nw = np.sqrt(ww)
        m11, m12, m22 = jg_sq, ga - reg * gb, gb - reg * gc
        b11 = m11 / ga
        b12 = (m12 - alpha * m11) / sa
        b22 = m22 - 2.0 * alpha * m12 + alpha * alpha * m11
        ghw = 0.0
    else:
        engine.subspace(alpha, 1.0 / sa)
        h = exchange(engine.hdr)
        ww, ghw = h[WW], h[GHW]
        if not (ww > 1e-24 * gc and ww > 0):  # gn_h parallel to g_h: one-dimensional subspace
            return np.array([[jg_sq / ga, 0.0], [0.0, 1.0]]), np.array([sa, 0.0]), lambda p_S: (p_S[0] / sa, 0.0)
        nw = np.sqrt(ww)
        engine.subspace_products()
        hp = exchange(engine.hdr)
        b11, b12, b22 = hp[B11], hp[B12], hp[B22]
    B_S = np.array([[b11, b12 / nw], [b12 / nw, b22 / ww]])
    return B_S, np.array([sa, ghw / nw]), lambda p_S: (p_S[0] / sa - p_S[1] * alpha / nw, p_S[1] / nw)


# ----------------------------------------------------------------------------- the loop on the device, several ranks

LM_RUN, LM_DONE, LM_NEED_HOST, LM_NEED_SUB = 0, 1, 2, 3
LM_HOST_FX, LM_HOST_CHOL, LM_HOST_NONFINITE, LM_HOST_BESIDE = 1, 2, 3, 4  # LmDev::host_reason (csrc/satba_lmdev.h)
LM_RUN_AHEAD = 3  # csrc/satba_lmdev.h


def drive_device_loop(engine, comm, lam_floor=0.0, max_patterns=None, watchdog_s=60.0):
    """
    Queue the launch patterns of the device-resident loop (csrc/satba_lmdev.h) for a sharded problem: the parts of a tick
    (engine.lm_part) with the all-reduces of the exchange buffer between them, all on one stream, nothing waits for the device --
    the decisions between the parts are one-thread kernels on all-reduced scalars, the same on every rank.

    Every rank must issue the SAME sequence of collectives, so the host does not act on "the latest report it happens to see":
    pattern i is queued once the device has reported pattern i - 1 - LM_RUN_AHEAD, and what pattern i is -- a tick, the
    degenerate-subspace pattern, or nothing because the loop has ended -- follows from two tick numbers in that report which are
    functions of all-reduced scalars only: the tick of the latest pause (the subspace pattern is pattern number pause +
    LM_RUN_AHEAD + 1) and the tick at which the loop left the running phase (end + LM_RUN_AHEAD patterns are queued in all).
    Returns the number of patterns queued.  Call engine.lm_begin first; engine.lm_state() afterwards waits for the stream.
    """
    import time

    hdr = engine.hdr
    queued, served = 0, 0

    def tick():
        engine.lm_part(0, lam_floor)
        comm.allreduce(engine, engine.len_lin)
        engine.lm_part(1)
        comm.allreduce(engine, hdr)
        engine.lm_part(2, lam_floor)
        if not (getattr(comm, "solve_in_messages", None) and comm.solve_in_messages(engine, engine.lm_part)):
            comm.allreduce_schur(engine)
            engine.lm_part(3)
        comm.allreduce(engine, hdr)
        engine.lm_part(4)
        comm.allreduce(engine, hdr)
        engine.lm_part(5)

    def sub_pattern():
        for part in (6, 7, 8):
            engine.lm_part(part)
            comm.allreduce(engine, hdr)
        engine.lm_part(9)

    while True:
        t0 = time.perf_counter()
        while True:
            done, phase, subreq, sub_tick, end_tick = engine.lm_poll()
            if done >= queued - LM_RUN_AHEAD:
                break
            if time.perf_counter() - t0 > watchdog_s:
                raise RuntimeError("device-resident loop: no progress report from the device for {} s".format(watchdog_s))
        if end_tick and queued >= end_tick + LM_RUN_AHEAD:
            return queued
        if max_patterns is not None and queued > max_patterns:
            raise RuntimeError("device-resident loop: {} launch patterns queued without reaching the end".format(queued))
        if subreq > served and queued == sub_tick + LM_RUN_AHEAD:
            sub_pattern()
            served = subreq
        else:
            tick()
        queued += 1


def trf_solve_sharded(engine, comm, ftol, xtol, gtol, max_nfev, loss, f_scale):
    """The loop for several ranks with its decisions on the device (drive_device_loop).  Returns (Result, state); the Result is None
    when the device handed over to the host -- state["host_reason"]: overflow of the fixed-point camera sums (the caller switches
    the summation route and continues with the host loop from the point the device stopped at, carrying state's counters), a
    factorisation that failed ten times, non-finite residuals in the initial point (errors, as in the host loop)."""
    engine.lm_begin(ftol=ftol, xtol=xtol, gtol=gtol, max_nfev=max_nfev, loss=loss, f_scale=f_scale)
    nmax = max_nfev if max_nfev is not None else engine.n_total * 100
    drive_device_loop(engine, comm, 0.0, max_patterns=24 * nmax + 1000)
    st = engine.lm_state()
    if int(st["phase"]) != LM_DONE:
        return None, st
    status = int(st["status"]) if int(st["status"]) >= 0 else 0
    return Result(cost=st["cost"], optimality=st["g_norm"], nfev=int(st["nfev"]), njev=int(st["njev"]), status=status,
                  success=status > 0, message=TERMINATION_MESSAGES[status], iterations=int(st["iterations"]),
                  initial_cost=st["initial_cost"]), st


# ----------------------------------------------------------------------------- the loop

def trf_solve(engine, comm=None, ftol=1e-8, xtol=1e-8, gtol=1e-8, max_nfev=None, loss="linear", f_scale=1.0,
              verbose=0, timers=None, native=None):
    """
    Run the loop on `engine` from its current x.  Returns a Result; the solution stays in the engine
    (engine.get_x(), engine.residuals()).  `timers`, if a dict, accumulates per-phase call counts.

    native: run the same loop below the C ABI (satba_solve_lm, csrc/satba_capi.hip) instead of in Python -- the default for a
    single rank when the engine offers it and no iteration table is requested (verbose < 2); with several ranks the
    all-reduces between the phases are issued from here.
    """
    comm = comm or SingleComm()
    resume = None
    if native is None:
        # (the per-phase counts of `timers` only exist in this Python loop)
        native = comm.world == 1 and not getattr(comm, "always", False) and verbose < 2 and hasattr(engine, "solve_lm") and timers is None
    if native:
        st = engine.solve_lm(ftol=ftol, xtol=xtol, gtol=gtol, max_nfev=max_nfev, loss=loss, f_scale=f_scale, verbose=0)
        res = Result(cost=st.cost, optimality=st.optimality, nfev=int(st.nfev), njev=int(st.njev), status=int(st.status),
                     success=st.status > 0, message=TERMINATION_MESSAGES[int(st.status)], iterations=int(st.iterations),
                     initial_cost=st.initial_cost)
        if verbose >= 1:
            print(res.message)
            print("Function evaluations {}, initial cost {:.4e}, final cost {:.4e}, first-order optimality {:.2e}."
                  .format(res.nfev, res.initial_cost, res.cost, res.optimality))
        return res
    if isinstance(comm, TorchComm) and (comm.world > 1 or comm.always) and hasattr(engine, "use_torch_stream"):
        engine.use_torch_stream()  # the all-reduces are queued on torch's stream: the phases must order with them
        # several ranks: the decisions of the loop on the device, the host only queues (no header read inside an iteration);
        # SATBA_HOST_LOOP=1 or an iteration table (verbose 2) keep the host loop below, and so does a hand-over by the device
        if (hasattr(engine, "lm_part") and verbose < 2 and timers is None and not os.environ.get("SATBA_HOST_LOOP")
                and engine.n_c <= 1024):
            engine.configure(loss, f_scale)
            res, dev_state = trf_solve_sharded(engine, comm, ftol, xtol, gtol, max_nfev, loss, f_scale)
            if res is not None:
                if verbose >= 1:
                    print(res.message)
                    print("Function evaluations {}, initial cost {:.4e}, final cost {:.4e}, first-order optimality {:.2e}."
                          .format(res.nfev, res.initial_cost, res.cost, res.optimality))
                return res
            # the device handed over (all ranks stop for the same reason: it follows from all-reduced headers)
            reason = int(dev_state["host_reason"])
            if reason == LM_HOST_NONFINITE:
                raise ValueError("Residuals are not finite in the initial point.")
            if reason == LM_HOST_CHOL:
                raise RuntimeError("reduced camera system could not be factorised")
            if reason == LM_HOST_BESIDE:
                # a front is only handed back when a wait BETWEEN the factorisation and the pair kernel running beside it timed out
                # (status bit 2, csrc/satba_lmdev.h: lm_decide1a); sharded fronts run one kernel after the other
                raise RuntimeError("device loop handed back a concurrent front on a sharded run")
            if reason == LM_HOST_FX and getattr(engine, "camera_sums_fallback", None):
                engine.camera_sums_fallback()  # only this reason changes the summation route
            if int(dev_state["nfev"]) > 0:
                resume = dev_state  # the work done on the device counts: evaluations, trust radius, initial cost
    hdr = engine.hdr
    slots = slice(engine.HDR_FIXED, engine.HDR_FIXED + comm.world)
    if max_nfev is None:
        max_nfev = engine.n_total * 100

    def exchange(n):
        comm.allreduce(engine, n)
        return engine.read_header()

    def front(Delta):
        """
        linearize -> prepare -> damped Gauss-Newton step at the current x, queued without a host round trip: the
        engine derives the damping from the trust radius itself (Delta None: first call, scipy's initial radius).
        One header read returns what scipy computes at the top of an iteration (cost, |g|_inf, ...) together with
        the Gram matrix of the step; scipy's gtol / max_nfev tests are taken right after it.
        """
        while True:
            engine.linearize()
            comm.allreduce(engine, engine.len_lin)
            engine.prepare(Delta is None)
            comm.allreduce(engine, hdr)
            engine.schur_auto(-1.0 if Delta is None else Delta, 0.0)
            if not (getattr(comm, "solve_in_messages", None) and comm.solve_in_messages(engine)):
                comm.allreduce_schur(engine)
                engine.solve()
            h = exchange(hdr)
            # K_FX_BAD (summed over the ranks): a term of some shard's fixed-point camera sums left its range -- every rank
            # switches to the camera-major sums and repeats the iteration (include/satba.h, SATBA_HDR_FX_BAD)
            # (the decision depends on the all-reduced flag only: every rank switches and repeats together, whatever route its own
            # handle was on -- a rank that returned here while the others loop again would leave their collectives unmatched)
            if h[K_FX_BAD] == 0 or not getattr(engine, "camera_sums_fallback", None):
                return h
            engine.camera_sums_fallback()

    engine.configure(loss, f_scale)
    if resume is None:
        h = front(None)
        cost, g_norm, Delta = h[K_COST], h[K_GINF], h[K_DELTA]
        if not np.isfinite(cost):
            raise ValueError("Residuals are not finite in the initial point.")
        nfev = njev = 1
        initial_cost = cost
    else:
        # continue where the device-resident loop stopped: the linearisation it could not book (at the device's current x) is
        # repeated here with the device's trust radius; x_scale's running maximum lives on the device and carries over by itself
        Delta = float(resume["Delta"])
        h = front(Delta)
        cost, g_norm = h[K_COST], h[K_GINF]
        nfev, njev, initial_cost = int(resume["nfev"]), int(resume["njev"]) + 1, float(resume["initial_cost"])

    status, iteration, step_norm, actual_reduction = None, 0, None, None
    lm_iterations = 0 if resume is None else int(resume["iterations"])
    if verbose == 2:
        _print_header()

    while True:
        if g_norm < gtol:
            status = 1
        if verbose == 2:
            _print_iteration(iteration, nfev, cost, actual_reduction, step_norm, g_norm)
        if status is not None or nfev == max_nfev:
            break

        # h: header of the solve phase for the current x and Delta.  reg = Cauchy-step regulariser of the
        # Gauss-Newton system (scipy trf.py:473-477), computed by the engine.  In scaled variables the system matrix
        # has a unit diagonal, so reg is relative to 1.  Where scipy's LSMR copes with a numerically singular system
        # (gauge freedom, flat valleys), a Cholesky needs a floor: escalate the damping until the factorisation
        # goes through.
        reg, jg_sq = h[K_LAM], h[K_JG_SQ]
        for attempt in range(10):
            if h[CHOL_FAIL] == 0 and np.isfinite(h[GRAM_C]):
                break
            reg = max(reg, 1e-16) * 100.0
            engine.schur(reg)
            if not (getattr(comm, "solve_in_messages", None) and comm.solve_in_messages(engine)):
                comm.allreduce_schur(engine)
                engine.solve()
            h = exchange(hdr)
        else:
            raise RuntimeError("reduced camera system could not be factorised")
        ga, gb, gc = h[GRAM_A], h[GRAM_B], h[GRAM_C]

        # the model restricted to span{g_h, gn_h} (scipy trf.py:481-485)
        B_S, g_S, coeffs = subspace_model(engine, exchange, ga, gb, gc, jg_sq, reg)

        actual_reduction = -1
        while actual_reduction <= 0 and nfev < max_nfev:
            p_S, _ = solve_trust_region_2d(B_S, g_S, Delta)
            predicted_reduction = -(0.5 * p_S @ B_S @ p_S + g_S @ p_S)
            engine.trial_gn(*coeffs(p_S))
            ht = exchange(hdr)
            cost_new = ht[COST_NEW]
            nfev += 1
            step_h_norm = np.linalg.norm(p_S)
            if not np.isfinite(cost_new):
                Delta = 0.25 * step_h_norm
                continue
            actual_reduction = cost - cost_new
            Delta_new, ratio = update_tr_radius(Delta, actual_reduction, predicted_reduction, step_h_norm,
                                                step_h_norm > 0.95 * Delta)
            step_norm = np.sqrt(ht[STEP_SQ])
            status = check_termination(actual_reduction, cost, step_norm, np.sqrt(ht[X_SQ]), ratio, ftol, xtol)
            if status is not None:
                break
            Delta = Delta_new

        if actual_reduction > 0:
            engine.accept()
            h = front(Delta)
            cost, g_norm = h[K_COST], h[K_GINF]
            njev += 1
        else:
            step_norm = 0
            actual_reduction = 0
            # same x, smaller radius: a new damped step for it (scipy recomputes it at the top of the next iteration)
            if status is None and nfev < max_nfev:
                h = front(Delta)
        iteration += 1
        lm_iterations += 1

    if status is None:
        status = 0
    res = Result(cost=cost, optimality=g_norm, nfev=nfev, njev=njev, status=status, success=status > 0,
                 message=TERMINATION_MESSAGES[status], iterations=lm_iterations, initial_cost=initial_cost)
    if verbose >= 1:
        print(res.message)
        print("Function evaluations {}, initial cost {:.4e}, final cost {:.4e}, first-order optimality {:.2e}."
              .format(nfev, initial_cost, cost, g_norm))
    return res
