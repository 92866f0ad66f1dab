"""
bench.py -- Levenberg-Marquardt iterations per second of the device bundle-adjustment solver on synthetic
tracks of BASELINE.json's headline shape (200 cameras x 1 M points x ~10 M observations, affine, R+T).

    python bench.py --gpus 1 --steps 200 --warmup 10
    python bench.py --gpus N ...            (starts N rank processes itself, as fresh children, before anything touches the GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step is ONE LM iteration of the whole problem: one linearisation (residual + analytic Jacobian -> normal
blocks), the x_scale="jac" update and Cauchy-step regulariser, one damped solve (Schur complement, dense
Cholesky, back-substitution), the 2-D subspace model and one trial-point evaluation with the accept / reject
decision -- exactly the phases satba/trf.py runs per iteration, with the same all-reduces and the same host
readbacks.  Inputs are resident in HBM before the timed region.  With N > 1 the SAME problem is sharded by
points over the ranks (strong scaling); value = steps / max-over-ranks wall time.

One JSON line is printed by rank 0; see DESIGN.md for `roofline` (fused residual+Jacobian kernel, algorithmic
bytes 48 K + 96 N per launch) and `cpu_baseline` (the reference's scipy path restated in oracle/, timed on this host:
the full solve at the C2 shape and one LM iteration on a bounded sub-problem of the benchmarked scene, scaled linearly in
the number of observations -- a lower bound on the CPU time, LSMR needs more iterations on larger problems; the measured
C3 run of BASELINE.md section 3 is quoted from profiles/ when it has been taken).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for path in (os.path.join(ROOT, "sat-bundleadjust_amd"), ROOT):
    if path not in sys.path:
        sys.path.insert(0, path)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling

# HBM bytes per k_linearize launch from rocprofv3 PMC passes of this same command (profiles/r5_pmc_hbm_traffic.json, from the
# summaries of tools/pmc_summary.py): 2 x FETCH_SIZE (gfx950 counts 128-B requests as 64 B, MI355X_MICROARCH.md section HBM) + WRITE_SIZE.
# PMC counters cannot be read from inside an unprofiled run, so the committed measurement is quoted for the (shape, loss) --
# i.e. the kernel instantiation -- it was taken on (one GPU), and the field is null otherwise.
def profiled_traffic(shape, loss, world):
    path = os.path.join(ROOT, "profiles", "r6_pmc_hbm_traffic.json")
    if world != 1 or not os.path.exists(path):
        return None
    with open(path) as fh:
        return json.load(fh).get("{}:{}".format(shape, loss))


def compulsory_bytes(K, N, model, n_params, loss, unit_weights, fused_prepare):
    """
    HBM bytes k_linearize has to move AS BUILT (csrc/satba_kernels.h), per launch: per observation the camera index (4) and the
    observed pixel (16), the weight (8) unless every weight is 1 and the loss is linear, the Jacobian row scales it stores for the
    later passes (16) on weighted / robust runs, and for RPC cameras the stored Jacobian rows (128 B for 3, 192 B for 6 parameters);
    per point x (24) in, V (48) and g_p (24) out, and -- single-rank loops, where the point part of the prepare phase rides in this
    kernel -- scale_inv (24) in, scale_inv, g_h and g_h / scale_inv (72) out.  The per-camera sums stay in LDS; the residuals are not
    stored.
    """
    unit = unit_weights and loss == "linear"
    per_obs = 20 + (0 if unit else 8 + 16)
    if model == "rpc":
        per_obs += 8 * (16 if 2 * n_params + 6 <= 16 else 24)
    return float(per_obs) * K + (96.0 + (96.0 if fused_prepare else 0.0)) * N


class PhaseProfile:
    """Per-phase device time of the multi-rank iteration (torch events on the stream the phases and the collectives are queued on)
    and the host's blocking header reads (wall clock): what `phase_ms` of the bench line is made of."""

    def __init__(self, torch):
        self.torch, self.ev, self.host = torch, [], {}

    def dev(self, name, fn):
        a, b = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
        a.record()
        out = fn()
        b.record()
        self.ev.append((name, a, b))
        return out

    def wait(self, name, fn):
        t0 = time.perf_counter()
        out = fn()
        self.host[name] = self.host.get(name, 0.0) + 1e3 * (time.perf_counter() - t0)
        return out

    def result(self, iterations):
        self.torch.cuda.synchronize()
        tot = {}
        for name, a, b in self.ev:
            tot[name] = tot.get(name, 0.0) + a.elapsed_time(b)
        out = {k: v / iterations for k, v in tot.items()}
        out.update({k: v / iterations for k, v in self.host.items()})
        return out


def lm_step(eng, comm, st, trf, prof=None):
    """One fixed-work LM iteration (see module docstring).  `st` carries cost, Delta between steps.  prof: a PhaseProfile (a separate,
    untimed pass of the bench: the events serialise nothing, but they are not part of the timed region)."""
    hdr = eng.hdr
    dev = prof.dev if prof else (lambda name, fn: fn())
    wait = prof.wait if prof else (lambda name, fn: fn())

    def exchange(n):
        dev("allreduce", lambda: comm.allreduce(eng, n))
        return wait("host_wait", eng.read_header)

    # linearize -> prepare -> damped Gauss-Newton step, queued without a host round trip (satba/trf.py: front)
    dev("linearize", eng.linearize)
    dev("allreduce", lambda: comm.allreduce(eng, eng.len_lin))
    dev("prepare", lambda: eng.prepare(st["first"]))
    dev("allreduce", lambda: comm.allreduce(eng, hdr))
    dev("schur", lambda: eng.schur_auto(-1.0 if st["first"] else st["Delta"], 1e-14))
    dev("allreduce_schur", lambda: comm.allreduce_schur(eng))
    dev("dense_solve", eng.solve)
    h = exchange(hdr)
    st["first"] = False
    cost, Delta, reg, jg_sq = h[trf.K_COST], h[trf.K_DELTA], h[trf.K_LAM], h[trf.K_JG_SQ]
    ga, gb, gc_ = h[trf.GRAM_A], h[trf.GRAM_B], h[trf.GRAM_C]
    B_S, g_S, coeffs = trf.subspace_model(eng, exchange, ga, gb, gc_, jg_sq, reg)
    p_S, newton = trf.solve_trust_region_2d(B_S, g_S, Delta)
    st["interior"] = st.get("interior", 0) + int(newton)
    predicted = -(0.5 * p_S @ B_S @ p_S + g_S @ p_S)
    dev("trial", lambda: eng.trial_gn(*coeffs(p_S)))
    h = exchange(hdr)
    cost_new = h[trf.COST_NEW]
    step_h_norm = np.linalg.norm(p_S)
    actual = cost - cost_new if np.isfinite(cost_new) else -1.0
    st["Delta"], _ = trf.update_tr_radius(Delta, actual, predicted, step_h_norm, step_h_norm > 0.95 * Delta)
    if actual > 0:
        dev("accept", eng.accept)
        st["accepted"] += 1
    st["cost"] = cost_new if actual > 0 else cost


def lm_step_native(eng, st):
    """The same iteration with the host side in C++ (satba_lm_step): what a single-rank solve runs (satba_solve_lm's loop body)."""
    r = eng.lm_step(st["first"], st.get("Delta", -1.0), 1e-14)
    st["first"] = False
    st["Delta"] = r["Delta"]
    st["lam"] = r["lam"]
    st["interior"] = st.get("interior", 0) + int(r["newton"])
    st["accepted"] += int(r["accepted"])
    st["cost"] = r["cost_new"] if r["accepted"] else r["cost"]


def cpu_baseline(scene, n_pts_sample, corr, full_c3=False):
    """
    The reference's scipy path (oracle/ba_oracle.solve_scipy == ref ba_core.py:284-297 on the numpy restatement of fun),
    BASELINE.md section 3:
      * the FULL solve at the C2 shape (10 x 5 k x 30 k, shipped tolerances): LM it/s = (nfev - 1) / wall;
      * one LM iteration (max_nfev 2: one finite-difference Jacobian, one LSMR solve, one trial) on the first n_pts_sample
        points of the benchmarked scene, scaled linearly in the number of observations to the full workload -- both the FD
        Jacobian and an LSMR iteration are linear in nnz, but LSMR needs MORE iterations on larger problems (290 per LM step at
        C2, ~2500 at C3), so this is a lower bound on the CPU time;
      * --cpu-c3: the measured C3 run (50 x 100 k x 1 M, max_nfev 3, ~10 min) of BASELINE.md; its result is kept in
        profiles/r2_cpu_baseline_C3.json and quoted by later runs, with the C4 figure extrapolated from it (x K ratio) and
        labelled as such.
    """
    from oracle import ba_oracle as O
    from satba import synth
    from satba.ba_params import BundleAdjustmentParameters

    out = {"unit": "LM iters/sec", "cores": 1, "kind": "port", "host_cores": os.cpu_count()}
    # C2, full solve
    model, c2corr, n_cam, n_pts, opp = synth.CONFIGS["C2"]
    sc2 = synth.make_scene(model, n_cam, n_pts, opp, seed=1, sigma_theta=2e-6)
    p2 = synth.make_params(sc2, {"correction_params": c2corr, "n_cam_fix": 1})
    t0 = time.perf_counter()
    r2 = O.solve_scipy(p2, {"verbose": 0})
    t2 = time.perf_counter() - t0
    out["C2_full_solve"] = {"seconds": t2, "nfev": int(r2.nfev), "lm_iters_per_sec": max(int(r2.nfev) - 1, 1) / t2, "n_obs": int(p2.n_obs)}
    # bounded sample of the benchmarked scene
    keep = scene.pts_ind < n_pts_sample
    p = BundleAdjustmentParameters.from_observations(
        scene.pts_ind[keep], scene.cam_ind[keep], scene.pts2d[keep], scene.pts3d[:n_pts_sample], scene.cameras,
        scene.cam_model, scene.pairs_to_triangulate, scene.camera_centers,
        {"verbose": False, "correction_params": corr, "n_cam_fix": 1})
    t0 = time.perf_counter()
    res = O.solve_scipy(p, {"verbose": 0}, max_nfev=2)
    dt = time.perf_counter() - t0
    full = scene.pts_ind.size
    t_full = dt * full / p.n_obs
    out.update({"value": 1.0 / t_full, "seconds_on_sample": dt, "nfev": int(res.nfev),
                "sample": "scipy least_squares(trf, lsmr, 2-point FD Jacobian) max_nfev=2 (1 LM iteration) on the first {} points / {} obs "
                          "of the benchmarked scene: {:.1f} s, scaled linearly to {} obs (a lower bound on the CPU time: LSMR "
                          "iterations grow with the problem); the scipy path uses one of the host's {} cores; full C2 solve "
                          "{:.1f} s / nfev {}".format(n_pts_sample, p.n_obs, dt, full, os.cpu_count(), t2, int(r2.nfev))})
    import glob
    c3_files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_cpu_baseline_C3.json")), key=lambda f: int(os.path.basename(f)[1:].split("_")[0]))
    c3_path = c3_files[-1] if c3_files else os.path.join(ROOT, "profiles", "r2_cpu_baseline_C3.json")  # the latest round's measured run
    if full_c3:
        model, c3corr, n_cam, n_pts, opp = synth.CONFIGS["C3"]
        sc3 = synth.make_scene(model, n_cam, n_pts, opp, seed=1, sigma_theta=1e-4)
        p3 = synth.make_params(sc3, {"correction_params": c3corr, "n_cam_fix": 1})
        t0 = time.perf_counter()
        r3 = O.solve_scipy(p3, {"verbose": 0}, max_nfev=3)
        t3 = time.perf_counter() - t0
        c3 = {"seconds": t3, "nfev": int(r3.nfev), "lm_iters": int(r3.nfev) - 1, "lm_iters_per_sec": (int(r3.nfev) - 1) / t3,
              "n_obs": int(p3.n_obs), "host_cores": os.cpu_count()}
        out["C3_measured"] = c3
        try:
            os.makedirs(os.path.dirname(c3_path), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "cpu_baseline_C3.json"), "w") as fh:  # (copied to profiles/rN_cpu_baseline_C3.json by hand)
                json.dump(c3, fh)
        except OSError:
            pass
    elif os.path.exists(c3_path):
        with open(c3_path) as fh:
            out["C3_measured"] = dict(json.load(fh), quoted_from="profiles/" + os.path.basename(c3_path) + " (taken with --cpu-c3 on an MI355X box's host)")
    if "C3_measured" in out and full > 0:
        c3 = out["C3_measured"]
        out["C4_extrapolated_from_C3"] = {"lm_iters_per_sec": c3["lm_iters_per_sec"] * c3["n_obs"] / full,
                                          "note": "measured C3 rate x (K_C3 / K): linear in observations, LSMR iteration count of C3"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--shape", default="C4", help="C2 | C3 | C4 (affine R+T) | C5 (rpc R) | P3 (perspective R+T); SURVEY.md section 8d")
    ap.add_argument("--restart-every", type=int, default=-1,
                    help="iterations per solve before the point returns to x0 (-1: 3 for the linear loss on one rank, else 0 = never)")
    ap.add_argument("--sigma-theta", type=float, default=1e-4, help="initial camera angle error [rad]")
    ap.add_argument("--cpu-sample-pts", type=int, default=12000,
                    help="points of the CPU-baseline sub-problem (0 = skip); 12000 points = 120 k observations, ~25 s on the GPU box's host "
                         "(20000 points took 45 s there)")
    ap.add_argument("--kernel-reps", type=int, default=20)
    ap.add_argument("--loss", default="linear", help="linear (headline) | soft_l1 | huber | cauchy | arctan")
    ap.add_argument("--driver", default="auto", help="host side of an LM iteration: native (device-resident loop, satba_lm_ticks; one rank) | device (the same for several ranks: tick parts + queued all-reduces) | "
                    "native-sync (C++ host loop, satba_lm_step: two header reads per iteration) | python (phase entry points + all-reduces) | "
                    "auto (native for one rank, device for several)")
    ap.add_argument("--cpu-c3", action="store_true", help="also run the measured C3 CPU baseline (max_nfev=3, ~10 min)")
    ap.add_argument("--backend", default="nccl", help="collective backend for several ranks: nccl (= RCCL; what the scaling runs use) | gloo (tests)")
    ap.add_argument("--share-gpu", action="store_true", help="tests: all ranks on GPU 0 (a one-GPU box; needs --backend gloo)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the wall clock of the drop-in call (ba_core.run_ba_optimization) after the timed loop")
    ap.add_argument("--camera-major", action="store_true", help="form the per-camera sums with the camera-major float64 pass from the "
                    "start (SATBA_FLAG_CAMERA_MAJOR_SUMS: the route the fixed-point sums fall back to)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks as fresh child processes.  Decided before torch is imported or
        # anything touches HIP in this process (a process that has initialised the GPU must never be replaced or forked into
        # the ranks); this parent only waits and passes the exit code on.
        import socket
        import subprocess

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.run(cmd, env=env).returncode)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus {} but {} rank(s) were launched".format(args.gpus, world))
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    from satba import sharding, synth, trf
    from satba.engine_hip import HipEngine

    model, corr, n_cam, n_pts, opp = synth.CONFIGS[args.shape]
    t_gen = time.perf_counter()
    scene = synth.make_scene(model, n_cam, n_pts, opp, seed=1, sigma_theta=args.sigma_theta if model != "rpc" else 1e-6)
    p = synth.make_params(scene, {"correction_params": corr, "n_cam_fix": 1})
    t_gen = time.perf_counter() - t_gen
    comm = trf.TorchComm() if world > 1 else trf.SingleComm()
    eng = HipEngine(p, sharding.make_shard(p, rank, world), deterministic=args.camera_major)
    eng.configure(args.loss, 1.0)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # single rank: the host side of an iteration is the library's C++ (satba_lm_step, the loop body of satba_solve_lm, which is
    # what ba_core's solve runs); several ranks: the Python phases with the all-reduces between them
    # several ranks: "device" = the device-resident loop in parts with the all-reduces queued between them (trf.drive_device_loop: no
    # host wait inside an iteration), "python" = the host loop with two blocking header reads per iteration (rounds 1-3)
    driver = args.driver if args.driver != "auto" else ("native" if world == 1 else ("python" if os.environ.get("SATBA_HOST_LOOP") else "device"))
    if driver.startswith("native") and world > 1:
        raise SystemExit("--driver native drives one rank")
    if driver == "device" and world == 1:
        raise SystemExit("--driver device is the several-rank form of --driver native")
    if world > 1:
        eng.use_torch_stream()
    step = (lambda: lm_step_native(eng, st)) if driver.startswith("native") else (lambda: lm_step(eng, comm, st, trf))
    st = {"first": True, "accepted": 0, "fail": 0, "cost": None}
    # Every timed step is an iteration a real solve performs.  The solve from x0 under the shipped tolerances (ftol 1e-4, xtol 1e-10,
    # max_iter 300: ba_core.init_optimization_config) is run once, untimed, and takes `restart` LM iterations (linear loss at the
    # headline shape: 2.3e12 -> 876443 -> 764749 -> 764502.09 -> no further reduction -> stop: 4); in the timed loop the point goes
    # back to x0 (satba_snapshot_x: device-to-device copy of x, inside the timed region) every `restart` iterations and the next
    # solve starts.  Without restarts all but the first steps would sit on the converged point, where the trust radius collapses to
    # 1e-15 and the loop's degenerate-subspace fallbacks run (extra passes, and run-dependent: 632 - 684 it/s over repeated runs).
    restart, solve_stats = max(args.restart_every, 0), None
    if args.restart_every < 0 and driver.startswith("native"):
        eng.snapshot_x(False)
        ls = eng.solve_lm(ftol=1e-4, xtol=1e-10, gtol=1e-8, max_nfev=300, loss=args.loss, f_scale=1.0)
        restart = max(1, int(ls.nfev) - 1)
        solve_stats = {"nfev": int(ls.nfev), "status": int(ls.status), "cost": float(ls.cost), "initial_cost": float(ls.initial_cost)}
        eng.snapshot_x(True)
    elif args.restart_every < 0:
        # several ranks (phase driver): the same count from the loop itself -- iterations until an accepted step reduces the cost
        # by less than ftol x cost (scipy's check_termination), every rank sees the same all-reduced scalars
        eng.snapshot_x(False)
        n_it, prev = 0, None
        while n_it < 300:
            acc = st["accepted"]
            step()
            n_it += 1
            if st["accepted"] > acc and prev is not None and prev - st["cost"] < 1e-4 * prev:
                break
            prev = st["cost"]
        restart = n_it
        solve_stats = {"nfev": n_it + 1, "status": 2 if n_it < 300 else 0, "cost": float(st["cost"]), "initial_cost": None}
        eng.snapshot_x(True)
        st.update(first=True, accepted=0)
        st.pop("interior", None)
    elif restart:
        eng.snapshot_x(False)
    n_step = [0]
    plain_step = step

    def step():  # noqa: F811
        if restart and n_step[0] and n_step[0] % restart == 0:
            eng.snapshot_x(True)
            st["first"] = True
        plain_step()
        n_step[0] += 1

    # One rank: the iterations run on the device-resident loop (satba_lm_run, csrc/satba_lmdev.h: the same phases, the decisions
    # between them -- and the return to x0 every `restart` iterations -- taken by one-thread kernels on the device; the host only
    # queues launches); `--driver native-sync` times the round-2 host side instead (satba_lm_step: two blocking header reads per
    # iteration, a failed factorisation is not repeated).
    ticks = driver == "native" and not os.environ.get("SATBA_HOST_LOOP") and (args.driver == "native" or bool(eng.info()["device_loop"]))
    if driver == "native" and not ticks:
        driver = "native-sync"  # what satba_solve_lm uses at this size (auto), or forced by the environment
    ls = None

    def run(n):
        """n iterations, the point going back to x0 every `restart` of them"""
        if ticks:
            return eng.lm_run(n, cycle_len=restart, lam_floor=1e-14)
        if driver == "device":
            eng.lm_begin(loss=args.loss, f_scale=1.0, never_stop=True, max_iterations=n, cycle_len=restart)
            trf.drive_device_loop(eng, comm, 1e-14, max_patterns=24 * n + 1000)
            return eng.lm_state()
        for _ in range(n):
            step()
        return None

    if args.warmup:
        run(args.warmup)
    sync()
    st["accepted"] = 0
    n_step[0] = 0
    if restart:
        eng.snapshot_x(True)
        st["first"] = True
    sync()
    t0 = time.perf_counter()
    ls = run(args.steps)
    sync()
    dt = time.perf_counter() - t0
    # the roofline kernel inside LM iterations: a second pass over the same iterations with HIP events around every k_linearize
    # launch, on its launch stream (the timed pass itself carries no events)
    if restart:
        eng.snapshot_x(True)
        st["first"] = True
    n_step[0] = 0
    keep = dict(st)
    eng.profile_linearize(True)
    run(min(args.steps, 40))
    n_lin, ms_lin = eng.profile_read()
    eng.profile_linearize(False)
    st.clear()
    st.update(keep)
    if ticks or driver == "device":
        if int(ls["phase"]) == 2:
            raise SystemExit("the device-resident loop stopped for the host (reason {}): use --driver native-sync".format(int(ls["host_reason"])))
        assert int(ls["iterations"]) == args.steps, ls
        st.update(accepted=int(ls["accepted"]), interior=int(ls["newton"]), cost=ls["cost_new"] if ls["actual"] > 0 else ls["cost"])
    phase_ms = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # where an iteration's time goes with several ranks: a second pass over the same iterations, every
        # phase and every collective between torch events on the stream they are queued on, the blocking header reads on the
        # host's clock; rank 0's figures, ms per iteration.  ("dense_solve" includes the scaling of the reduced system and the
        # back-substitution of the points; the all-reduce of S is `allreduce_schur`, pack and unpack included.)
        if restart:
            eng.snapshot_x(True)
            st["first"] = True
        n_step[0] = 0
        keep2 = dict(st)
        prof = PhaseProfile(torch)
        n_prof = min(args.steps, 20)
        for _ in range(n_prof):
            if restart and n_step[0] and n_step[0] % restart == 0:
                eng.snapshot_x(True)
                st["first"] = True
            lm_step(eng, comm, st, trf, prof)
            n_step[0] += 1
        phase_ms = prof.result(n_prof)
        st.clear()
        st.update(keep2)

    # dominant kernels, HIP events on the launch stream (rank-local shard)
    eng.linearize()
    comm.allreduce(eng, eng.len_lin)
    eng.prepare(False)
    eng.schur(1e-6)
    kern = {}
    for name in ("linearize", "residual", "jvp", "backsub", "schur", "cholesky"):
        eng.linearize(); eng.prepare(False); eng.schur(1e-6)
        kern[name] = eng.time_kernel(name, args.kernel_reps if name not in ("schur", "cholesky") else max(2, args.kernel_reps // 5))
    torch.cuda.synchronize()

    if rank == 0:
        K_loc, N_loc = eng.n_obs, eng.n_pts
        alg_bytes = 48.0 * K_loc + 96.0 * N_loc
        # the roofline kernel's duration: average over its launches INSIDE the timed LM iterations (HIP events on the launch
        # stream); kernel_ms["linearize"] is the same kernel in 20 back-to-back launches after the loop (a few % slower: the loop
        # leaves part of the working set in the Infinity Cache)
        t_lin = (ms_lin / n_lin if n_lin else kern["linearize"]) * 1e-3
        achieved = alg_bytes / t_lin / 1e9
        traffic = profiled_traffic(args.shape, args.loss, world)
        info = eng.info()
        comp_bytes = compulsory_bytes(K_loc, N_loc, model, p.n_params, args.loss, bool(info["unit_weights"]), driver.startswith("native"))
        rate_as_built = comp_bytes / t_lin / 1e9
        rate_counter = (traffic / t_lin / 1e9) if traffic else None
        # `bound`: the roofline the kernel is priced against -- HBM (it streams observations; no matrix-core work).  `limiter`: what
        # actually holds it below that roof.  The kernel moves `compulsory_bytes_as_built` (the counters agree to a few %); when that
        # rate is below half the HBM peak the memory system is not the limit: round 6 measured the floor (profiles/r6_linearize_floor.txt:
        # LDS atomics compiled out -3 %, all camera-sum work out -19 %, deeper prefetch nothing) -- vector issue at four waves per SIMD
        # with the LDS pipe beside it.  (Rounds 3-5 wrote "lds" into `bound`.)
        bound = "hbm"
        limiter = "hbm" if (rate_counter if rate_counter is not None else rate_as_built) >= 0.5 * HBM_PEAK_GBPS else "vector issue + LDS pipe at 4 waves/SIMD"
        out = {
            "metric": "LM iters/sec at 200 cams x 1M pts x 10M obs (affine, R+T)" if args.shape == "C4" else
                      "LM iters/sec, config {}".format(args.shape),
            "value": args.steps / dt, "unit": "LM iters/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic", "loss": args.loss,
            # every reduction has a fixed order or is integer arithmetic: repeated runs give the same bits on both routes
            "deterministic": True, "camera_sums": "fixed_point_lds" if info["cam_sums_lds"] else "camera_major_pass",
            "fixed_point_fallbacks": int(info["fx_fallbacks"]),
            "host_driver": driver,
            "host_driver_reason": ("device-resident loop (satba_lm_run): the library's default on one rank" if ticks else
                                   "several ranks: tick parts with the all-reduces queued between them" if driver == "device" else
                                   "host loop: forced by --driver / SATBA_HOST_LOOP / SATBA_DEVICE_LOOP=0 (or more than 1 024 camera unknowns)"),
            # the factorisation beside the pair kernel in the last front: 1 ran that way, 0 not applicable (several ranks, few cameras,
            # switched off), -1 a wait timed out and the handle fell back to one kernel after the other (`value` is then ~10 % lower)
            "chol_beside": int(info["chol_beside"]), "chol_beside_timeouts": int(info["chol_beside_timeouts"]),
            "restart_every": restart, "solve_shipped_tolerances": solve_stats,
            "config": {"workload": "{}: {} cams x {} pts x {} obs, {}, correction {}, 1 fixed camera, seed 1"
                       .format(args.shape, n_cam, n_pts, p.n_obs, model, "+".join(corr)),
                       "sharding": "points over {} rank(s)".format(world),
                       "obs_per_rank0": K_loc},
            "obs_per_sec_residual_jacobian": world * K_loc / t_lin,
            # achieved / frac: SURVEY 8d's algorithmic bytes (48 K + 96 N: a throughput-equivalent, the kernel no longer moves all
            # of them); *_as_built: the bytes this kernel has to move; achieved_counter: the bytes the PMC counters saw it move
            # (bound and the as-built fraction first: `frac` is the throughput equivalent and must not be read without them)
            "roofline": {"bound": bound, "limiter": limiter, "frac_as_built": rate_as_built / HBM_PEAK_GBPS,
                         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "achieved_counter": rate_counter,
                         "compulsory_bytes_as_built": comp_bytes, "achieved_as_built": rate_as_built,
                         "kernel": "k_linearize",
                         "algorithmic_bytes_per_launch": alg_bytes, "ms_per_launch": 1e3 * t_lin, "launches_timed": n_lin,
                         "ms_per_launch_back_to_back": kern["linearize"]},
            "kernel_ms": kern,
            # several ranks: device time per phase and collective, host time in blocking header reads (per iteration).  Taken in a separate
            # pass with the PYTHON host loop (lm_step: torch events around every phase, two blocking header reads per iteration) -- the
            # timed figure above comes from `host_driver`, which for "device" has no host wait at all: host_wait below is that pass's
            "phase_ms": phase_ms, "phase_ms_driver": "python" if phase_ms is not None else None,
            "accepted_steps": st["accepted"], "interior_2d_steps": st.get("interior", 0), "final_cost": st["cost"], "scene_gen_s": t_gen,
            "launch_patterns_executed": int(ls["ticks"]) if ls else None,  # > steps when a factorisation had to be repeated with more damping
        }
        if world == 1 and not args.no_e2e:
            # The drop-in call a user of the reference makes: ba_core.run_ba_optimization(p, ls_params) (ref:bundle_adjust/ba_core.py:244-332;
            # the reference times the solve inside it at :283-299).  Wall clock of the first call -- engine creation: upload of the four
            # observation arrays, layout construction on the device -- and of a warm one (cached engine), split into its parts; the
            # solve is the one under the shipped tolerances from x0 (nfev evaluations), not the fixed-work loop timed above.
            from satba import ba_core

            eng.close()  # (the benchmark's own handle: the call below builds the one a caller would get)
            e2e = {"call": "ba_core.run_ba_optimization(p, {loss, verbose: 0}, False, False)"}
            # "new_engine": the first call on this `p` -- upload of the four observation arrays and the layout build -- in a process whose
            # HIP context and code object are already up (this benchmark has run); a genuinely first call of a fresh process adds ~0.9 s of
            # start-up on top (tools/e2e_time.py).  "warm": the cached engine.  (Rounds 4-5 called the former "first".)
            for which in ("new_engine", "warm"):
                tm = {}
                t0 = time.perf_counter()
                ret = ba_core.run_ba_optimization(p, {"loss": args.loss, "verbose": 0, "timings": tm}, False, False)
                tm["wall_s"] = time.perf_counter() - t0  # (the caller holds the five results; giving 210 MB of them back to the system is another 8 ms at this size)
                del ret
                tm["host_overhead_frac"] = 1.0 - tm["solve_s"] / tm["wall_s"]
                e2e[which] = tm
            out["e2e"] = e2e
        if args.cpu_sample_pts > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(scene, min(args.cpu_sample_pts, n_pts), corr, full_c3=args.cpu_c3)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
