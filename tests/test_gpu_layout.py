"""
Round-2 components, through the C ABI on the GPU:
  * the index structures satba_problem_create builds ON THE DEVICE (csrc/satba_layout.h: track-length-sorted sliced ELL,
    camera-major lists, per-camera-pair lists) against a numpy restatement -- integer work: bit-exact;
  * satba_solve_lm (the trust-region loop in C++) against the Python loop of satba/trf.py: same nfev, status, x;
  * satba_outliers against vectors captured from ref:bundle_adjust/ba_outliers.py (tests/golden/outliers.npz): thresholds
    equal, removed set index-exact;
  * bitwise-repeatable runs on both routes to the per-camera sums (fixed point / camera-major), and the fall-back between them;
  * a pipeline-shaped driver through the drop-in names only (ref:bundle_adjust/ba_pipeline.py:700-728).
"""
import numpy as np
import pytest

import cases
from satba import ba_core, ba_outliers, synth, trf
from satba.engine_hip import HipEngine

pytestmark = pytest.mark.gpu


def numpy_layout(p, C):
    """The structures of csrc/satba_layout.h restated with numpy (the 'host builder' the device is checked against)."""
    N, M, K = p.n_pts, p.n_cam, p.n_obs
    pts, cam = p.pts_ind.astype(np.int64), p.cam_ind.astype(np.int64)
    cnt = np.bincount(pts, minlength=N)
    perm = np.argsort(cnt, kind="stable")
    rank = np.empty(N, dtype=np.int64)
    rank[perm] = np.arange(N)
    pt_cnt = cnt[perm]
    n_slices = (N + 63) // 64
    last = np.minimum(64 * np.arange(n_slices) + 63, N - 1)
    slice_base = np.concatenate(([0], np.cumsum(64 * pt_cnt[last])))
    ofs = np.searchsorted(pts, np.arange(N + 1))
    k = np.arange(K) - ofs[pts]
    q = rank[pts]
    pos = slice_base[q >> 6] + 64 * k + (q & 63)
    e_cam = np.full(slice_base[-1], -1, dtype=np.int64)
    e_cam[pos] = cam
    order = np.lexsort((q, cam))  # camera-major, internal point ascending
    ipt_ofs = np.concatenate(([0], np.cumsum(pt_cnt)))
    io_idx = ipt_ofs[q] + k  # internal point-major index of every observation
    out = dict(perm=perm, rank=rank, pt_cnt=pt_cnt, slice_base=slice_base, e_cam=e_cam, obs_pos=pos, cm_pt=q[order], cm_pos=pos[order],
               cm_io=io_idx[order], ipt_ofs=ipt_ofs, cam_ofs=np.searchsorted(cam[order], np.arange(M + 1)))
    # pair lists: all camera pairs of every point in internal point order, stable sort by pair index
    io = np.lexsort((k, q))  # internal point-major
    keys, hq, hpi, hpj = [], [], [], []
    b = 0
    for n in pt_cnt[pt_cnt > 0]:
        cs, ps, qq = cam[io[b: b + n]], io_idx[io[b: b + n]], q[io[b]]
        for x in range(n):
            for y in range(x + 1, n):
                keys.append(cs[x] * M - cs[x] * (cs[x] + 1) // 2 + (cs[y] - cs[x] - 1))
                hq.append(qq); hpi.append(ps[x]); hpj.append(ps[y])
        b += n
    keys, hq, hpi, hpj = (np.array(a, dtype=np.int64) for a in (keys, hq, hpi, hpj))
    s = np.argsort(keys, kind="stable")
    n_pairs = M * (M - 1) // 2
    comp = keys[s] * (C + 1) + hq[s] * C // N
    out.update(pair_pts=hq[s], pair_pi=hpi[s], pair_pj=hpj[s], pair_ofs=np.searchsorted(comp, np.arange(n_pairs * (C + 1) + 1)),
               pair_ij=np.array([(i, j) for i in range(M) for j in range(i + 1, M)]).ravel())
    return out


@pytest.mark.parametrize("model,M,N,opp,seed", [("affine", 9, 3000, 4, 3), ("affine", 70, 200, 68, 5), ("rpc", 3, 100, 2, 7),
                                                ("perspective", 5, 64, 3, 1), ("affine", 2, 1, 2, 2)])
def test_device_layout_is_bit_exact(gpu, model, M, N, opp, seed):
    scene = synth.make_scene(model, M, N, opp, seed=seed)
    p = synth.make_params(scene, {"correction_params": ["R"], "n_cam_fix": 1})
    eng = HipEngine(p)
    info = eng.info()
    want = numpy_layout(p, int(info["pair_chunks"]))
    assert info["ell_len"] == want["slice_base"][-1] and info["pair_entries"] == want["pair_pts"].size
    for name, ref in want.items():
        got = eng.get_layout(name)
        assert got.shape == ref.shape and np.array_equal(got, ref), name
    # a point with more than 64 observations is nothing special in this layout (round 1 had a split-tile path)
    assert model != "affine" or M < 70 or np.bincount(p.pts_ind).max() > 64
    eng.close()


def numpy_merged_records(lay, M, N, C, spc):
    """The merged records of the weighted / robust runs (csrc/satba_layout.h: Layout::w_fix ...) restated with numpy from the base layout."""
    cnt, ipt_ofs = lay["pt_cnt"], lay["ipt_ofs"]
    size = np.concatenate((((cnt + 6 + 7) // 8) * 8, [8]))               # pieces per record (a multiple of 8: it ends on a line); + the zero record
    begin = np.concatenate(([0], np.cumsum(size)))[:-1]
    w_fix = begin + size - 6                                             # piece of X0: six pieces in front of the end
    sc_ofs = w_fix - np.concatenate((cnt, [0]))                          # piece of the first row scale
    q = lay["pair_pts"]
    di = cnt[q] - (lay["pair_pi"] - ipt_ofs[q])
    dj = cnt[q] - (lay["pair_pj"] - ipt_ofs[q])
    cq = lay["cm_pt"]
    out = dict(w_fix=w_fix, sc_ofs=sc_ofs, pair_rec=w_fix[q], pair_kk=di | (dj << 16), cm_rec=w_fix[cq],
               cm_sc=w_fix[cq] - cnt[cq] + (lay["cm_io"] - ipt_ofs[cq]))
    dg = []
    for cam in range(M):
        b0, e0 = lay["cam_ofs"][cam], lay["cam_ofs"][cam + 1]
        chunk = cq[b0:e0] * C // max(N, 1)
        for ch in range(C):
            b, e = b0 + np.searchsorted(chunk, ch), b0 + np.searchsorted(chunk, ch + 1)
            dg.extend(b + (e - b) * s // spc for s in range(spc))
    out["dg_ofs"] = np.array(dg + [lay["cam_ofs"][M]], dtype=np.int64)
    return out


@pytest.mark.parametrize("model,M,N,opp,seed", [("affine", 9, 3000, 4, 3), ("affine", 70, 200, 68, 5), ("perspective", 5, 64, 3, 1),
                                                ("affine", 2, 1, 2, 2)])
def test_merged_record_layout_is_bit_exact(gpu, model, M, N, opp, seed):
    """The index structures of the weighted / robust runs (round 5), built on the device by the first such linearisation: every array
    against the numpy restatement; before that linearisation they do not exist."""
    scene = synth.make_scene(model, M, N, opp, seed=seed)
    p = synth.make_params(scene, {"correction_params": ["R"], "n_cam_fix": 1})
    eng = HipEngine(p)
    with pytest.raises(ValueError):
        eng.get_layout("w_fix")
    eng.configure("linear", 1.0)
    eng.linearize()                      # unit weights, linear loss: the merged records are not built
    with pytest.raises(ValueError):
        eng.get_layout("w_fix")
    eng.configure("soft_l1", 1.0)
    eng.linearize()
    info = eng.info()
    C, spc = int(info["pair_chunks"]), int(info["diag_items_per_chunk"])
    assert spc >= 1
    lay = {k: eng.get_layout(k).astype(np.int64) for k in ("pt_cnt", "ipt_ofs", "pair_pts", "pair_pi", "pair_pj", "cm_pt", "cm_io", "cam_ofs")}
    want = numpy_merged_records(lay, M, p.n_pts, C, spc)
    for name, ref in want.items():
        got = eng.get_layout(name)
        assert got.shape == ref.shape and np.array_equal(got, ref), name
    assert np.all(want["w_fix"] % 8 == 2)  # every record ends on a 128-byte line
    eng.close()


def test_layout_rejects_unsorted_cameras(gpu):
    scene = synth.make_affine_scene(4, 30, 3, seed=1)
    p = synth.make_params(scene, {"correction_params": ["R"]})
    p.cam_ind = p.cam_ind.copy()
    first = np.nonzero(np.diff(p.pts_ind) == 0)[0][0]
    p.cam_ind[[first, first + 1]] = p.cam_ind[[first + 1, first]]  # cameras of one point descending
    with pytest.raises(ValueError):
        HipEngine(p)
    p.cam_ind[first] = 99
    with pytest.raises(ValueError):
        HipEngine(p)


@pytest.mark.parametrize("name,loss", [("affine_small_R", "linear"), ("affine_small_R", "soft_l1"), ("persp_small_RT", "linear"),
                                       ("affine_C2_R", "linear")])
@pytest.mark.parametrize("tight", [True, False])
def test_solve_lm_matches_python_loop(gpu, name, loss, tight):
    """satba_solve_lm (C++) against satba/trf.py (Python): the same scalars drive the same decisions."""
    _, make_p, g, _ = cases.solve_case(name)
    tol = dict(ftol=1e-15, xtol=1e-15, gtol=1e-15) if tight else dict(ftol=1e-4, xtol=1e-10, gtol=1e-8)
    outs = []
    for native in (False, True):
        p = make_p()
        eng = HipEngine(p)  # the default path is bitwise repeatable: the two loops see identical scalars
        res = trf.trf_solve(eng, max_nfev=300, loss=loss, native=native, **tol)
        outs.append((res, eng.get_x()))
        eng.close()
    (ra, xa), (rb, xb) = outs
    assert ra.nfev == rb.nfev and ra.status == rb.status and ra.njev == rb.njev and ra.iterations == rb.iterations
    assert abs(ra.cost - rb.cost) <= 1e-14 * ra.cost
    assert np.abs(xa - xb).max() <= 1e-12 * np.abs(xa).max()


def test_lm_step_matches_python_iteration(gpu):
    """satba_lm_step (the iteration bench.py times) against the same iteration driven from Python phase by phase."""
    import bench

    _, make_p, _, _ = cases.solve_case("affine_small_R")
    runs = []
    for native in (False, True):
        eng = HipEngine(make_p())  # the default path is bitwise repeatable: both drivers see identical scalars
        eng.configure("linear", 1.0)
        st = {"first": True, "accepted": 0, "fail": 0, "cost": None}
        trace = []
        for _ in range(6):
            if native:
                bench.lm_step_native(eng, st)
            else:
                bench.lm_step(eng, trf.SingleComm(), st, trf)
            trace.append((st["cost"], st["Delta"], st["accepted"], st.get("interior", 0)))
        runs.append((trace, eng.get_x()))
        eng.close()
    (ta, xa), (tb, xb) = runs
    for a, b in zip(ta, tb):
        assert a[2:] == b[2:]
        assert abs(a[0] - b[0]) <= 1e-13 * abs(a[0]) and abs(a[1] - b[1]) <= 1e-12 * abs(a[1])
    assert np.abs(xa - xb).max() <= 1e-12 * np.abs(xa).max()


def test_device_resident_iterations_match_host_driven_iterations(gpu):
    """
    satba_lm_run (decisions taken by one-thread kernels on the device, gated launches, accepted points copied on the device;
    csrc/satba_lmdev.h) against satba_lm_step (the same iteration with two header reads and the decisions on the host): same
    accept / reject sequence, same radius, same point -- to the last bit, since both run the same kernels on the same inputs.
    Then with cycles: back to the kept point every 3 iterations, on the device, against the same done from the host.
    """
    import bench

    _, make_p, _, _ = cases.solve_case("affine_small_R")
    eng = HipEngine(make_p())
    eng.configure("linear", 1.0)
    st = {"first": True, "accepted": 0, "fail": 0, "cost": None}
    for _ in range(7):
        bench.lm_step_native(eng, st)
    x_host = eng.get_x()
    eng.close()
    eng = HipEngine(make_p())
    eng.configure("linear", 1.0)
    ls = eng.lm_run(7, lam_floor=1e-14)  # bench.lm_step_native's damping floor
    assert int(ls["phase"]) == 1 and int(ls["iterations"]) == 7 and int(ls["ticks"]) >= 7 and int(ls["accepted"]) == st["accepted"]
    assert ls["Delta"] == st["Delta"], (ls["Delta"], st["Delta"])
    assert (ls["cost_new"] if ls["actual"] > 0 else ls["cost"]) == st["cost"], (ls, st)
    assert np.array_equal(eng.get_x(), x_host)
    eng.close()
    # cycles
    xs = []
    for device in (False, True):
        eng = HipEngine(make_p())
        eng.configure("linear", 1.0)
        eng.snapshot_x(False)
        if device:
            eng.lm_run(8, cycle_len=3, lam_floor=1e-14)
        else:
            st = {"first": True, "accepted": 0, "fail": 0, "cost": None}
            for i in range(8):
                if i and i % 3 == 0:
                    eng.snapshot_x(True)
                    st["first"] = True
                bench.lm_step_native(eng, st)
        xs.append(eng.get_x())
        eng.close()
    assert np.array_equal(xs[0], xs[1])


@pytest.mark.parametrize("name,loss,tight", [("affine_small_R", "soft_l1", True), ("persp_small_RT", "linear", True), ("affine_C2_R", "linear", False)])
def test_device_resident_solve_matches_host_loop(gpu, monkeypatch, name, loss, tight):
    """satba_solve_lm on the device-resident loop against the same solve with the decisions on the host (SATBA_HOST_LOOP): identical."""
    _, make_p, g, _ = cases.solve_case(name)
    tol = dict(ftol=1e-15, xtol=1e-15, gtol=1e-15) if tight else dict(ftol=1e-4, xtol=1e-10, gtol=1e-8)
    outs = []
    for host in (False, True):
        if host:
            monkeypatch.setenv("SATBA_HOST_LOOP", "1")
        eng = HipEngine(make_p())
        st = eng.solve_lm(max_nfev=300, loss=loss, **tol)
        outs.append(((st.cost, st.nfev, st.njev, st.iterations, st.status, st.optimality, st.initial_cost), eng.get_x()))
        eng.close()
    assert outs[0][0] == outs[1][0] and np.array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("model,M,N,opp,corr", [("affine", 40, 4000, 6, ["R", "T"]), ("affine", 70, 5000, 5, ["R"]),
                                                ("affine", 200, 20000, 8, ["R", "T"]), ("perspective", 30, 3000, 5, ["R", "T"]),
                                                ("affine", 27, 2000, 6, ["R", "T"])])
@pytest.mark.parametrize("loop,loss", [("host", "linear"), ("device", "linear"), ("host", "soft_l1")])
def test_factorisation_beside_the_pair_kernel_is_the_sequential_solve(gpu, monkeypatch, model, M, N, opp, corr, loop, loss):
    """
    One rank, unit weights, more than two tile columns: the tile Cholesky runs on its own stream beside k_schur_pairs and takes every
    tile when the producers of its columns have counted themselves in (front_schur_solve, C3Args::arrive).  Same arithmetic in the
    same order as scale -> factorise -> substitute one after the other (SATBA_CHOL_BESIDE=0): identical to the last bit, for the
    loop with the decisions on the host and for the device-resident one.  soft_l1: the pair kernel works on point-range chunks, the
    last chunk's item of a pair adds the partials (SchurArgs::pair_cnt) in the order of the reduce pass it replaces.
    """
    scene = synth.make_scene(model, M, N, opp, seed=11)
    if loop == "host":
        monkeypatch.setenv("SATBA_HOST_LOOP", "1")
    monkeypatch.setenv("SATBA_SCHUR_MERGE", "1")  # one item per camera pair for the few cameras of a test, too (the default from 8 192 pairs on)
    outs = []
    for beside in ("1", "0", "1"):
        monkeypatch.setenv("SATBA_CHOL_BESIDE", beside)
        eng = HipEngine(synth.make_params(scene, {"correction_params": corr, "n_cam_fix": 1}))
        st = eng.solve_lm(max_nfev=30, loss=loss, ftol=1e-12, xtol=1e-12, gtol=1e-12)
        outs.append(((st.cost, st.nfev, st.njev, st.iterations, st.status, st.optimality, st.initial_cost), eng.get_x()))
        assert int(eng.info()["chol_beside"]) == int(beside)
        eng.close()
    assert outs[0][0] == outs[1][0] == outs[2][0]
    assert np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][1], outs[2][1])
    assert outs[0][0][0] < 0.5 * outs[0][0][6]  # (the solve went somewhere)


@pytest.mark.parametrize("loop", ["device", "host"])
def test_factorisation_beside_the_pair_kernel_falls_back_when_launches_are_serialised(gpu, loop):
    """
    A tool that runs one kernel at a time (counter collection; here AMD_SERIALIZE_KERNEL=3) leaves the factorisation waiting for a
    pair kernel that has not been started: its wait times out (status bit 1), the handle goes back to one kernel after the other and
    the front is repeated with the same damping -- the solve is the sequential one, bit for bit, in BOTH loops (round 5: the
    device-resident loop used to book the time-out as a failed factorisation and escalate the damping; it now hands the front back,
    LM_HOST_BESIDE, and carries on).  The stall is a few milliseconds (round 4: 0.3 - 1 s): asserted below 50 ms per time-out.
    """
    import os
    import subprocess
    import sys

    code = (
        "import sys, hashlib, time; sys.path.insert(0, %r)\n"
        "from satba import synth\nfrom satba.engine_hip import HipEngine\n"
        "scene = synth.make_scene('affine', 40, 3000, 6, seed=11)\n"
        "eng = HipEngine(synth.make_params(scene, {'correction_params': ['R', 'T'], 'n_cam_fix': 1}))\n"
        "eng.solve_lm(max_nfev=2, loss='linear')\n"  # (first call: module load, stream creation -- not part of the stall)
        "eng.set_x(eng.p.params_opt.copy())\n"
        "t0 = time.perf_counter()\n"
        "st = eng.solve_lm(max_nfev=6, loss='linear', ftol=1e-12, xtol=1e-12, gtol=1e-12)\n"
        "dt = time.perf_counter() - t0\n"
        "print('RESULT', repr(st.cost), st.nfev, st.status, hashlib.sha1(eng.get_x().tobytes()).hexdigest())\n"
        "i = eng.info(); print('BESIDE', int(i['chol_beside']), 'TIMEOUTS', int(i['chol_beside_timeouts']), 'SECONDS', dt)\n"
    ) % os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sat-bundleadjust_amd")
    outs, secs, timeouts = [], [], []
    for env_add in ({"SATBA_CHOL_BESIDE": "0", "AMD_SERIALIZE_KERNEL": "3"}, {"AMD_SERIALIZE_KERNEL": "3"}):
        env = dict(os.environ, SATBA_SCHUR_MERGE="1", **env_add)
        if loop == "host":
            env["SATBA_HOST_LOOP"] = "1"
        else:
            env.pop("SATBA_HOST_LOOP", None)
            env["SATBA_DEVICE_LOOP"] = "1"
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=240)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][-1])
        info = [ln for ln in r.stdout.splitlines() if ln.startswith("BESIDE")][-1].split()
        assert int(info[1]) == (0 if "SATBA_CHOL_BESIDE" in env_add else -1), r.stdout
        timeouts.append(int(info[3]))
        secs.append(float(info[5]))
    assert outs[0] == outs[1], outs
    assert timeouts[0] == 0 and timeouts[1] >= 1, timeouts
    # both runs are serialised; the second one also lost `timeouts` fronts to the wait
    assert secs[1] - secs[0] < 0.050 * timeouts[1], (secs, timeouts)


def test_snapshot_restores_the_point_and_the_solve_repeats(gpu):
    """satba_snapshot_x (what bench.py restarts its solve with): the point comes back bit for bit, and the solve that follows
    repeats the first one exactly (default path: fixed-point camera sums)."""
    import bench

    _, make_p, _, _ = cases.solve_case("affine_small_R")
    p = make_p()
    eng = HipEngine(p)
    eng.configure("linear", 1.0)
    with pytest.raises(Exception):
        eng.snapshot_x(True)  # nothing kept yet
    x0 = eng.get_x()
    eng.snapshot_x(False)
    traces = []
    for _ in range(2):
        st = {"first": True, "accepted": 0, "fail": 0, "cost": None}
        trace = []
        for _ in range(4):
            bench.lm_step_native(eng, st)
            trace.append((st["cost"], st["Delta"], st["accepted"]))
        traces.append((trace, eng.get_x()))
        assert np.abs(traces[-1][1] - x0).max() > 0
        eng.snapshot_x(True)
        assert np.array_equal(eng.get_x(), x0)
    assert traces[0][0] == traces[1][0] and np.array_equal(traces[0][1], traces[1][1])
    eng.close()


def test_solve_lm_rejects_multi_rank_handles_and_bad_loss(gpu):
    from satba import sharding

    _, make_p, _, _ = cases.solve_case("affine_small_R")
    p = make_p()
    eng = HipEngine(p, sharding.make_shard(p, 0, 2))
    with pytest.raises(ValueError):
        eng.solve_lm()
    with pytest.raises(ValueError):
        eng.solve_lm(loss="nope")
    eng.close()


@pytest.mark.parametrize("name,M,N,opp", [("a", 6, 600, 4), ("b", 12, 1500, 5), ("c", 3, 40, 2)])
def test_outliers_match_reference(gpu, name, M, N, opp):
    """Thresholds and removed set of ref:bundle_adjust/ba_outliers.py:112-155 -- index-exact (golden from the reference)."""
    g = cases.golden("outliers")
    scene = synth.make_affine_scene(M, N, opp, seed=13, sigma_theta=2e-6)
    p = synth.make_params(scene, {"correction_params": ["R"], "n_cam_fix": 0, "reduce": False}, dense=True)
    p.pts2d = g[name + "_pts2d"].copy()
    # (1) errors handed in: the reference's own vector
    remove, cam_thr, n = ba_outliers.compute_obs_mask(g[name + "_err"], p)
    assert np.array_equal(np.array(cam_thr), g[name + "_cam_thr"])
    assert n == int(g[name + "_n"]) and np.array_equal(remove, g[name + "_removed"])
    C_new, thr2, n2 = ba_outliers.compute_obs_to_remove(g[name + "_err"], p)
    assert n2 == n and np.array_equal(np.isnan(C_new[2 * p.cam_ind, p.pts_ind]), g[name + "_removed"])
    # (2) errors computed on the device from the residuals at the current x
    eng = ba_core.get_engine(p)
    eng.configure("linear", 1.0)
    eng.set_x(ba_core._frozen_vars(p.params_opt.copy(), p))
    thr_d, rm_d, n_d = eng.outliers()
    assert np.array_equal(thr_d, g[name + "_cam_thr"]) and n_d == n and np.array_equal(rm_d, g[name + "_removed"])
    # (3) predefined threshold
    rm3, thr3, _ = ba_outliers.compute_obs_mask(g[name + "_err"], p, predef_thr=3.14159)
    assert np.array_equal(np.array(thr3), g[name + "_thr_predef"]) and np.array_equal(rm3, g[name + "_removed_predef"])


@pytest.mark.parametrize("route", ["default", "camera_major"])
@pytest.mark.parametrize("loss", ["linear", "soft_l1"])
def test_runs_repeat_bitwise(gpu, loss, route):
    """
    Every reduction of the library has a fixed order, and the per-camera sums of the linearisation are either 64-bit fixed point
    (default: integer addition does not depend on the order the lanes arrive in) or a fixed-order camera-major pass
    (SATBA_FLAG_CAMERA_MAJOR_SUMS): two handles, two runs, identical bits -- on both routes.
    """
    scene = synth.make_affine_scene(12, 4000, 5, seed=4, sigma_theta=2e-5)
    xs, costs = [], []
    for _ in range(3):
        p = synth.make_params(scene, {"correction_params": ["R", "T"], "n_cam_fix": 1})
        eng = HipEngine(p, deterministic=(route == "camera_major"))
        info = eng.info()
        assert info["cam_sums_lds"] == (1 if route == "default" else 0)
        res = trf.trf_solve(eng, ftol=1e-10, xtol=1e-12, gtol=1e-12, max_nfev=40, loss=loss)
        assert eng.info()["fx_fallbacks"] == 0
        xs.append(eng.get_x())
        costs.append((res.cost, res.nfev, res.optimality))
        eng.close()
    assert costs[0] == costs[1] == costs[2] and np.array_equal(xs[0], xs[1]) and np.array_equal(xs[0], xs[2])
    # and the other route walks the same path (40 evaluations do not converge the robust run: the two trajectories are compared,
    # not two minima -- they differ by the rounding of the per-camera sums only)
    p = synth.make_params(scene, {"correction_params": ["R", "T"], "n_cam_fix": 1})
    eng = HipEngine(p, deterministic=(route == "default"))
    res = trf.trf_solve(eng, ftol=1e-10, xtol=1e-12, gtol=1e-12, max_nfev=40, loss=loss)
    assert abs(res.cost - costs[0][0]) < 1e-5 * res.cost  # (the ftol test may trip one evaluation apart: nfev is not compared)
    eng.close()


@pytest.mark.parametrize("native", [True, False])
def test_fixed_point_overflow_falls_back_to_camera_major_sums(gpu, monkeypatch, native):
    """
    A term beyond its bound raises header slot K_FX_BAD; the loops (satba_solve_lm in C++, trf.trf_solve in Python) then switch
    the handle to the camera-major sums and repeat the iteration.  SATBA_FX_SHRINK makes the bounds 1e9 times too small.
    """
    scene = synth.make_affine_scene(8, 2000, 5, seed=5, sigma_theta=2e-5)
    kw = dict(ftol=1e-12, xtol=1e-12, gtol=1e-12, max_nfev=40, loss="linear")
    opts = {"correction_params": ["R"], "n_cam_fix": 1}  # rotations only: a well-determined minimum (R+T on affine cameras is a flat valley)
    p = synth.make_params(scene, opts)
    ref = HipEngine(p)
    res_ref = trf.trf_solve(ref, native=native, **kw)
    x_ref = ref.get_x()
    assert ref.info()["fx_fallbacks"] == 0
    ref.close()
    monkeypatch.setenv("SATBA_FX_SHRINK", "1e-9")
    eng = HipEngine(synth.make_params(scene, opts))
    assert eng.info()["cam_sums_lds"] == 1
    res = trf.trf_solve(eng, native=native, **kw)
    info = eng.info()
    assert info["fx_fallbacks"] == 1 and info["cam_sums_lds"] == 0
    assert res.status == res_ref.status and abs(res.cost - res_ref.cost) < 1e-10 * res_ref.cost
    assert np.abs(eng.get_x() - x_ref).max() < 1e-8 * np.abs(x_ref).max()
    eng.close()


def test_pipeline_shaped_driver_through_dropin_names(gpu, tmp_path):
    """
    ref:bundle_adjust/ba_pipeline.py:700-728 with the modules swapped for this package's: define_ba_parameters ->
    run_ba_softL1 -> clean_outlier_observations -> run_ba_L2 -> reconstruct_vars -> save_debug_figures, default options
    (save_figures=True, clean_outliers=True).
    """
    import matplotlib

    matplotlib.use("Agg")
    from satba import ba_params, geo_utils

    scene = synth.make_affine_scene(8, 2500, 5, seed=6, sigma_theta=2e-6, pts_float32=True)
    rng = np.random.default_rng(0)
    bad = rng.random(scene.n_obs) < 0.02
    scene.pts2d[bad] += rng.normal(0, 30.0, (int(bad.sum()), 2))
    C = scene.to_dense_C()
    scene.pairs_to_triangulate = [(i, j) for i in range(8) for j in range(i + 1, 8)]  # every pair has enough baseline
    d = {"n_cam_fix": 0, "n_pts_fix": 0, "ref_cam_weight": 1.0, "correction_params": ["R"], "verbose": False}
    # ba_pipeline.py:292-307: the initial points are triangulated from the tracks
    from satba import ft_triangulate

    pts3d_init = ft_triangulate.init_pts3d(C, scene.cameras, "affine", scene.pairs_to_triangulate)
    assert pts3d_init.dtype == np.float32 and np.median(np.linalg.norm(pts3d_init - scene.pts3d_true, axis=1)) < 100.0
    p = ba_params.BundleAdjustmentParameters(C, pts3d_init, scene.cameras, "affine", scene.pairs_to_triangulate, scene.camera_centers, d)
    ba_iters = 0
    _, ba_sol, init_e, ba_e, iters = ba_core.run_ba_optimization(p, {"loss": "soft_l1", "f_scale": 1.0, "max_iter": 300, "verbose": 0}, False, False)
    ba_iters += iters
    n_before = p.n_obs
    p.reconstruct_vars(ba_sol, pts3d_init, scene.cameras)
    p = ba_outliers.rm_outliers(ba_e, p, verbose=False)  # re-triangulates the surviving tracks (ba_outliers.py:89-93)
    removed = n_before - p.n_obs
    assert 0.5 * bad.sum() < removed < 3 * bad.sum() + 50  # the injected gross errors (and little else) are gone
    _, ba_sol, init_e2, ba_e, iters = ba_core.run_ba_optimization(p, None, False, False)
    ba_iters += iters
    corrected_pts3d, corrected_cameras = p.reconstruct_vars(ba_sol, pts3d_init, scene.cameras)
    assert ba_e.mean() < 0.5 and len(corrected_cameras) == 8 and len(p.estimated_params) == 8
    ba_core.save_histogram_of_errors(str(tmp_path / "ba_figures" / "error_histograms.png"), init_e, ba_e)
    lat, lon, _ = geo_utils.ecef_to_latlon_custom(*np.asarray(scene.pts3d, dtype=np.float64).T)
    box = [[lon.min(), lat.min()], [lon.max(), lat.min()], [lon.max(), lat.max()], [lon.min(), lat.max()], [lon.min(), lat.min()]]
    fp = [{"type": "Polygon", "coordinates": [box]}]
    for tag, e in (("before", init_e2), ("after", ba_e)):
        path = tmp_path / "ba_figures" / "error_{}.png".format(tag)
        ba_core.save_heatmap_of_reprojection_error(str(path), p, e, fp, None, smooth=2, global_transform=None)
        assert path.stat().st_size > 1000
    assert (tmp_path / "ba_figures" / "error_histograms.png").stat().st_size > 1000 and ba_iters > 2


@pytest.mark.parametrize("driver", ["native", "python"])
def test_bench_line_contract(gpu, driver):
    """bench.py on the small shape, as a child process: one JSON line with the keys the driver's contract names, both host drivers."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--shape", "C2", "--steps", "8", "--warmup", "2", "--cpu-sample-pts", "0",
                          "--driver", driver], capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "restart_every"):  # cpu_baseline: the default run's extra leg, skipped here (--cpu-sample-pts 0)
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 8 and d["warmup"] == 2 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["unit"] == "LM iters/sec" and d["higher_is_better"] is True and "workload" in d["config"]
    assert abs(d["value"] - 1000.0 / d["ms_per_step"]) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and isinstance(r["limiter"], str) and r["peak"] == 8000.0 and 0.0 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # the bytes the kernel has to move as built, beside SURVEY 8d's algorithmic bytes: 20 K + 96 N for unit weights and the linear loss
    fused = 96.0 * 5000 if driver == "native" else 0.0  # one-rank loops: the point part of the prepare phase rides in k_linearize
    assert r["compulsory_bytes_as_built"] == 20.0 * d["config"]["obs_per_rank0"] + 96.0 * 5000 + fused and 0.0 < r["frac_as_built"] <= r["frac"]
    assert d["deterministic"] is True and d["camera_sums"] == "fixed_point_lds" and d["fixed_point_fallbacks"] == 0
    assert r["launches_timed"] == 8 and d["restart_every"] >= 1 and d["host_driver"] == driver
    assert d["accepted_steps"] >= 4  # restarts keep the timed steps productive


@pytest.mark.parametrize("driver", ["device", "python"])
def test_bench_line_two_ranks(gpu, driver):
    """bench.py --gpus 2 the way the scaling runs launch it (torch.distributed.run, one process per rank), on this one GPU over gloo
    (--backend gloo --share-gpu: test switches): the line of rank 0, whole-job value, the per-phase breakdown, both host drivers."""
    import json
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--shape", "C2", "--steps", "8", "--warmup", "2", "--cpu-sample-pts", "0",
           "--backend", "gloo", "--share-gpu", "--driver", driver]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 8 and d["host_driver"] == driver and d["scaling"] == "strong"
    assert abs(d["value"] - 1000.0 / d["ms_per_step"]) < 1e-6 * d["value"]
    ph = d["phase_ms"]
    for k in ("linearize", "prepare", "schur", "dense_solve", "trial", "allreduce", "allreduce_schur", "host_wait"):
        assert k in ph and ph[k] >= 0.0, k
    assert d["accepted_steps"] >= 4 and d["config"]["obs_per_rank0"] < 29981  # a shard, not the whole scene
