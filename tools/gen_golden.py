"""
Golden-vector generator -- runs ONLY in the build container, where /root/reference is mounted.

Imports the reference's own hot-path modules in place (never copied; recipe of SURVEY.md appendix A: stub
the third-party modules that are imported at module level but unused on the path) and records their outputs
on seeded synthetic scenes into tests/golden/*.npz.  Inputs are rebuilt at test time from the same seeds by
`satba.synth`; the small ones are also stored so a generator change cannot silently move the vectors.

    MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py
"""
import importlib
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sat-bundleadjust_amd"))
OUT = os.path.join(ROOT, "tests", "golden")


def import_reference():
    for name in ["rpcm", "rasterio", "rasterio.crs", "rasterio.errors", "pyproj", "utm", "srtm4"]:
        sys.modules[name] = types.ModuleType(name)
    sys.modules["rasterio"].errors = sys.modules["rasterio.errors"]
    sys.modules["rasterio.errors"].NotGeoreferencedWarning = type("NotGeoreferencedWarning", (Warning,), {})
    pkg = types.ModuleType("bundle_adjust")
    pkg.__path__ = ["/root/reference/bundle_adjust"]
    sys.modules["bundle_adjust"] = pkg
    mods = {}
    for m in ["ba_core", "ba_params", "ba_rotate", "cam_utils", "geo_utils", "ba_outliers"]:
        mods[m] = importlib.import_module("bundle_adjust." + m)
    return types.SimpleNamespace(**mods)


ref = import_reference()
from satba import synth  # noqa: E402  (own generator; the reference has none)


def ref_params(scene, d):
    d = dict({"verbose": False}, **d)
    return ref.ba_params.BundleAdjustmentParameters(
        scene.to_dense_C(), scene.pts3d, scene.cameras, scene.cam_model, scene.pairs_to_triangulate,
        scene.camera_centers, d)


def fd3_jacobian_blocks(f, v, p, n_p):
    """3-point finite differences of a residual function, gathered into per-observation blocks."""
    K = p.pts_ind.size
    n_c = p.n_cam * n_p
    Jc = np.zeros((K, 2, n_p))
    Jp = np.zeros((K, 2, 3))
    h = np.cbrt(np.finfo(float).eps) * np.maximum(1.0, np.abs(v))
    # columns of different cameras (points) never share an observation: perturb one slot of all of them at once
    for s in range(n_p):
        dv = np.zeros_like(v)
        idx = np.arange(p.n_cam) * n_p + s
        dv[idx] = h[idx]
        d = (f(v + dv) - f(v - dv)).reshape(K, 2) / (2 * h[idx][p.cam_ind])[:, None]
        Jc[:, :, s] = d
    for s in range(3):
        dv = np.zeros_like(v)
        idx = n_c + np.arange(p.n_pts) * 3 + s
        dv[idx] = h[idx]
        d = (f(v + dv) - f(v - dv)).reshape(K, 2) / (2 * h[idx][p.pts_ind])[:, None]
        Jp[:, :, s] = d
    return Jc, Jp


def tight_solve(p, loss="linear", f_scale=1.0, max_nfev=100):
    from scipy.optimize import least_squares

    A = ref.ba_core.build_jacobian_sparsity(p)
    return least_squares(ref.ba_core.fun, p.params_opt.copy(), jac_sparsity=A, x_scale="jac", method="trf",
                         loss=loss, f_scale=f_scale, ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=max_nfev,
                         tr_options={"atol": 1e-12, "btol": 1e-12}, args=(p,))


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, {k: np.asarray(v).shape for k, v in arrays.items()})


def golden_fun_and_jac():
    """G1 (fun at 3 random v), G2 (3-point FD Jacobian blocks of the reference fun), G3 (sparsity), per model."""
    cases = {
        "affine_RT": ("affine", 4, 50, 3, {"correction_params": ["R", "T"], "n_cam_fix": 0}),
        "affine_R_fix": ("affine", 6, 120, 4, {"correction_params": ["R"], "n_cam_fix": 1, "n_pts_fix": 7,
                                                "ref_cam_weight": 2.5}),
        "persp_RT": ("perspective", 4, 50, 3, {"correction_params": ["R", "T"], "n_cam_fix": 1}),
        "rpc_RT": ("rpc", 4, 60, 3, {"correction_params": ["R", "T"], "n_cam_fix": 0}),
        "rpc_R": ("rpc", 3, 40, 2, {"correction_params": ["R"], "n_cam_fix": 1}),
    }
    for name, (model, M, N, opp, d) in cases.items():
        scene = synth.make_scene(model, M, N, opp, seed=7)
        p = ref_params(scene, dict(d, reduce=False))
        rng = np.random.default_rng(11)
        n_c = p.n_cam * p.n_params
        vs, rs = [], []
        for k in range(3):
            v = p.params_opt.copy()
            v[:n_c] += rng.normal(0, 1e-6, n_c) * (k > 0)
            v[n_c:] += rng.normal(0, 1.0, v.size - n_c) * (k > 0)
            vs.append(v.copy())
            rs.append(ref.ba_core.fun(v.copy(), p))
        out = dict(v=np.array(vs), r=np.array(rs), params_opt=p.params_opt, cam_params=p.cam_params,
                   pts_ind=p.pts_ind, cam_ind=p.cam_ind, pts2d=p.pts2d, pts2d_w=p.pts2d_w,
                   pts3d=scene.pts3d, n_params=p.n_params)
        if model == "rpc":
            # float64 re-evaluation of the same chain (the reference stores float32, ba_core.py:150) for Jacobian checks
            def f64(v, p=p):
                pts3d, cam_params = p.get_vars_ready_for_fun(v.copy())
                X = ref.ba_core.adjust_pts3d(pts3d[p.pts_ind], cam_params[p.cam_ind])
                proj = np.zeros((p.pts_ind.size, 2))
                for c in np.unique(p.cam_ind).tolist():
                    sel = p.cam_ind == c
                    proj[sel] = ref.cam_utils.apply_rpc_projection(p.cameras[c], X[sel])
                return np.repeat(p.pts2d_w, 2) * (proj - p.pts2d).ravel()
            fref = f64
            out["r64"] = np.array([f64(v) for v in vs])
        else:
            fref = lambda v, p=p: ref.ba_core.fun(v.copy(), p)  # noqa: E731
        Jc, Jp = fd3_jacobian_blocks(fref, vs[1], p, p.n_params)
        A = ref.ba_core.build_jacobian_sparsity(p).tocsr()
        out.update(Jc=Jc, Jp=Jp, A_indices=A.indices, A_indptr=A.indptr, A_shape=np.array(A.shape))
        save("fun_" + name, **out)


def golden_params():
    """G4 / G5: packing incl. reduce, fixed cams / points, ref_cam_weight; unpack / reconstruct round trip."""
    scene = synth.make_affine_scene(7, 80, 3, seed=3)
    # make the reduce step do something: cameras 0-2 are frozen; blank a few tracks out of the free cameras,
    # and give camera 1 no observation at all
    C = scene.to_dense_C()
    C[8:, :10] = np.nan
    C[2:4, :] = np.nan
    n_pts_fix = 10
    d = {"n_cam_fix": 3, "n_pts_fix": n_pts_fix, "reduce": True, "verbose": False, "correction_params": ["R", "T"],
         "ref_cam_weight": 3.0}
    pairs = [(0, 1), (0, 2), (2, 5), (4, 6)]
    p = ref.ba_params.BundleAdjustmentParameters(C, scene.pts3d.astype(np.float32), scene.cameras, "affine", pairs,
                                                 scene.camera_centers, d)
    v = p.params_opt.copy()
    rng = np.random.default_rng(5)
    v += rng.normal(0, 1e-3, v.size)
    v_in = v.copy()
    pts3d_u, cam_params_u = p.get_vars_ready_for_fun(v)
    corrected_pts3d, corrected_cameras = p.reconstruct_vars(v.copy(), scene.pts3d.astype(np.float32), scene.cameras)
    save("params_affine_reduce", C=C, pts3d=scene.pts3d.astype(np.float32), cameras=np.array(scene.cameras),
         pairs=np.array(pairs), C_red=p.C, pts3d_red=p.pts3d, cam_params=p.cam_params, pts_ind=p.pts_ind,
         cam_ind=p.cam_ind, pts2d=p.pts2d, pts2d_w=p.pts2d_w, params_opt=p.params_opt,
         counters=np.array([p.n_cam, p.n_pts, p.n_cam_fix, p.n_pts_fix, p.n_cam_opt, p.n_pts_opt, p.n_obs, p.n_params]),
         cam_prev=p.cam_prev_indices, pts_prev=p.pts_prev_indices, pairs_red=np.array(p.pairs_to_triangulate),
         v_in=v_in, v_after=v, pts3d_u=pts3d_u, cam_params_u=cam_params_u, corrected_pts3d=corrected_pts3d,
         corrected_cameras=np.array(corrected_cameras), cameras_ba=np.array(p.cameras_ba),
         est_R=np.array([e["R"] for e in p.estimated_params]), est_T=np.array([e["T"] for e in p.estimated_params]))

    # perspective + rpc camera packing (G8 includes the matrices of ref:tests/test_functions.py:20-38)
    P_persp = np.array([[7.29623172e-02, -5.17799277e-02, -1.02734764e-02, -9.62027582e04],
                        [-5.01011603e-02, -6.23291457e-02, -4.15721807e-02, -2.59250341e05],
                        [2.78193760e-08, 7.15619726e-08, -1.43761111e-07, 1.00000000e00]])
    P_aff = np.array([[7.61064055e-01, -9.35843155e-01, -1.00554841e-01, -1.13554311e06],
                      [6.65950776e-02, -7.40405784e-02, 1.36333044e00, 4.07093217e06],
                      [0.0, 0.0, 0.0, 1.0]])
    cp_persp = ref.ba_params.load_cam_params_from_camera(P_persp, None, "perspective")
    cp_aff = ref.ba_params.load_cam_params_from_camera(P_aff, None, "affine")
    cp_rpc = ref.ba_params.load_cam_params_from_camera(None, np.array([1.0, 2.0, 3.0]), "rpc")
    save("cam_params", P_persp=P_persp, P_aff=P_aff, cp_persp=cp_persp, cp_aff=cp_aff, cp_rpc=cp_rpc,
         P_persp_back=ref.ba_params.load_camera_from_cam_params(cp_persp, "perspective"),
         P_aff_back=ref.ba_params.load_camera_from_cam_params(cp_aff, "affine"),
         rpc_back=ref.ba_params.load_camera_from_cam_params(cp_rpc, "rpc"))
    r = np.arange(12, dtype=float) - 5.5
    w = np.array([1.0, 2.0, 1.0, 4.0, 1.0, 0.5])
    save("reproj_err", r=r, w=w, err_w=ref.ba_core.compute_reprojection_error(r, w),
         err=ref.ba_core.compute_reprojection_error(r),
         track_err=ref.ba_core.compute_mean_reprojection_error_per_track(
             np.array([1.0, 2.0, 3.0, 4.0, 5.0]), np.array([0, 0, 1, 2, 2]), np.array([0, 1, 1, 0, 2])))


def golden_solves():
    """G6 (tight protocol) and G7 (as-shipped run_ba_optimization) on small and C2-shape problems."""
    import io
    from contextlib import redirect_stdout

    specs = [
        # name, model, M, N, obs/pt, seed, d, losses
        ("affine_small_R", "affine", 6, 400, 4, 2, {"correction_params": ["R"], "n_cam_fix": 1}, ["linear", "soft_l1"]),
        ("affine_small_RT", "affine", 6, 400, 4, 2, {"correction_params": ["R", "T"], "n_cam_fix": 1}, ["linear"]),
        ("persp_small_R", "perspective", 5, 300, 4, 4, {"correction_params": ["R"], "n_cam_fix": 1}, ["linear"]),
        ("affine_C2_R", "affine", 10, 5000, 6, 1, {"correction_params": ["R"], "n_cam_fix": 1}, ["linear", "soft_l1"]),
    ]
    for name, model, M, N, opp, seed, d, losses in specs:
        scene = synth.make_scene(model, M, N, opp, seed=seed)
        out = dict(n_obs=scene.n_obs)
        for loss in losses:
            p = ref_params(scene, dict(d, reduce=False))
            res = tight_solve(p, loss=loss, max_nfev=200)
            print(name, loss, "tight: status", res.status, "nfev", res.nfev, "cost %.10f" % res.cost)
            out.update({"tight_x_" + loss: res.x, "tight_fun_" + loss: res.fun,
                        "tight_stats_" + loss: np.array([res.cost, res.nfev, res.status, res.optimality])})
            p = ref_params(scene, dict(d, reduce=False))
            with redirect_stdout(io.StringIO()):
                vars_init, vars_ba, err_init, err_ba, iters = ref.ba_core.run_ba_optimization(
                    p, {"loss": loss, "verbose": 0}, False, False)
            print(name, loss, "as shipped: nfev", iters, "mean err %.6f -> %.6f" % (err_init.mean(), err_ba.mean()))
            out.update({"ship_x_" + loss: vars_ba, "ship_err_init_" + loss: err_init, "ship_err_" + loss: err_ba,
                        "ship_iters_" + loss: iters})
        save("solve_" + name, **out)

    # RPC: config-1-like plumbing case (2 shipped RPCs, synthetic tracks), as-shipped soft_l1 then L2 (ba_pipeline.py:706-712)
    scene = synth.make_rpc_scene(2, 2000, 2, seed=1, sigma_theta=5e-6)
    p = ref_params(scene, {"correction_params": ["R"], "reduce": False})
    with redirect_stdout(io.StringIO()):
        _, v1, e0, e1, it1 = ref.ba_core.run_ba_optimization(p, {"loss": "soft_l1", "f_scale": 1.0, "max_iter": 300,
                                                                 "verbose": 0}, False, False)
        p.params_opt = v1.copy()
        _, v2, _, e2, it2 = ref.ba_core.run_ba_optimization(p, {"verbose": 0}, False, False)
    print("rpc config1-like: err %.4f -> %.4f -> %.4f, nfev %d + %d" % (e0.mean(), e1.mean(), e2.mean(), it1, it2))
    save("solve_rpc_config1", x_softl1=v1, x_l2=v2, err_init=e0, err_softl1=e1, err_l2=e2, iters=np.array([it1, it2]))


def golden_solves_round4():
    """Round 4: the tight protocol at the correction mode bench.py times (R+T) on a BASELINE shape, and a gauge-free case
    (no frozen camera).  Both have flat directions: only gauge-invariant outputs are compared (cost, residual vector, errors)."""
    specs = [
        ("affine_C2_RT", "affine", 10, 5000, 6, 1, {"correction_params": ["R", "T"], "n_cam_fix": 1}, ["linear"]),
        ("affine_small_free", "affine", 6, 400, 4, 2, {"correction_params": ["R"], "n_cam_fix": 0}, ["linear"]),
    ]
    for name, model, M, N, opp, seed, d, losses in specs:
        scene = synth.make_scene(model, M, N, opp, seed=seed)
        out = dict(n_obs=scene.n_obs)
        for loss in losses:
            p = ref_params(scene, dict(d, reduce=False))
            res = tight_solve(p, loss=loss, max_nfev=300)
            print(name, loss, "tight: status", res.status, "nfev", res.nfev, "cost %.10f" % res.cost, "optimality %.3e" % res.optimality)
            out.update({"tight_x_" + loss: res.x, "tight_fun_" + loss: res.fun,
                        "tight_stats_" + loss: np.array([res.cost, res.nfev, res.status, res.optimality])})
        save("solve_" + name, **out)



def rpc_fun_f64(p):
    """The reference's rpc residual chain with the float32 store of ba_core.py:150 left out."""
    def f64(v):
        pts3d, cam_params = p.get_vars_ready_for_fun(v.copy())
        X = ref.ba_core.adjust_pts3d(pts3d[p.pts_ind], cam_params[p.cam_ind])
        proj = np.zeros((p.pts_ind.size, 2))
        for c in np.unique(p.cam_ind).tolist():
            sel = p.cam_ind == c
            proj[sel] = ref.cam_utils.apply_rpc_projection(p.cameras[c], X[sel])
        return np.repeat(p.pts2d_w, 2) * (proj - p.pts2d).ravel()
    return f64


def golden_tight_rpc_persp():
    """
    G6 for the two models round 1 left without a tight-protocol vector: rpc (["R"], one frozen camera; the reference's
    functions evaluated in float64, i.e. without the float32 store of ba_core.py:150, which a finite-difference Jacobian
    cannot see through) and perspective R+T.
    """
    from scipy.optimize import least_squares

    specs = [
        # every camera sees every point: the two shipped RPCs alternate over the cameras, so a point seen only by cameras of
        # the same parity has no parallax and the problem a flat valley (tight scipy and an exact LM then stop at different
        # points of it)
        ("rpc_small_R", "rpc", 4, 300, 4, 5, {"correction_params": ["R"], "n_cam_fix": 1}, ["linear", "soft_l1"]),
        ("persp_small_RT", "perspective", 5, 300, 4, 4, {"correction_params": ["R", "T"], "n_cam_fix": 1}, ["linear"]),
    ]
    for name, model, M, N, opp, seed, d, losses in specs:
        kw = {"sigma_theta": 5e-6} if model == "rpc" else {}
        scene = synth.make_scene(model, M, N, opp, seed=seed, **kw)
        out = dict(n_obs=scene.n_obs)
        for loss in losses:
            p = ref_params(scene, dict(d, reduce=False))
            f = rpc_fun_f64(p) if model == "rpc" else (lambda v, p=p: ref.ba_core.fun(v.copy(), p))
            A = ref.ba_core.build_jacobian_sparsity(p)
            res = least_squares(f, p.params_opt.copy(), jac_sparsity=A, x_scale="jac", method="trf", loss=loss,
                                f_scale=1.0, ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=300,
                                tr_options={"atol": 1e-12, "btol": 1e-12})
            print(name, loss, "tight: status", res.status, "nfev", res.nfev, "cost %.10f" % res.cost)
            out.update({"tight_x_" + loss: res.x, "tight_fun_" + loss: res.fun,
                        "tight_stats_" + loss: np.array([res.cost, res.nfev, res.status, res.optimality])})
            if model == "rpc":
                # scipy's forward differences (step 1.5e-8 x max(1, |x|): 9 cm on ECEF coordinates) are biased by the
                # curvature of the cubic RPC chain: the stationary point of J_fd^T f = 0 sits 1e-3 (relative) away from the
                # minimiser in the angles (measured here: exact-Jacobian LM vs this run).  The same solver with scipy's own
                # jac="3-point" option removes the bias; both solutions are stored.
                res3 = least_squares(f, p.params_opt.copy(), jac="3-point", jac_sparsity=A, x_scale="jac", method="trf",
                                     loss=loss, f_scale=1.0, ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=300,
                                     tr_options={"atol": 1e-12, "btol": 1e-12})
                print(name, loss, "tight, 3-point FD: status", res3.status, "nfev", res3.nfev, "cost %.10f" % res3.cost)
                out.update({"tight3_x_" + loss: res3.x, "tight3_fun_" + loss: res3.fun,
                            "tight3_stats_" + loss: np.array([res3.cost, res3.nfev, res3.status, res3.optimality])})
        save("solve_" + name, **out)


def golden_tight3():
    """
    Round 5: the tight protocol with scipy's own jac="3-point" option for EVERY solve case (rpc had such vectors since round 2, in
    solve_rpc_small_R.npz; the ones written here are converged further, see below).  scipy's default forward differences (relative step 1.5e-8) put the stationary point of J_fd^T f = 0 a little
    away from the minimiser of the reference's own `fun`; central differences (relative step 6e-6, error O(h^2) x third
    derivative) remove that bias, so these vectors say where the reference's cost function has its minimum -- which is what an
    exact-Jacobian solver must reproduce to 1e-6.  Separate file: the 2-point vectors of rounds 1-4 stay byte for byte.
    """
    from scipy.optimize import least_squares

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cases

    out = {}
    for name, (model, M, N, opp, seed, d, losses) in cases.SOLVE_CASES.items():
        scene = synth.make_scene(model, M, N, opp, seed=seed, **cases.SCENE_KW.get(name, {}))
        for loss in losses:
            p = ref_params(scene, dict(d, reduce=False))
            A = ref.ba_core.build_jacobian_sparsity(p)
            kw = dict(jac="3-point", jac_sparsity=A, x_scale="jac", method="trf", loss=loss, f_scale=1.0,
                      tr_options={"atol": 1e-12, "btol": 1e-12})
            if model == "rpc":
                # rpc (the reference chain in float64, rpc_fun_f64): scipy's xtol test compares |dx| with xtol |x|, and |x| is
                # dominated by ECEF coordinates (1e7 m) while the unknown angles are 1e-5 rad -- the tight run of round 2 stops with
                # the angles 2e-10 rad (2.7e-6 relative) short of the minimum (its gradient there is 100 x the one at the point an
                # exact LM reaches, and its cost higher: tools/tight_metrics.py).  Here the step and cost tests are switched off
                # (None) and the same solver runs a fixed number of evaluations: the reference's cost function minimised as far as
                # its own solver takes it.
                res3 = least_squares(rpc_fun_f64(p), p.params_opt.copy(), ftol=None, xtol=None, gtol=1e-15,
                                     max_nfev=60 if loss == "linear" else 200, **kw)
            else:
                res3 = least_squares(ref.ba_core.fun, p.params_opt.copy(), ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=600,
                                     args=(p,), **kw)
            print(name, loss, "tight, 3-point FD: status", res3.status, "nfev", res3.nfev, "cost %.12f" % res3.cost,
                  "optimality %.3e" % res3.optimality, flush=True)
            # Round 6: the same solver RESTARTED from its own end point until it returns the point it was started from.  scipy's
            # trust radius only shrinks near the end of a tight run (every step at rounding level is rejected), and the run stops on
            # xtol with whatever gradient is left: affine_C2_R / soft_l1 ended with optimality 5.2e3 (the linear run of the same scene:
            # 12.6), 8.4e-7 of |f| from the reference's own forward-difference run.  With the step and cost tests switched off
            # (ftol = xtol = None, 150 / 200 / 300 evaluations: the way the rpc cases are treated above) it stays on exactly that point --
            # every further trial is rejected --, but a fresh call (fresh radius, fresh x_scale) takes four more evaluations to
            # optimality 29.6, cost lower by 4.6e-9, 1.34e-6 of |f| away, and the next call returns its start: the reference's
            # stationary point.  An exact-Jacobian LM (CPU oracle) ends 2.5e-8 of |f| from it (1.3e-6 from the first end point).
            # The other cases return their start at once or move by 3e-9 of |f|.
            if model != "rpc":
                fun3 = ref.ba_core.fun
                for restart in range(8):
                    nxt = least_squares(fun3, res3.x.copy(), ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=600, args=(p,), **kw)
                    same = np.array_equal(nxt.x, res3.x)
                    moved = np.linalg.norm(nxt.fun - res3.fun) / np.linalg.norm(res3.fun)
                    if not same:
                        print("   restart", restart, "nfev", nxt.nfev, "cost %.12f" % nxt.cost, "optimality %.3e" % nxt.optimality,
                              "moved %.2e of |f|" % moved, flush=True)
                        nxt.nfev += res3.nfev
                        res3 = nxt
                    else:
                        break
            key = name + "_" + loss
            out.update({"x_" + key: res3.x, "fun_" + key: res3.fun,
                        "stats_" + key: np.array([res3.cost, res3.nfev, res3.status, res3.optimality])})
    save("solve_tight3", **out)


def golden_outliers():
    """
    ref:bundle_adjust/ba_outliers.py get_elbow_value / compute_obs_to_remove (pure numpy, importable) on the reprojection
    errors of a scene with injected gross errors: thresholds per camera and the removed set, plus single-vector elbows.
    """
    rng = np.random.default_rng(21)
    out = {}
    for name, M, N, opp, frac in (("a", 6, 600, 4, 0.03), ("b", 12, 1500, 5, 0.10), ("c", 3, 40, 2, 0.0)):
        scene = synth.make_affine_scene(M, N, opp, seed=13, sigma_theta=2e-6)
        p = ref_params(scene, {"correction_params": ["R"], "n_cam_fix": 0, "reduce": False})
        pts2d = p.pts2d.copy()
        bad = rng.random(p.n_obs) < frac
        p.pts2d[bad] += rng.normal(0, 25.0, (int(bad.sum()), 2))
        r = ref.ba_core.fun(p.params_opt.copy(), p)
        err = ref.ba_core.compute_reprojection_error(r, p.pts2d_w)
        C_new, cam_thr, n = ref.ba_outliers.compute_obs_to_remove(err, p)
        removed = np.isnan(C_new[2 * p.cam_ind, p.pts_ind]) & ~np.isnan(p.C[2 * p.cam_ind, p.pts_ind])
        C2, thr2, n2 = ref.ba_outliers.compute_obs_to_remove(err, p, predef_thr=3.14159)
        removed2 = np.isnan(C2[2 * p.cam_ind, p.pts_ind])
        out.update({name + "_pts2d": p.pts2d, name + "_err": err, name + "_cam_thr": np.array(cam_thr), name + "_n": n,
                    name + "_removed": removed, name + "_thr_predef": np.array(thr2), name + "_removed_predef": removed2,
                    name + "_bad": bad})
        print("outliers", name, "obs", p.n_obs, "injected", int(bad.sum()), "removed", n, "thr", cam_thr)
    vecs, elbows = [], []
    for k in range(6):
        n = [5, 17, 200, 1, 2, 1000][k]
        v = np.abs(rng.normal(0, 1, n)) + (rng.random(n) < 0.1) * rng.uniform(5, 50, n)
        e, ok = ref.ba_outliers.get_elbow_value(v) if n > 1 else (v[0], True)
        vecs.append(np.pad(v, (0, 1000 - n), constant_values=np.nan))
        elbows.append([e, float(ok), n])
    out.update(elbow_vecs=np.array(vecs), elbow_out=np.array(elbows))
    save("outliers", **out)


def import_reference_triangulation():
    """
    ref:bundle_adjust/feature_tracks/ft_triangulate.py imported in place.  What it needs and the image lacks:
      * cv2.triangulatePoints -> the restatement in oracle/triangulate_oracle.py (OpenCV is absent: parity of that one call is unpinned);
      * lib/disp_to_h.so -> the reference's own C compiled by oracle/Makefile into oracle/_ref/disp_to_h.so; the reference's ctypes
        binding (s2p/triangulation.py) is used unchanged, only the path it opens is redirected;
      * s2p/geographiclib (pyproj, geojson): only its pyproj_crs() is touched, to name the default output CRS.
    """
    import ctypes
    sys.path.insert(0, ROOT)
    from oracle import triangulate_oracle as T

    cv2 = types.ModuleType("cv2")

    def triangulate_points(P1, P2, a, b):
        X = T.linear_triangulation_multiple_pts(P1, P2, np.asarray(a).T, np.asarray(b).T)
        return np.vstack([X.T, np.ones((1, X.shape[0]))])
    cv2.triangulatePoints = triangulate_points
    sys.modules["cv2"] = cv2
    s2p = types.ModuleType("bundle_adjust.s2p")
    s2p.__path__ = ["/root/reference/bundle_adjust/s2p"]
    sys.modules["bundle_adjust.s2p"] = s2p
    geo = types.ModuleType("bundle_adjust.s2p.geographiclib")
    geo.pyproj_crs = lambda name: name
    sys.modules["bundle_adjust.s2p.geographiclib"] = geo
    s2p.geographiclib = geo
    ft = types.ModuleType("bundle_adjust.feature_tracks")
    ft.__path__ = ["/root/reference/bundle_adjust/feature_tracks"]
    sys.modules["bundle_adjust.feature_tracks"] = ft
    real_cdll = ctypes.CDLL

    def cdll(path, *a, **k):
        if str(path).endswith("disp_to_h.so"):
            path = os.path.join(ROOT, "oracle", "_ref", "disp_to_h.so")
        return real_cdll(path, *a, **k)
    ctypes.CDLL = cdll
    try:
        return importlib.import_module("bundle_adjust.feature_tracks.ft_triangulate")
    finally:
        ctypes.CDLL = real_cdll


def tri_pairs(M, rng, ok=lambda i, j: True):
    """pairs_to_triangulate as a pipeline could hand them over, plus the cases the loop has to survive: arbitrary order, a pair listed
    twice, a reversed pair, a pair naming a camera the matrix does not have"""
    pairs = [(i, j) for i in range(M) for j in range(i + 1, M) if rng.random() < 0.6 and ok(i, j)]
    rng.shuffle(pairs)
    pairs = [tuple(int(v) for v in pr) for pr in pairs]
    pairs.insert(3, pairs[0])
    pairs.insert(5, (pairs[1][1], pairs[1][0]))
    pairs.insert(2, (0, M + 1))
    return pairs


def golden_init_pts3d():
    """ref:bundle_adjust/feature_tracks/ft_triangulate.py:57-127 on seeded scenes of the three camera models."""
    tri = import_reference_triangulation()
    out = {}
    for name, model, M, N, opp in (("affine", "affine", 7, 400, 4), ("persp", "perspective", 6, 300, 3), ("rpc", "rpc", 5, 200, 3)):
        scene = synth.make_scene(model, M, N, opp, seed=31)
        C = scene.to_dense_C()
        C[:, 5] = np.nan; C[0:2, 5] = [10.0, 20.0]       # a track seen by one camera only: stays (0, 0, 0)
        # the rpc scene cycles two real RPCs over its cameras: only pairs of different models have a baseline
        pairs = tri_pairs(M, np.random.default_rng(7), (lambda i, j: (i + j) % 2 == 1) if model == "rpc" else (lambda i, j: True))
        pts = tri.init_pts3d(C, scene.cameras, model, pairs, verbose=False)
        c_i, c_j = pairs[0]
        t = np.where(~np.isnan(C[2 * c_i]) & ~np.isnan(C[2 * c_j]))[0]
        oi, oj = C[2 * c_i:2 * c_i + 2, t].T, C[2 * c_j:2 * c_j + 2, t].T
        if model == "rpc":
            pw, err = tri.rpc_triangulation(scene.cameras[c_i], scene.cameras[c_j], oi, oj)
            out[name + "_pair_err"] = err
        else:
            pw = tri.linear_triangulation_multiple_pts(scene.cameras[c_i], scene.cameras[c_j], oi, oj)
        out.update({name + "_C": C, name + "_pairs": np.array(pairs), name + "_pts3d": pts, name + "_pair_obs_i": oi, name + "_pair_obs_j": oj,
                    name + "_pair_pts3d": pw})
        d = np.linalg.norm(pts[t].astype(np.float64) - np.asarray(scene.pts3d_true if hasattr(scene, "pts3d_true") else scene.pts3d)[t], axis=1)
        print("init_pts3d", name, pts.dtype, pts.shape, "pairs", len(pairs), "median distance to the scene points", np.median(d))
    save("init_pts3d", **out)


def rpcfit_cases():
    """(name, target (n, 2), input_locs (n, 3)) of the RPC re-fit fixtures: a 10 x 10 x 10 grid over the footprint of a shipped RPC,
    projected through the RPC behind a small corrective rotation (what ba_rpcfit.fit_Rt_corrected_rpc feeds to weighted_lsq), and a
    7 x 7 x 7 grid through an affine camera (fit_rpc_from_projection_matrix)."""
    from satba import geo_utils as G
    from satba.ba_core import adjust_pts3d
    from satba.rpc_model import RPCModel

    out = []
    for k, f in enumerate(synth.default_rpc_files()):
        r = RPCModel.from_file(f)
        lon, lat, alt = [a.reshape(-1) for a in np.meshgrid(np.linspace(r.lon_offset - 0.9 * r.lon_scale, r.lon_offset + 0.9 * r.lon_scale, 10),
                                                            np.linspace(r.lat_offset - 0.9 * r.lat_scale, r.lat_offset + 0.9 * r.lat_scale, 10),
                                                            np.linspace(r.alt_offset - r.alt_scale, r.alt_offset + r.alt_scale, 10), indexing="ij")]
        X = np.stack(G.latlon_to_ecef_custom(lat, lon, alt), 1)
        up = X.mean(0) / np.linalg.norm(X.mean(0))
        Rt = np.concatenate([[3e-6 * (k + 1), -2e-6, 4e-6], [0.0, 0.0, 0.0], X.mean(0) + 5e5 * up]).reshape(1, 9)
        Xa = adjust_pts3d(X, Rt)
        la, lo, al = G.ecef_to_latlon_custom(Xa[:, 0], Xa[:, 1], Xa[:, 2])
        col, row = r.projection(lo, la, al)
        out.append(("rpc%d" % k, np.stack([col, row], 1), np.stack([lon, lat, alt], 1)))
    scene = synth.make_scene("affine", 3, 50, 3, seed=4)
    la, lo, al = G.ecef_to_latlon_custom(*scene.pts3d_true.T)
    lon, lat, alt = [a.reshape(-1) for a in np.meshgrid(np.linspace(lo.min(), lo.max(), 7), np.linspace(la.min(), la.max(), 7),
                                                        np.linspace(al.min() - 500, al.max() + 500, 7), indexing="ij")]
    X = np.stack(G.latlon_to_ecef_custom(lat, lon, alt), 1)
    P = np.asarray(scene.cameras[0])
    proj = P @ np.hstack([X, np.ones((len(X), 1))]).T
    out.append(("affine", (proj[:2] / proj[2]).T, np.stack([lon, lat, alt], 1)))
    return out


def golden_rpcfit():
    """ref:bundle_adjust/ba_rpcfit.py:88-153 weighted_lsq (+ check_errors) through the imported reference module."""
    from satba.rpc_model import RPCModel

    class RefRpc(RPCModel):  # rpcm.RPCModel as ba_rpcfit.initialize_rpc uses it: built from a dict of "0" strings, then filled
        def __init__(self, d, dict_format="geotiff"):
            for a in ("row_offset", "col_offset", "lat_offset", "lon_offset", "alt_offset", "row_scale", "col_scale", "lat_scale",
                      "lon_scale", "alt_scale"):
                setattr(self, a, 0.0)
            for a in ("row_num", "row_den", "col_num", "col_den"):
                setattr(self, a, [0.0] * 20)
    sys.modules["rpcm"].RPCModel = RefRpc
    fit = importlib.import_module("bundle_adjust.ba_rpcfit")
    out = {}
    for name, target, locs in rpcfit_cases():
        r = fit.weighted_lsq(target.copy(), locs.copy())
        err = fit.check_errors(r, locs, target)
        out.update({name + "_target": target, name + "_locs": locs, name + "_err": err, name + "_table": r.to_table()})
        print("rpcfit", name, "n", len(target), "fit error max / median [px]: %.3g / %.3g" % (err.max(), np.median(err)))
    save("rpcfit", **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["fun", "params", "solves", "solves4", "tight2", "tight3", "outliers", "init_pts3d", "rpcfit"]
    if "fun" in which:
        golden_fun_and_jac()
    if "params" in which:
        golden_params()
    if "solves" in which:
        golden_solves()
    if "solves4" in which:
        golden_solves_round4()
    if "tight2" in which:
        golden_tight_rpc_persp()
    if "tight3" in which:
        golden_tight3()
    if "outliers" in which:
        golden_outliers()
    if "init_pts3d" in which:
        golden_init_pts3d()
    if "rpcfit" in which:
        golden_rpcfit()
