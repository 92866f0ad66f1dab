"""
Pins the CPU oracle (oracle/) and the host-side packing code (satba.ba_params, satba.cam_utils, satba.ba_rotate)
on vectors captured from the reference itself by tools/gen_golden.py.  No GPU.
"""
import ctypes
import os

import numpy as np
import pytest

import cases
from oracle import ba_oracle as O
from oracle import lm_oracle as L
from satba import ba_core, ba_params, ba_rotate, cam_utils, geo_utils, synth
from satba.rpc_model import RPCModel


@pytest.mark.parametrize("name", list(cases.FUN_CASES))
def test_oracle_fun_reproduces_reference(name):
    _, p, g = cases.fun_case(name)
    assert np.array_equal(p.pts_ind, g["pts_ind"]) and np.array_equal(p.cam_ind, g["cam_ind"])
    assert np.array_equal(p.pts2d, g["pts2d"]) and np.array_equal(p.pts2d_w, g["pts2d_w"])
    assert np.allclose(p.params_opt, g["params_opt"], rtol=1e-14, atol=0)  # perspective T goes through an RQ + solve
    for k in range(3):
        r = O.fun(g["v"][k], p)
        # bit-exact for affine / rpc; perspective differs by the last bit of the RQ-decomposed T (1e-9 m on 6e6 m)
        assert np.abs(r - g["r"][k]).max() <= (1e-9 if p.cam_model == "perspective" else 0.0)


@pytest.mark.parametrize("name", list(cases.FUN_CASES))
def test_dense_and_sparse_constructors_agree(name):
    _, p_sparse, g = cases.fun_case(name)
    _, p_dense, _ = cases.fun_case(name, dense=True)
    for attr in ("pts_ind", "cam_ind", "pts2d", "pts2d_w", "params_opt", "cam_params"):
        assert np.array_equal(getattr(p_sparse, attr), getattr(p_dense, attr)), attr
    assert p_sparse.n_obs == p_dense.n_obs and p_sparse.n_params == p_dense.n_params == int(g["n_params"])


@pytest.mark.parametrize("name", list(cases.FUN_CASES))
def test_oracle_analytic_jacobian_vs_reference_finite_differences(name):
    _, p, g = cases.fun_case(name)
    v = ba_core._frozen_vars(g["v"][1].copy(), p)
    _, _, _, Jc, Jp = L.weighted_system(v, p, rpc_f32=False)
    assert np.abs(Jc - g["Jc"]).max() < 1e-7 * np.abs(g["Jc"]).max()
    assert np.abs(Jp - g["Jp"]).max() < 1e-7 * np.abs(g["Jp"]).max()


@pytest.mark.parametrize("name", list(cases.FUN_CASES))
def test_jacobian_sparsity_pattern(name):
    _, p, g = cases.fun_case(name)
    for A in (O.jacobian_sparsity(p), ba_core.build_jacobian_sparsity(p)):
        assert tuple(A.shape) == tuple(g["A_shape"])
        A.sort_indices()
        assert np.array_equal(A.indptr, g["A_indptr"]) and np.array_equal(A.indices, g["A_indices"])


@pytest.mark.parametrize("name", ["affine_small_R", "persp_small_R"])
def test_oracle_scipy_driver_reproduces_reference_runs(name):
    """Same scipy, same fun, same call: the oracle's as-shipped and tight runs must land on the reference's."""
    _, make_p, g, losses = cases.solve_case(name)
    loss = losses[0]
    p = make_p()
    res = O.solve_scipy(p, {"loss": loss})
    if p.cam_model == "affine":  # inputs are bit-identical to the reference's: so is the whole scipy run
        assert res.nfev == int(g["ship_iters_" + loss])
        assert np.allclose(res.x, g["ship_x_" + loss], rtol=1e-9, atol=1e-9)
    else:  # perspective T differs in the last bit (RQ + solve): the early-stopping path may take one more step
        assert abs(res.nfev - int(g["ship_iters_" + loss])) <= 1
        assert abs(O.reprojection_error(res.fun, p.pts2d_w).mean() - g["ship_err_" + loss].mean()) < 1e-3
    res = O.solve_scipy(make_p(), {"loss": loss, "max_iter": 200}, tight=True)
    st = g["tight_stats_" + loss]
    assert abs(res.cost - st[0]) < 1e-10 * st[0] and res.status == int(st[2])


@pytest.mark.parametrize("name", ["rpc_small_R", "persp_small_RT"])
def test_oracle_scipy_driver_reproduces_tight_rpc_and_perspective(name):
    """The tight-protocol vectors of the two models round 1 had none for (rpc in float64, perspective R+T)."""
    _, make_p, g, losses = cases.solve_case(name)
    loss = losses[0]
    res = O.solve_scipy(make_p(), {"loss": loss, "max_iter": 300}, tight=True, rpc_store_dtype=np.float64)
    st = g["tight_stats_" + loss]
    assert abs(res.cost - st[0]) < 1e-9 * st[0] and res.status == int(st[2])
    p = make_p()
    n_c = p.n_cam * p.n_params
    # Only cost and status are reproducible between two forward-difference runs of these two cases: last-bit differences
    # of the residuals (numpy restatement; perspective T from an RQ + solve) are amplified by the differencing and move
    # the stopping point along weakly determined directions (rpc angles by 2e-3 relative, perspective T by kilometres
    # along the line of sight) at equal cost.  The exact-Jacobian engine below lands on the reference's vectors.
    # the exact-Jacobian CPU engine (what the device implements) against the reference run
    e = L.OracleEngine(p, rpc_f32=False)
    e.set_x(p.params_opt.copy())
    from satba import trf
    r2 = trf.trf_solve(e, ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=300, loss=loss)
    key = "tight3_" if p.cam_model == "rpc" else "tight_"  # rpc: forward differences bias the reference run (gen_golden.py)
    xt, ft = g[key + "x_" + loss], g[key + "fun_" + loss]
    assert abs(r2.cost - g[key + "stats_" + loss][0]) < 1e-9 * r2.cost
    assert np.linalg.norm(e.residuals() - ft) < 1e-6 * np.linalg.norm(ft)
    assert np.abs(e.get_x()[:n_c] - xt[:n_c]).max() < (1e-5 if p.cam_model == "rpc" else 1e-6) * np.abs(xt[:n_c]).max()


@pytest.mark.parametrize("name,loss", [("affine_small_R", "linear"), ("affine_small_RT", "linear"), ("persp_small_R", "linear"),
                                       ("affine_small_free", "linear")])
def test_reference_forward_differences_displace_its_own_minimum(name, loss):
    """
    Round 5: why the parity tests hold the solver to the reference's 3-point run (tests/golden/solve_tight3.npz) and not to its default
    forward-difference run.  (1) The reference's OWN two tight runs -- same `fun`, same scipy, jac="2-point" against jac="3-point" --
    sit 5e-7 .. 3e-6 of |f| apart: that is the bias of a forward-difference Jacobian (relative step 1.5e-8), not of the solver
    compared with it.  (2) The exact-Jacobian LM of this repository (host loop of satba/trf.py over the CPU oracle's engine: the
    loop the device kernels are held bit-identical to) lands on the 3-point run to better than 1e-7 -- twenty times closer than the
    reference is to itself.
    """
    from satba import trf

    _, make_p, g, _ = cases.solve_case(name)
    g3 = cases.golden("solve_tight3")
    f2, f3 = g["tight_fun_" + loss], g3["fun_{}_{}".format(name, loss)]
    own = np.linalg.norm(f2 - f3) / np.linalg.norm(f3)
    assert 3e-7 < own < 5e-6, own
    p = make_p()
    eng = L.OracleEngine(p)
    res = trf.trf_solve(eng, ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=300, loss=loss)
    f = eng.residuals()
    mine = np.linalg.norm(f - f3) / np.linalg.norm(f3)
    assert mine < 1e-7 and mine < 0.05 * own, (mine, own)
    assert abs(res.cost - g3["stats_{}_{}".format(name, loss)][0]) < 1e-9 * res.cost
    if name not in cases.FLAT_CASES:
        n_c = p.n_cam * p.n_params
        x3 = g3["x_{}_{}".format(name, loss)]
        assert np.abs(eng.get_x()[:n_c] - x3[:n_c]).max() < 1e-8 * np.abs(x3[:n_c]).max()


def test_rpc_projection_against_reference_c():
    """oracle.rpc_projection and satba.RPCModel.projection vs the reference's own C evaluator (ref:c/rpc.c:442-452)."""
    lib_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "librpc.so")
    if not os.path.exists(lib_path):
        pytest.skip("oracle/_ref/librpc.so not built (needs /root/reference; run `make -C oracle`)")
    lib = ctypes.CDLL(lib_path)

    class Rpc(ctypes.Structure):  # struct rpc of ref:c/rpc.h:14-32
        _fields_ = [(n, ctypes.c_double * k) for n, k in (
            ("numx", 20), ("denx", 20), ("numy", 20), ("deny", 20), ("scale", 3), ("offset", 3), ("inumx", 20),
            ("idenx", 20), ("inumy", 20), ("ideny", 20), ("iscale", 3), ("ioffset", 3), ("dmval", 4), ("imval", 4))] + [
            ("delta", ctypes.c_double)]

    lib.eval_rpci.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(Rpc)] + [ctypes.c_double] * 3
    rng = np.random.default_rng(0)
    for path in synth.default_rpc_files():
        r = RPCModel.from_file(path)
        s = Rpc()  # attribute <-> struct mapping of ref:bundle_adjust/s2p/triangulation.py:43-61
        s.inumx[:], s.idenx[:], s.inumy[:], s.ideny[:] = r.col_num, r.col_den, r.row_num, r.row_den
        s.ioffset[:] = [r.lon_offset, r.lat_offset, r.alt_offset]
        s.iscale[:] = [r.lon_scale, r.lat_scale, r.alt_scale]
        s.offset[:] = [r.col_offset, r.row_offset, r.alt_offset]
        s.scale[:] = [r.col_scale, r.row_scale, r.alt_scale]
        lon = r.lon_offset + rng.uniform(-0.02, 0.02, 50)
        lat = r.lat_offset + rng.uniform(-0.02, 0.02, 50)
        alt = r.alt_offset + rng.uniform(-500, 500, 50)
        col, row = O.rpc_projection(r, lon, lat, alt)
        col2, row2 = r.projection(lon, lat, alt)
        out = (ctypes.c_double * 2)()
        for i in range(50):
            lib.eval_rpci(out, ctypes.byref(s), lon[i], lat[i], alt[i])
            assert abs(out[0] - col[i]) < 1e-9 and abs(out[1] - row[i]) < 1e-9
        assert np.abs(col - col2).max() < 1e-9 and np.abs(row - row2).max() < 1e-9


def test_ba_params_reduce_and_roundtrip_match_reference():
    g = cases.golden("params_affine_reduce")
    d = {"n_cam_fix": 3, "n_pts_fix": 10, "reduce": True, "verbose": False, "correction_params": ["R", "T"],
         "ref_cam_weight": 3.0}
    cams = [c for c in g["cameras"]]
    centers = [np.zeros(3) for _ in cams]
    p = ba_params.BundleAdjustmentParameters(g["C"], g["pts3d"], cams, "affine", [tuple(x) for x in g["pairs"]], centers, d)
    counters = [p.n_cam, p.n_pts, p.n_cam_fix, p.n_pts_fix, p.n_cam_opt, p.n_pts_opt, p.n_obs, p.n_params]
    assert [int(x) for x in counters] == [int(x) for x in g["counters"]]
    assert np.array_equal(np.isnan(p.C), np.isnan(g["C_red"])) and np.array_equal(np.nan_to_num(p.C), np.nan_to_num(g["C_red"]))
    assert p.pts3d.dtype == np.float32 and np.array_equal(p.pts3d, g["pts3d_red"])
    for attr, key in (("pts_ind", "pts_ind"), ("cam_ind", "cam_ind"), ("pts2d", "pts2d"), ("pts2d_w", "pts2d_w"),
                      ("cam_prev_indices", "cam_prev"), ("pts_prev_indices", "pts_prev")):
        assert np.array_equal(getattr(p, attr), g[key]), attr
    assert np.array_equal(np.array(p.pairs_to_triangulate), g["pairs_red"])
    assert np.allclose(p.cam_params, g["cam_params"], rtol=1e-14, atol=0)
    assert np.allclose(p.params_opt, g["params_opt"], rtol=1e-14, atol=0)
    # unpack: frozen rows restored, and written through into the caller's vector like the reference does
    v = g["v_in"].copy()
    pts3d_u, cam_params_u = p.get_vars_ready_for_fun(v)
    assert np.allclose(v, g["v_after"], rtol=1e-14, atol=0)
    assert np.allclose(pts3d_u, g["pts3d_u"], rtol=1e-15, atol=0) and np.allclose(cam_params_u, g["cam_params_u"], rtol=1e-14)
    corrected_pts3d, corrected_cameras = p.reconstruct_vars(g["v_in"].copy(), g["pts3d"], cams)
    assert np.allclose(corrected_pts3d, g["corrected_pts3d"], rtol=1e-7)  # float32 container
    assert np.allclose(np.array(corrected_cameras), g["corrected_cameras"], rtol=1e-12, atol=1e-9)
    assert np.allclose(np.array(p.cameras_ba), g["cameras_ba"], rtol=1e-12, atol=1e-9)
    assert np.allclose(np.array([e["R"] for e in p.estimated_params]), g["est_R"], rtol=1e-14)
    assert np.allclose(np.array([e["T"] for e in p.estimated_params]), g["est_T"], rtol=1e-14)


def test_camera_packing_matches_reference():
    g = cases.golden("cam_params")
    assert np.allclose(ba_params.load_cam_params_from_camera(g["P_aff"], None, "affine"), g["cp_aff"], rtol=1e-13)
    assert np.allclose(ba_params.load_cam_params_from_camera(g["P_persp"], None, "perspective"), g["cp_persp"], rtol=1e-9)
    assert np.array_equal(ba_params.load_cam_params_from_camera(None, np.array([1.0, 2.0, 3.0]), "rpc"), g["cp_rpc"])
    assert np.allclose(ba_params.load_camera_from_cam_params(g["cp_aff"], "affine"), g["P_aff_back"], rtol=1e-13)
    assert np.allclose(ba_params.load_camera_from_cam_params(g["cp_persp"], "perspective"), g["P_persp_back"], rtol=1e-12)
    assert np.array_equal(ba_params.load_camera_from_cam_params(g["cp_rpc"], "rpc"), g["rpc_back"])
    with pytest.raises(ba_params.Error):
        synth.make_params(synth.make_affine_scene(3, 10, 2), {"correction_params": ["R", "T", "K"]})


def test_reprojection_error_helpers_match_reference():
    g = cases.golden("reproj_err")
    assert np.allclose(ba_core.compute_reprojection_error(g["r"], g["w"]), g["err_w"], rtol=1e-15)
    assert np.allclose(ba_core.compute_reprojection_error(g["r"]), g["err"], rtol=1e-15)
    assert np.allclose(O.reprojection_error(g["r"], g["w"]), g["err_w"], rtol=1e-15)
    te = ba_core.compute_mean_reprojection_error_per_track(np.array([1.0, 2.0, 3.0, 4.0, 5.0]), np.array([0, 0, 1, 2, 2]),
                                                           np.array([0, 1, 1, 0, 2]))
    assert te.dtype == np.float32 and np.array_equal(te, g["track_err"])


def test_roundtrips_of_the_reference_unit_tests():
    """Counterparts of ref:tests/test_functions.py:19-63 on this package's own helpers (same literal matrices)."""
    g = cases.golden("cam_params")
    K, R, _, oC = cam_utils.decompose_perspective_camera(g["P_persp"])
    assert np.allclose(g["P_persp"], cam_utils.compose_perspective_camera(K, R, oC))
    assert np.allclose(g["P_aff"], cam_utils.compose_affine_camera(*cam_utils.decompose_affine_camera(g["P_aff"])))
    R = np.array([[0.25538431, -0.96424759, -0.07074919], [0.86330366, 0.19447877, 0.46570891],
                  [-0.43529948, -0.18001279, 0.8821053]])
    e = ba_rotate.euler_angles_from_R(R)
    assert np.allclose(R, ba_rotate.euler_angles_to_R(*e))
    assert np.allclose(e, ba_rotate.quaternion_to_euler(*ba_rotate.euler_to_quaternion(*e)))
    assert np.allclose(R, ba_rotate.quaternion_to_R(*ba_rotate.R_to_quaternion(R)))


def test_host_projection_helpers_match_oracle():
    for name in ("affine_RT", "persp_RT", "rpc_RT"):
        _, p, g = cases.fun_case(name)
        pts3d, cam_params = p.get_vars_ready_for_fun(g["v"][1].copy())
        if p.cam_model == "affine":
            a = ba_core.project_affine(pts3d, cam_params, p.pts_ind, p.cam_ind)
        elif p.cam_model == "perspective":
            a = ba_core.project_perspective(pts3d, cam_params, p.pts_ind, p.cam_ind)
        else:
            a = ba_core.project_rpc(pts3d, p.cameras, cam_params, p.pts_ind, p.cam_ind)
            assert a.dtype == np.float32
        assert np.abs(a - O.project(g["v"][1], p)).max() < 1e-9
    lat, lon, alt = geo_utils.ecef_to_latlon_custom(*geo_utils.latlon_to_ecef_custom(11.0, -72.7, 3500.0))
    assert abs(lat - 11.0) < 1e-9 and abs(lon + 72.7) < 1e-9 and abs(alt - 3500.0) < 1e-6


# ---------------------------------------------------------------------------------------------- initial triangulation
@pytest.mark.parametrize("name", list(cases.TRI_CASES))
def test_triangulation_oracle_reproduces_reference(name):
    """oracle/triangulate_oracle.py vs the vectors of the imported reference function (tools/gen_golden.py golden_init_pts3d):
    the running float32 mean over the pairs in list order is bit-exact; so is the RPC chain (the fixture ran the reference's C)."""
    from oracle import triangulate_oracle as T

    scene, C, pairs, g = cases.tri_case(name)
    keep = np.arange(C.shape[1]) != 5  # the fixture blanked track 5 except for camera 0
    assert np.array_equal(C[:, keep], scene.to_dense_C()[:, keep], equal_nan=True)  # the stored inputs are the seeded scene
    pts = T.init_pts3d(C, scene.cameras, scene.cam_model, pairs)
    assert pts.dtype == np.float32 and np.array_equal(pts, g["pts3d"])
    assert np.all(pts[5] == 0.0)  # seen by one camera only
    c_i, c_j = pairs[0]
    if scene.cam_model == "rpc":
        pw, err = T.rpc_triangulation(scene.cameras[c_i], scene.cameras[c_j], g["pair_obs_i"], g["pair_obs_j"])
        assert np.array_equal(err.reshape(-1, 1), g["pair_err"])
    else:
        pw = T.linear_triangulation_multiple_pts(scene.cameras[c_i], scene.cameras[c_j], g["pair_obs_i"], g["pair_obs_j"])
    assert np.array_equal(pw, g["pair_pts3d"])


def test_rpc_triangulation_oracle_against_reference_c():
    """oracle.stereo_corresp_to_lonlatalt vs the reference's own library (ref:c/disp_to_h.c:40-64, built by oracle/Makefile)."""
    from oracle import triangulate_oracle as T

    lib_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "disp_to_h.so")
    if not os.path.exists(lib_path):
        pytest.skip("oracle/_ref/disp_to_h.so not built (needs /root/reference; run `make -C oracle`)")
    lib = ctypes.CDLL(lib_path)

    class Rpc(ctypes.Structure):  # struct rpc of ref:c/rpc.h:14-32
        _fields_ = [(n, ctypes.c_double * k) for n, k in (
            ("numx", 20), ("denx", 20), ("numy", 20), ("deny", 20), ("scale", 3), ("offset", 3), ("inumx", 20),
            ("idenx", 20), ("inumy", 20), ("ideny", 20), ("iscale", 3), ("ioffset", 3), ("dmval", 4), ("imval", 4))] + [
            ("delta", ctypes.c_double)]

    def fill(r):  # ref:bundle_adjust/s2p/triangulation.py:43-79 with delta = 0.1 (:99-100)
        s = Rpc()
        s.inumx[:], s.idenx[:], s.inumy[:], s.ideny[:] = r.col_num, r.col_den, r.row_num, r.row_den
        s.numx[:] = s.denx[:] = s.numy[:] = s.deny[:] = [float("nan")] * 20
        s.ioffset[:] = [r.lon_offset, r.lat_offset, r.alt_offset]
        s.iscale[:] = [r.lon_scale, r.lat_scale, r.alt_scale]
        s.offset[:] = [r.col_offset, r.row_offset, r.alt_offset]
        s.scale[:] = [r.col_scale, r.row_scale, r.alt_scale]
        s.delta = 0.1
        return s

    r1, r2 = [RPCModel.from_file(f) for f in synth.default_rpc_files()]
    rng = np.random.default_rng(3)
    n = 500
    lon = r1.lon_offset + rng.uniform(-0.01, 0.01, n); lat = r1.lat_offset + rng.uniform(-0.005, 0.005, n)
    alt = r1.alt_offset + rng.uniform(-100, 100, n)
    p1 = np.stack(r1.projection(lon, lat, alt), 1) + rng.normal(0, 0.2, (n, 2))
    p2 = np.stack(r2.projection(lon, lat, alt), 1) + rng.normal(0, 0.2, (n, 2))
    out = np.zeros((n, 3)); err = np.zeros((n, 1), np.float32)
    a32, b32 = np.ascontiguousarray(p1.astype(np.float32)), np.ascontiguousarray(p2.astype(np.float32))
    vp = ctypes.c_void_p
    lib.stereo_corresp_to_lonlatalt(out.ctypes.data_as(vp), err.ctypes.data_as(vp), a32.ctypes.data_as(vp), b32.ctypes.data_as(vp),
                                    ctypes.c_int(n), ctypes.byref(fill(r1)), ctypes.byref(fill(r2)))
    lla, e = T.stereo_corresp_to_lonlatalt(r1, r2, p1, p2)
    # same sequence of IEEE operations; a compiler that contracts a*b+c differently would still agree to these bounds
    assert np.abs(lla[:, :2] - out[:, :2]).max() < 1e-12 and np.abs(lla[:, 2] - out[:, 2]).max() < 1e-6
    assert np.abs(e - err[:, 0]).max() < 1e-6
    assert np.abs(out[:, 2] - alt).max() < 5.0  # and the library triangulates the scene it was given


# ---------------------------------------------------------------------------------------------- RPC re-fit
@pytest.mark.parametrize("name", ["rpc0", "rpc1", "affine"])
def test_rpcfit_oracle_reproduces_reference(name):
    """oracle/rpcfit_oracle.weighted_lsq vs the vectors of the imported reference function (tools/gen_golden.py rpcfit): same numpy
    calls, so coefficients and errors agree to rounding."""
    from oracle import rpcfit_oracle as F

    g = np.load(os.path.join(cases.GOLDEN, "rpcfit.npz"))
    m, _ = F.weighted_lsq(g[name + "_target"], g[name + "_locs"])
    tab = np.concatenate([m["col_num"], m["col_den"], m["row_num"], m["row_den"],
                          [m["lon_offset"], m["lon_scale"], m["lat_offset"], m["lat_scale"], m["alt_offset"], m["alt_scale"],
                           m["col_offset"], m["col_scale"], m["row_offset"], m["row_scale"]]])
    assert np.abs(tab - g[name + "_table"]).max() <= 1e-9 * np.abs(g[name + "_table"]).max()
    assert np.abs(F.check_errors(m, g[name + "_locs"], g[name + "_target"]) - g[name + "_err"]).max() < 1e-6
    # and satba.rpc_model.RPCModel evaluates the fitted model like the reference's container did
    r = RPCModel.from_table(g[name + "_table"])
    col, row = r.projection(*g[name + "_locs"].T)
    assert np.abs(np.linalg.norm(np.stack([col, row], 1) - g[name + "_target"], axis=1) - g[name + "_err"]).max() < 1e-9


def test_rpc_text_round_trip(tmp_path):
    """RPCModel.write_to_file writes the `KEY: value unit` format of the reference's .rpc_adj outputs; from_file reads it back."""
    for k, path in enumerate(synth.default_rpc_files()):
        r = RPCModel.from_file(path)
        out = tmp_path / "{}.rpc_adj".format(k)
        r.write_to_file(str(out))
        lines = out.read_text().splitlines()
        assert len(lines) == 90 and lines[0].startswith("LINE_OFF: ") and lines[0].endswith(" pixels") and lines[10].startswith("LINE_NUM_COEFF_1: ")
        assert lines[89].startswith("SAMP_DEN_COEFF_20: ")
        back = RPCModel.from_file(str(out))
        assert np.abs(back.to_table() - r.to_table()).max() <= 1e-12 * 1.0 + 1e-12 * np.abs(r.to_table()).max()
