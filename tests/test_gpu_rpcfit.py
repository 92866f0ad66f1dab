"""
RPC re-fit on the device (satba.ba_rpcfit -> satba_rpc_fit / satba_rpc_localization) against the oracle and the vectors of the
reference function (tests/golden/rpcfit.npz).  SURVEY section 8f #4.

Tolerances.  The normal matrices of this fit are ill-conditioned (cond 1e13 - 1e17; for an affine camera the unregularised first
solve is rank deficient in exact arithmetic): replacing numpy.linalg.inv by another correct solver moves the reference's own
fitted projection by 5e-5 px (RPC cases) to 7e-4 px (affine case) and may change the number of re-weighted passes.  What is
asserted is therefore the fitted PROJECTION against the reference's (2e-3 px) and the fit error against the target (same level
as the reference's), not the coefficients.
"""
import numpy as np
import pytest

import cases
from oracle import rpcfit_oracle as F
from oracle import triangulate_oracle as T
from satba import ba_rpcfit, cam_utils, geo_utils, synth
from satba.rpc_model import RPCModel

pytestmark = pytest.mark.gpu


def _golden():
    import os

    return np.load(os.path.join(cases.GOLDEN, "rpcfit.npz"))


@pytest.mark.parametrize("name", ["rpc0", "rpc1", "affine"])
def test_weighted_lsq_against_reference_vectors(gpu, name):
    g = _golden()
    target, locs = g[name + "_target"], g[name + "_locs"]
    rpc = ba_rpcfit.weighted_lsq(target, locs)
    ref = RPCModel.from_table(g[name + "_table"])
    # offsets and scales are plain extrema: exact
    assert np.array_equal(rpc.to_table()[80:], ref.to_table()[80:])
    assert rpc.col_den[0] == 1.0 and rpc.row_den[0] == 1.0
    p_dev = np.stack(rpc.projection(*locs.T), 1); p_ref = np.stack(ref.projection(*locs.T), 1)
    assert np.abs(p_dev - p_ref).max() < 2e-3
    err = ba_rpcfit.check_errors(rpc, locs, target)
    assert err.max() < 1.05 * g[name + "_err"].max() + 2e-3 and np.median(err) < 1.05 * np.median(g[name + "_err"]) + 1e-3
    # against points the fit has not seen: half-way between the grid nodes, inside the box
    mid = 0.5 * (locs[:-1] + locs[1:])
    assert np.abs(np.stack(rpc.projection(*mid.T), 1) - np.stack(ref.projection(*mid.T), 1)).max() < 2e-3


def test_batch_equals_single_fits(gpu):
    g = _golden()
    t = np.stack([g["rpc0_target"], g["rpc1_target"]]); x = np.stack([g["rpc0_locs"], g["rpc1_locs"]])
    rpcs, info = ba_rpcfit.weighted_lsq_batch(t, x, return_info=True)
    for k in range(2):
        one = ba_rpcfit.weighted_lsq(t[k], x[k])
        assert np.array_equal(one.to_table(), rpcs[k].to_table())
        m, it = F.weighted_lsq(t[k], x[k])
        assert info["iters"][k] == it and abs(info["rmse"][k] - F.rmse_row_col(m, x[k], t[k])) < 1e-3
    with pytest.raises(ValueError):
        ba_rpcfit.weighted_lsq(t[0][:10], x[0][:10])  # fewer samples than unknowns
    flat = x[0].copy(); flat[:, 2] = 100.0  # constant altitude: zero scale, singular normal equations
    with pytest.raises(np.linalg.LinAlgError):
        ba_rpcfit.weighted_lsq(t[0], flat)
    assert ba_rpcfit.weighted_lsq_batch(t[:0], x[:0]) == []
    bad = t[0].copy(); bad[3, 0] = np.nan
    with pytest.raises(ValueError):
        ba_rpcfit.weighted_lsq(bad, x[0])


def test_localization_inverts_the_projection(gpu):
    """image points over the whole image (and a margin) at altitudes over the model's range -> lon / lat -> projection: back on the
    image points; and the same root as the oracle's restatement of the reference's C localisation (ref:c/rpc.c:372-408)."""
    rng = np.random.default_rng(5)
    for path in synth.default_rpc_files():
        r = RPCModel.from_file(path)
        n = 3000
        col = r.col_offset + rng.uniform(-1.02, 1.02, n) * r.col_scale; row = r.row_offset + rng.uniform(-1.02, 1.02, n) * r.row_scale
        alt = r.alt_offset + rng.uniform(-1, 1, n) * r.alt_scale
        lo, la = r.localization(col, row, alt)
        assert np.isfinite(lo).all() and np.isfinite(la).all()
        c2, r2 = r.projection(lo, la, alt)
        assert np.abs(c2 - col).max() < 1e-5 and np.abs(r2 - row).max() < 1e-5  # the iteration stops at 1e-9 of the image scale
        o = T._Rpc(r, 0.1)
        lo2, la2 = o.eval_rpc(col[:300], row[:300], alt[:300])
        assert np.abs(lo[:300] - lo2).max() < 1e-9 and np.abs(la[:300] - la2).max() < 1e-9  # degrees: 0.1 mm on the ground
        assert r.localization(np.zeros((0,)), np.zeros((0,)), np.zeros((0,)))[0].shape == (0,)
        g = r.localization(np.full((2, 3), r.col_offset), np.full((2, 3), r.row_offset), r.alt_offset)  # broadcasting like numpy
        assert g[0].shape == (2, 3) and abs(g[0][0, 0] - r.lon_offset) < 0.05


def test_fit_Rt_corrected_rpc_reproduces_the_corrected_projection(gpu):
    """ref:bundle_adjust/ba_rpcfit.py:270-345 end to end: grid over the image, localisation through the original RPC, corrected
    projection, fit, coverage check.  The fitted RPC must reproduce x = P(R (X - C) + C) on fresh points."""
    from satba import ba_core
    from satba.ba_core import adjust_pts3d

    r = RPCModel.from_file(synth.default_rpc_files()[0])
    crop = {"col0": 0, "row0": 0, "width": int(2 * r.col_scale), "height": int(2 * r.row_scale)}
    lon0, lat0 = r.lon_offset, r.lat_offset
    c = np.array(geo_utils.latlon_to_ecef_custom(lat0, lon0, r.alt_offset))
    C = c + 5e5 * c / np.linalg.norm(c)
    Rt = np.concatenate([[4e-6, -3e-6, 5e-6], np.zeros(3), C]).reshape(1, 9)
    rng = np.random.default_rng(1)
    # ground points under the image (the RPC's lon / lat box is far larger than the footprint)
    alt = r.alt_offset + rng.uniform(-0.5, 0.5, 500) * r.alt_scale
    lo, la = r.localization(rng.uniform(0, crop["width"], 500), rng.uniform(0, crop["height"], 500), alt)
    pts = np.stack(geo_utils.latlon_to_ecef_custom(la, lo, alt), 1)
    rpc, err, margin = ba_rpcfit.fit_Rt_corrected_rpc(Rt, None, r, crop, pts)
    assert margin in (10, 20, 40, 80, 160, 320, 640, 1280) and err.shape == (1000,) and err.max() < 0.5
    want = cam_utils.apply_rpc_projection(r, adjust_pts3d(pts, Rt))
    got = cam_utils.apply_rpc_projection(rpc, pts)
    assert np.abs(got - want).max() < 0.05  # pixels; the correction itself moves the points by tens of pixels
    assert np.abs(cam_utils.apply_rpc_projection(r, pts) - want).max() > 1.0
    # several cameras in one launch: the same models as one by one
    r1 = RPCModel.from_file(synth.default_rpc_files()[1])
    crop1 = {"col0": 0, "row0": 0, "width": int(2 * r1.col_scale), "height": int(2 * r1.row_scale)}
    # several cameras at once, device resident (satba_rpc_refit: mesh, localisation, corrected projection, fit, errors and coverage
    # test on the device): the same margins, the same mesh, and -- the fit is ill-conditioned, so its coefficients
    # answer to the last bits of the samples -- the same PROJECTION as the host-driven route camera by camera
    Rt1 = Rt * np.r_[2.0 * np.ones(3), np.ones(6)]
    many, info = ba_rpcfit.fit_Rt_corrected_rpcs([Rt, Rt1], None, [r, r1], [crop, crop1], return_info=True)
    one = ba_rpcfit.fit_Rt_corrected_rpc(Rt1, None, r1, crop1, pts)
    for (rpc_d, err_d, margin_d), (rpc_h, err_h, margin_h), rr, RT, k in ((many[0], (rpc, err, margin), r, Rt, 0), (many[1], one, r1, Rt1, 1)):
        assert margin_d == margin_h and err_d.shape == err_h.shape
        locs_d, target_d = info["input_locs"][k], info["target"][k]
        cr = crop if k == 0 else crop1
        cols, rows, alts = cam_utils.generate_point_mesh([-margin_h, cr["width"] + margin_h, 10], [-margin_h, cr["height"] + margin_h, 10],
                                                         [rr.alt_offset - rr.alt_scale, rr.alt_offset + rr.alt_scale, 10])
        assert np.abs(locs_d[:, 2] - alts).max() == 0.0  # numpy.linspace's arithmetic
        lon_h, lat_h = rr.localization(cols, rows, alts)
        assert np.abs(locs_d[:, 0] - lon_h).max() < 1e-11 and np.abs(locs_d[:, 1] - lat_h).max() < 1e-11
        X = np.stack(geo_utils.latlon_to_ecef_custom(lat_h, lon_h, alts), 1)
        assert np.abs(target_d - cam_utils.apply_rpc_projection(rr, ba_core.adjust_pts3d(X, RT))).max() < 1e-6
        assert np.abs(np.stack(rpc_d.projection(*locs_d.T), 1) - np.stack(rpc_h.projection(*locs_d.T), 1)).max() < 2e-3
        assert np.abs(err_d - err_h).max() < 2e-3 and np.abs(err_d - ba_rpcfit.check_errors(rpc_d, locs_d, target_d)).max() < 1e-9
    # the affine route (ba_rpcfit.py:201-267): an RPC copying a projection matrix that maps into a crop at (col0, row0)
    scene = synth.make_scene("affine", 2, 300, 2, seed=3)
    P = np.asarray(scene.cameras[0], dtype=np.float64).copy()
    X = scene.pts3d_true
    proj = cam_utils.apply_projection_matrix(P, X)
    P[0] -= proj[:, 0].min() * P[2]; P[1] -= proj[:, 1].min() * P[2]  # crop coordinates start at (0, 0)
    proj = cam_utils.apply_projection_matrix(P, X)
    crop = {"col0": 120, "row0": 75, "width": int(np.ceil(proj[:, 0].max())), "height": int(np.ceil(proj[:, 1].max()))}
    shift = np.array([crop["col0"], crop["row0"]], dtype=np.float64)
    la, lo, al = geo_utils.ecef_to_latlon_custom(*X.T)
    # the "original" RPC of that image, needed to localise the grid: fitted on a box around the scene
    glon, glat, galt = [v.reshape(-1) for v in np.meshgrid(np.linspace(lo.min() - 0.01, lo.max() + 0.01, 8), np.linspace(la.min() - 0.01, la.max() + 0.01, 8),
                                                           np.linspace(al.min() - 9000, al.max() + 9000, 8), indexing="ij")]
    G = np.stack(geo_utils.latlon_to_ecef_custom(glat, glon, galt), 1)
    base = ba_rpcfit.weighted_lsq(cam_utils.apply_projection_matrix(P, G) + shift, np.stack([glon, glat, galt], 1))
    rpc2, err2, margin2 = ba_rpcfit.fit_rpc_from_projection_matrix(P, None, base, crop, X, n_samples=8)
    assert err2.shape == (512,) and err2.max() < 0.05 and margin2 >= 10  # the regularised fit (h = 1e-3) leaves ~1e-2 px on this box
    assert np.abs(cam_utils.apply_rpc_projection(rpc2, X) - (proj + shift)).max() < 0.05


def test_rpc_pipeline_from_tracks_to_refitted_rpcs(gpu, tmp_path):
    """The reference's main use (tests/config1.json: cam_model "rpc") through the drop-in names, ref:bundle_adjust/ba_pipeline.py:700-728
    minus feature tracking: triangulate the tracks -> BundleAdjustmentParameters -> soft L1 -> outlier rejection (re-triangulates) ->
    L2 -> reconstruct_vars -> re-fit one RPC per camera -> write / read .rpc_adj.  The re-fitted RPCs, applied to the adjusted points
    with no correction, must explain the observations as well as the corrected cameras do."""
    from satba import ba_core, ba_outliers, ba_params, ft_triangulate, loader

    M = 4
    scene = synth.make_scene("rpc", M, 1200, 4, seed=12, sigma_theta=5e-6)
    rng = np.random.default_rng(2)
    bad = rng.random(scene.n_obs) < 0.02
    scene.pts2d[bad] += rng.normal(0, 25.0, (int(bad.sum()), 2))
    C = scene.to_dense_C()
    pairs = [(i, j) for i in range(M) for j in range(i + 1, M) if (i + j) % 2 == 1]  # the two shipped models alternate
    pts0 = ft_triangulate.init_pts3d(C, scene.cameras, "rpc", pairs)
    keep = np.abs(pts0).max(axis=1) > 0  # tracks seen only by same-model cameras have no pair
    C, pts0 = C[:, keep], pts0[keep]
    d = {"n_cam_fix": 0, "n_pts_fix": 0, "ref_cam_weight": 1.0, "correction_params": ["R"], "verbose": False}
    p = ba_params.BundleAdjustmentParameters(C, pts0, scene.cameras, "rpc", pairs, scene.camera_centers, d)
    _, sol, e0, e1, _ = ba_core.run_ba_optimization(p, {"loss": "soft_l1", "f_scale": 1.0, "max_iter": 300, "verbose": 0}, False, False)
    p.reconstruct_vars(sol, pts0, scene.cameras)
    n_before = p.n_obs
    p = ba_outliers.rm_outliers(e1, p, verbose=False)
    assert 0.3 * bad[np.isin(scene.pts_ind, np.nonzero(keep)[0])].sum() < n_before - p.n_obs
    _, sol, e2, e3, _ = ba_core.run_ba_optimization(p, None, False, False)
    pts_ba, cams_ba = p.reconstruct_vars(sol, np.asarray(pts0, dtype=np.float64), list(scene.cameras))
    assert e3.mean() < 0.5 and e3.mean() < 0.2 * e0.mean()
    crops = [{"col0": 0, "row0": 0, "width": int(2 * r.col_scale), "height": int(2 * r.row_scale)} for r in scene.cameras]
    fits = ba_rpcfit.fit_Rt_corrected_rpcs([np.asarray(c).reshape(1, 9) for c in cams_ba], None, scene.cameras, crops)
    names = [str(tmp_path / "rpcs_adj" / "im{}.rpc_adj".format(k)) for k in range(M)]
    loader.save_rpcs(names, [f[0] for f in fits])
    loader.write_point_cloud_ply(str(tmp_path / "pts3d_adj.ply"), p.pts3d_ba)
    loader.save_estimated_params(str(tmp_path), ["im{}".format(k) for k in range(M)], p.estimated_params)
    new_rpcs = [RPCModel.from_file(fn) for fn in names]
    # the observations through the re-fitted models, no correction any more
    err_new = np.zeros(p.n_obs)
    for k in range(M):
        sel = p.cam_ind == k
        proj = cam_utils.apply_rpc_projection(new_rpcs[k], p.pts3d_ba[p.pts_ind[sel]])
        err_new[sel] = np.linalg.norm(proj - p.pts2d[sel], axis=1)
        assert fits[k][1].max() < 0.05
    assert abs(err_new.mean() - e3.mean()) < 0.02, (err_new.mean(), e3.mean())
    assert np.abs(loader.read_point_cloud_ply(str(tmp_path / "pts3d_adj.ply")) - p.pts3d_ba).max() < 1e-6
