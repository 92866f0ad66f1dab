"""
Initial triangulation on the device (satba.ft_triangulate -> satba_init_pts3d / satba_triangulate_pairwise) against the oracle
and the vectors of the reference function (tests/golden/init_pts3d.npz).  SURVEY section 8f #3.

Tolerances.  The per-pair triangulations are float64 on both sides but not the same sequence of operations (fused
multiply-adds; the RPC cubics reduced to bivariate ones per height): they agree to ~1e-8 m, asserted at 1e-6 m for the linear
method and 1e-4 m for the RPC method (its two nested iterations stop on thresholds, 1e-5 m of height and 5e-6 px: two
implementations may stop one step apart).  The track means are float32 with the reference's own sequence of float32
operations, so they are bit-identical whenever the float64 points round to the same float32 -- all but the rare entry that
sits on a rounding boundary (ulp 0.125 - 0.5 m at ECEF magnitudes, the resolution the reference itself works at).  Given
identical float64 points the mean IS bit-exact: test_running_mean_is_bit_exact feeds the device's own pairwise results to
the oracle's loop.
"""
import numpy as np
import pytest

import cases
from oracle import triangulate_oracle as T
from satba import ft_triangulate as FT
from satba import synth

pytestmark = pytest.mark.gpu


def _ulp_diff(a, b):
    a = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    return np.abs(a - b)


def _assert_f32_close(got, want, max_ulp=4, max_frac=0.005):
    assert got.dtype == np.float32 and got.shape == want.shape
    d = _ulp_diff(got, want)
    assert d.max() <= max_ulp, "float32 means differ by {} ulp".format(d.max())
    assert np.mean(d > 0) <= max_frac, "{:.2%} of the float32 entries differ".format(np.mean(d > 0))


@pytest.mark.parametrize("name", list(cases.TRI_CASES))
def test_init_pts3d_against_reference_vectors(gpu, name):
    scene, C, pairs, g = cases.tri_case(name)
    pts = FT.init_pts3d(C, scene.cameras, scene.cam_model, pairs)
    _assert_f32_close(pts, g["pts3d"])
    assert np.all(pts[5] == 0.0)
    c_i, c_j = pairs[0]
    if scene.cam_model == "rpc":
        pw, err = FT.rpc_triangulation(scene.cameras[c_i], scene.cameras[c_j], g["pair_obs_i"], g["pair_obs_j"])
        assert err.shape == g["pair_err"].shape and err.dtype == np.float32
        assert np.abs(err - g["pair_err"]).max() < 1e-5
        assert np.abs(pw - g["pair_pts3d"]).max() < 1e-4
    else:
        pw = FT.linear_triangulation_multiple_pts(scene.cameras[c_i], scene.cameras[c_j], g["pair_obs_i"], g["pair_obs_j"])
        assert np.abs(pw - g["pair_pts3d"]).max() < 1e-6
    assert pw.dtype == np.float64


@pytest.mark.parametrize("model,M,N,opp", [("affine", 12, 3000, 5), ("perspective", 9, 2000, 4), ("rpc", 12, 1500, 8)])
def test_running_mean_is_bit_exact(gpu, model, M, N, opp):
    """The oracle's pair loop fed with the device's own float64 triangulations must reproduce the device's means bit for bit:
    pair order, duplicate and reversed pairs, the float32 update."""
    scene = synth.make_scene(model, M, N, opp, seed=5)
    C = scene.to_dense_C()
    rng = np.random.default_rng(11)
    ok = (lambda i, j: (i + j) % 2 == 1) if model == "rpc" else (lambda i, j: True)
    pairs = [(i, j) for i in range(M) for j in range(i + 1, M) if ok(i, j)]
    rng.shuffle(pairs)
    pairs = [tuple(int(v) for v in pr) for pr in pairs]
    pairs += [pairs[2], (pairs[4][1], pairs[4][0]), (1, M + 3)]

    def dev_pair(c_i, c_j, oi, oj):
        if model == "rpc":
            return FT.rpc_triangulation(scene.cameras[c_i], scene.cameras[c_j], oi, oj)[0]
        return FT.linear_triangulation_multiple_pts(scene.cameras[c_i], scene.cameras[c_j], oi, oj)
    want = T.init_pts3d(C, scene.cameras, model, pairs, triangulate=dev_pair)
    got, info = FT.init_pts3d_from_observations(scene.pts_ind, scene.cam_ind, scene.pts2d, N, scene.cameras, model, pairs, return_info=True)
    assert np.array_equal(got, want)
    # as many triangulations per track as the reference's loop performs
    seen = ~np.isnan(C[::2])
    n_ref = np.zeros(N, dtype=np.int64)
    for c_i, c_j in pairs:
        if c_i < M and c_j < M:
            n_ref += seen[c_i] & seen[c_j]
    assert np.array_equal(info["n_tri"], n_ref)
    assert n_ref.max() > 24  # more pairs on a track than its sorted buffer holds at a time (TRI_BUF): the refill path ran
    # and against the oracle's own float64 chain
    _assert_f32_close(got, T.init_pts3d(C, scene.cameras, model, pairs))


@pytest.mark.parametrize("model", ["affine", "rpc"])
def test_ordered_pair_lists_take_the_direct_path(gpu, model, monkeypatch):
    """A pair list in ascending order (what a pipeline's pair selection produces) is enumerated without sorting: same bits as the
    general path (SATBA_TRI_GENERAL) and as the oracle's loop fed with the device's triangulations."""
    M, N = 10, 2500
    scene = synth.make_scene(model, M, N, 6, seed=8)
    ok = (lambda i, j: (i + j) % 2 == 1) if model == "rpc" else (lambda i, j: (i * 7 + j) % 5 != 0)
    pairs = [(i, j) for i in range(M) for j in range(i + 1, M + 2) if ok(i, j)]  # includes pairs naming cameras >= M
    args = (scene.pts_ind, scene.cam_ind, scene.pts2d, N, scene.cameras, model, pairs)
    direct, info = FT.init_pts3d_from_observations(*args, return_info=True)
    monkeypatch.setenv("SATBA_TRI_GENERAL", "1")
    general, info_g = FT.init_pts3d_from_observations(*args, return_info=True)
    monkeypatch.delenv("SATBA_TRI_GENERAL")
    assert np.array_equal(direct, general) and np.array_equal(info["n_tri"], info_g["n_tri"]) and info["n_tri"].sum() > 3 * N

    def dev_pair(c_i, c_j, oi, oj):
        if model == "rpc":
            return FT.rpc_triangulation(scene.cameras[c_i], scene.cameras[c_j], oi, oj)[0]
        return FT.linear_triangulation_multiple_pts(scene.cameras[c_i], scene.cameras[c_j], oi, oj)
    assert np.array_equal(direct, T.init_pts3d(scene.to_dense_C(), scene.cameras, model, pairs, triangulate=dev_pair))


def test_pairwise_against_oracle_float64(gpu):
    for model in ("affine", "perspective", "rpc"):
        scene = synth.make_scene(model, 4, 4000, 4, seed=9)
        C = scene.to_dense_C()
        t = np.where(~np.isnan(C[0]) & ~np.isnan(C[2]))[0]
        oi, oj = C[0:2, t].T, C[2:4, t].T
        if model == "rpc":
            want, werr = T.rpc_triangulation(scene.cameras[0], scene.cameras[1], oi, oj)
            got, gerr = FT.rpc_triangulation(scene.cameras[0], scene.cameras[1], oi, oj)
            assert np.abs(gerr[:, 0] - werr).max() < 1e-5
            assert np.abs(got - want).max() < 1e-4
        else:
            want = T.linear_triangulation_multiple_pts(scene.cameras[0], scene.cameras[1], oi, oj)
            got = FT.linear_triangulation_multiple_pts(scene.cameras[0], scene.cameras[1], oi, oj)
            assert np.abs(got - want).max() < 1e-6
        assert np.isfinite(got).all() and len(t) > 500


def test_edge_cases(gpu):
    scene = synth.make_scene("affine", 4, 50, 3, seed=1)
    C = scene.to_dense_C()
    # no pairs, no tracks
    assert np.array_equal(FT.init_pts3d(C, scene.cameras, "affine", []), np.zeros((50, 3), np.float32))
    assert FT.init_pts3d(C[:, :0], scene.cameras, "affine", [(0, 1)]).shape == (0, 3)
    assert FT.linear_triangulation_multiple_pts(scene.cameras[0], scene.cameras[1], np.zeros((0, 2)), np.zeros((0, 2))).shape == (0, 3)
    # pairs that only name cameras the matrix does not have
    assert np.all(FT.init_pts3d(C, scene.cameras, "affine", [(0, 9), (7, 1)]) == 0.0)
    with pytest.raises(ValueError):
        FT.init_pts3d(C, scene.cameras, "affine", [(1, 1)])
    with pytest.raises(ValueError):
        FT.init_pts3d(C, scene.cameras, "affine", [(-1, 2)])
    # a track whose cameras form no listed pair keeps the zero initial value, the others do not
    pts = FT.init_pts3d(C, scene.cameras, "affine", [(0, 1)])
    both = ~np.isnan(C[0]) & ~np.isnan(C[2])
    assert np.all(pts[~both] == 0.0) and np.all(np.abs(pts[both]).max(axis=1) > 1e5)


@pytest.mark.parametrize("general", [False, True])
@pytest.mark.parametrize("model,M,N,opp", [("affine", 12, 3000, 5), ("perspective", 9, 2000, 4), ("rpc", 12, 1500, 8)])
def test_resident_triangulation_is_the_upload_path_bit_for_bit(gpu, model, M, N, opp, general, monkeypatch):
    """satba_init_pts3d_resident (the tracks of a problem handle, not uploaded again: what ref:bundle_adjust/ba_outliers.py:89-93 needs
    after the outlier rejection) against satba_init_pts3d on the same observations: without a mask, and with a third of the
    observations marked as removed (tracks shrink, some lose every listed pair: n_tri = 0 and a zero row, like the reference)."""
    scene = synth.make_scene(model, M, N, opp, seed=13)
    p = synth.make_params(scene, {"correction_params": ["R"], "n_cam_fix": 1})
    ok = (lambda i, j: (i + j) % 2 == 1) if model == "rpc" else (lambda i, j: (i * 5 + j) % 7 != 0)
    pairs = [(i, j) for i in range(M) for j in range(i + 1, M) if ok(i, j)]
    if general:  # out of order, with a duplicate and a reversed pair: the sorted-buffer pass
        rng = np.random.default_rng(3)
        rng.shuffle(pairs)
        pairs = [tuple(int(v) for v in pr) for pr in pairs]
        pairs += [pairs[1], (pairs[3][1], pairs[3][0])]
    up, info_u = FT.init_pts3d_from_observations(p.pts_ind, p.cam_ind, p.pts2d, p.n_pts, p.cameras, model, pairs, return_info=True)
    res, info_r = FT.init_pts3d_resident(p, pairs, return_info=True)
    assert np.array_equal(res, up) and np.array_equal(info_r["n_tri"], info_u["n_tri"]) and info_u["n_tri"].sum() > N
    remove = np.random.default_rng(5).random(p.n_obs) < 0.33
    keep = ~remove
    up, info_u = FT.init_pts3d_from_observations(p.pts_ind[keep], p.cam_ind[keep], p.pts2d[keep], p.n_pts, p.cameras, model, pairs, return_info=True)
    res, info_r = FT.init_pts3d_resident(p, pairs, remove=remove, return_info=True)
    assert np.array_equal(res, up) and np.array_equal(info_r["n_tri"], info_u["n_tri"])
    assert (info_u["n_tri"] == 0).any() and not res[info_u["n_tri"] == 0].any()


def test_rm_outliers_retriangulates_on_the_resident_tracks(gpu, monkeypatch):
    """rm_outliers on an object without a dense C: the re-triangulation of the surviving tracks through the handle's resident
    observations gives the object the upload path (SATBA_TRI_UPLOAD) gives."""
    from satba import ba_core, ba_outliers

    scene = synth.make_scene("affine", 8, 4000, 5, seed=21, noise_px=0.4)
    outl = np.random.default_rng(2).random(scene.pts2d.shape[0]) < 0.01
    scene.pts2d[outl] += 40.0
    p = synth.make_params(scene, {"correction_params": ["R"], "n_cam_fix": 1})
    err = ba_core.compute_reprojection_error(ba_core.fun(p.params_opt, p), p.pts2d_w)
    assert ba_core.cached_engine(p) is not None  # (fun built it: the resident path is the one that runs)
    new_r = ba_outliers.rm_outliers(err, p)
    monkeypatch.setenv("SATBA_TRI_UPLOAD", "1")
    new_u = ba_outliers.rm_outliers(err, p)
    assert new_r is not p and new_r.n_pts == new_u.n_pts < p.n_pts + 1 and new_r.n_obs == new_u.n_obs < p.n_obs
    assert np.array_equal(np.asarray(new_r.pts3d), np.asarray(new_u.pts3d))
    assert np.array_equal(new_r.pts_ind, new_u.pts_ind) and np.array_equal(new_r.cam_ind, new_u.cam_ind)
    assert np.array_equal(new_r.pts_prev_indices, new_u.pts_prev_indices)


def test_resident_triangulation_edge_cases(gpu):
    """No pairs at all, a pair list that names only missing cameras, every observation removed: zero rows and zero counts, no error."""
    scene = synth.make_scene("affine", 5, 300, 3, seed=2)
    p = synth.make_params(scene, {"correction_params": ["R"], "n_cam_fix": 1})
    for pairs in ([], [(7, 9), (5, 6)]):
        pts, info = FT.init_pts3d_resident(p, pairs, return_info=True)
        assert pts.shape == (p.n_pts, 3) and not pts.any() and not info["n_tri"].any()
    pts, info = FT.init_pts3d_resident(p, [(0, 1), (1, 2)], remove=np.ones(p.n_obs, dtype=bool), return_info=True)
    assert not pts.any() and not info["n_tri"].any()
    with pytest.raises(ValueError):
        FT.init_pts3d_resident(p, [(0, 1)], remove=np.zeros(p.n_obs + 1, dtype=bool))
