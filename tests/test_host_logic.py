"""Host-side logic: the scalar trust-region helpers against scipy's own, sharding, the C-ABI surface.  No GPU."""
import os
import re

import numpy as np
import pytest

import cases
from oracle import lm_oracle as L
from satba import engine_hip, sharding, synth, trf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_trust_region_helpers_match_scipy():
    from scipy.optimize._lsq import common as sc

    rng = np.random.default_rng(0)
    for _ in range(300):
        A = rng.normal(size=(2, 2))
        B = A @ A.T if rng.random() < 0.8 else A + A.T  # mostly PD, sometimes indefinite
        g = rng.normal(size=2) * 10 ** rng.uniform(-3, 3)
        Delta = 10 ** rng.uniform(-3, 2)
        p1, n1 = trf.solve_trust_region_2d(B, g, Delta)
        p2, n2 = sc.solve_trust_region_2d(B, g, Delta)
        q = lambda p: 0.5 * p @ B @ p + g @ p  # noqa: E731
        assert n1 == n2 and np.linalg.norm(p1) <= Delta * (1 + 1e-9)
        assert q(p1) <= q(p2) + 1e-9 * (abs(q(p2)) + 1e-300)
        a, b, lb, ub = rng.normal(), rng.normal(), 0.0, abs(rng.normal())
        assert np.allclose(trf.minimize_quadratic_1d(a, b, lb, ub), sc.minimize_quadratic_1d(a, b, lb, ub))
        args = (abs(rng.normal()), rng.normal(), rng.normal(), abs(rng.normal()), rng.random() < 0.5)
        assert trf.update_tr_radius(*args) == sc.update_tr_radius(*args)
        t = (rng.normal() * 1e-6, abs(rng.normal()), abs(rng.normal()) * 1e-8, abs(rng.normal()), rng.random(), 1e-4, 1e-8)
        assert trf.check_termination(*t) == sc.check_termination(*t)


@pytest.mark.parametrize("name,loss", [("affine_small_R", "linear"), ("affine_small_R", "soft_l1"), ("persp_small_R", "linear")])
def test_trf_driver_on_cpu_engine_converges_to_tight_scipy(name, loss):
    """The host loop (satba/trf.py) driven by the oracle engine: same minimiser as the reference's tight scipy run."""
    _, make_p, g, _ = cases.solve_case(name)
    p = make_p()
    eng = L.OracleEngine(p)
    res = trf.trf_solve(eng, ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=300, loss=loss)
    xt, st = g["tight_x_" + loss], g["tight_stats_" + loss]
    n_c = p.n_cam * p.n_params
    assert res.status in (2, 3, 4) and abs(res.cost - st[0]) < 1e-9 * st[0]
    assert np.abs(eng.get_x()[:n_c] - xt[:n_c]).max() < 1e-6 * np.abs(xt[:n_c]).max()


def test_trf_driver_max_nfev_and_status_codes():
    _, make_p, _, _ = cases.solve_case("affine_small_R")
    p = make_p()
    eng = L.OracleEngine(p)
    res = trf.trf_solve(eng, max_nfev=1)
    assert res.nfev == 1 and res.status == 0 and not res.success and np.array_equal(eng.get_x(), p.params_opt)
    res = trf.trf_solve(L.OracleEngine(p), ftol=1e-4, xtol=1e-10, max_nfev=300)
    assert res.status == 2 and res.success and res.nfev < 10
    res = trf.trf_solve(L.OracleEngine(p), gtol=1e30)
    assert res.status == 1 and res.nfev == 1
    bad = make_p()
    bad.params_opt[-1] = np.inf
    with pytest.raises(ValueError):
        trf.trf_solve(L.OracleEngine(bad))


def test_point_sharding_partitions_observations():
    scene = synth.make_affine_scene(7, 1000, 4, seed=2)
    p = synth.make_params(scene, {"n_pts_fix": 30})
    for world in (1, 2, 3, 8):
        shards = [sharding.make_shard(p, r, world) for r in range(world)]
        assert shards[0].p0 == 0 and shards[-1].p1 == p.n_pts and shards[0].o0 == 0 and shards[-1].o1 == p.n_obs
        for a, b in zip(shards, shards[1:]):
            assert a.p1 == b.p0 and a.o1 == b.o0
        for s in shards:
            sel = p.pts_ind[s.o0: s.o1]
            assert sel.size == 0 or (sel.min() >= s.p0 and sel.max() < s.p1)
            assert abs((s.o1 - s.o0) - p.n_obs / world) <= np.bincount(p.pts_ind).max()
        assert sum(s.n_pts_fix for s in shards) == 30 and shards[0].n_pts_fix == min(30, shards[0].n_pts)
        x = np.arange(p.params_opt.size, dtype=float)
        n_c = p.n_cam * p.n_params
        assert np.array_equal(np.concatenate([x[:n_c]] + [s.local_x(p, x)[n_c:] for s in shards]), x)


def test_c_abi_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "satba.h")).read()
    declared = set(re.findall(r"\b(satba_[a-z0-9_]+)\s*\(", header))
    assert declared == set(engine_hip.SYMBOLS), declared ^ set(engine_hip.SYMBOLS)
    lib = engine_hip.load_library()  # raises OSError if the library was not built: the build is part of the contract
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert lib.satba_version() >= 1
    assert engine_hip.HDR_FIXED == int(re.search(r"#define SATBA_HDR_FIXED (\d+)", header).group(1))
    from satba import rpc_model

    assert rpc_model.RPC_TABLE_LEN == int(re.search(r"#define SATBA_RPC_TABLE_LEN (\d+)", header).group(1))
    assert ctypes_sizeof_desc() == 12 * 4 + 2 * 8 + 6 * 8


def ctypes_sizeof_desc():
    import ctypes

    return ctypes.sizeof(engine_hip.ProblemDesc)


def test_product_path_fails_loudly_without_device_or_library(tmp_path):
    """No CPU fallback: a missing library is an OSError, a missing device a runtime error -- never a silent result."""
    with pytest.raises(OSError):
        engine_hip.load_library(str(tmp_path / "libsatba_hip.so"))
    import torch

    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    scene = synth.make_affine_scene(3, 20, 3, seed=9)
    p = synth.make_params(scene, {"correction_params": ["R"]})
    from satba import ba_core

    with pytest.raises((engine_hip.SatbaError, RuntimeError)):
        ba_core.fun(p.params_opt.copy(), p)
    with pytest.raises((engine_hip.SatbaError, RuntimeError)):
        ba_core.run_ba_optimization(p, {"verbose": 0}, False, False)
    from satba import ft_triangulate

    with pytest.raises((engine_hip.SatbaError, RuntimeError)):
        ft_triangulate.init_pts3d(scene.to_dense_C(), scene.cameras, "affine", [(0, 1)])
    with pytest.raises((engine_hip.SatbaError, RuntimeError)):
        ft_triangulate.linear_triangulation_multiple_pts(scene.cameras[0], scene.cameras[1], np.zeros((2, 2)), np.zeros((2, 2)))


def test_triangulation_arguments_are_checked_before_the_device_is_touched():
    """satba_init_pts3d validates its lists on the host (SATBA_E_ARG -> ValueError), GPU or not."""
    from satba import ft_triangulate

    scene = synth.make_affine_scene(3, 20, 3, seed=9)
    C = scene.to_dense_C()
    with pytest.raises(ValueError):
        ft_triangulate.init_pts3d(C, scene.cameras, "affine", [(1, 1)])
    with pytest.raises(ValueError):
        ft_triangulate.init_pts3d(C, scene.cameras, "affine", [(0, -2)])
    with pytest.raises(ValueError):
        ft_triangulate.init_pts3d_from_observations([0, 25], [0, 1], np.zeros((2, 2)), 20, scene.cameras, "affine", [(0, 1)])
    with pytest.raises(ValueError):
        ft_triangulate.init_pts3d_from_observations([0, 0], [0, 7], np.zeros((2, 2)), 20, scene.cameras, "affine", [(0, 1)])


# ----------------------------------------------------------------------------- figure helpers of the drop-in surface

def test_utm_series_against_published_coordinates():
    """Krueger-series UTM (satba/geo_utils.py) against coordinates any converter prints (metre level)."""
    from satba import geo_utils as G

    e, n = G.utm_from_lonlat([2.3522], [48.8566])  # Paris, zone 31
    assert G.utm_zone_from_lonlat(2.3522, 48.8566) == 31 and abs(e[0] - 452482.5) < 1.0 and abs(n[0] - 5411717.2) < 1.0
    e, n = G.utm_from_lonlat([-58.3816], [-34.6037])  # Buenos Aires, zone 21 south
    assert G.utm_zone_from_lonlat(-58.3816, -34.6037) == 21 and abs(e[0] - 373317.5) < 1.0 and abs(n[0] + 10e6 - 6170036.2) < 1.0
    assert G.epsg_code_from_utm_zone(31) == 32631 and G.epsg_code_from_utm_zone(21, north=False) == 32721
    # central meridian: easting 500 km exactly, scale factor 0.9996 on the meridian arc
    e, n = G.utm_from_lonlat([3.0, 3.0], [0.0, 1.0])
    assert abs(e[0] - 500000.0) < 1e-6 and abs(n[0]) < 1e-6 and abs(n[1] - 0.9996 * 110574.3886) < 0.01


def test_idw_interpolation_semantics():
    """ref:bundle_adjust/ba_core.py:525-567: 1/d weights over the N nearest, exact hits, N = 1."""
    from satba import ba_core

    rng = np.random.default_rng(0)
    pts, z = rng.uniform(0, 10, (40, 2)), rng.normal(size=40)
    q = rng.uniform(0, 10, (25, 2))
    got = ba_core.idw_interpolation(pts, z, q, N=5)
    d = np.linalg.norm(q[:, None] - pts[None], axis=2)
    idx = np.argsort(d, axis=1)[:, :5]
    w = 1.0 / np.take_along_axis(d, idx, 1)
    assert np.allclose(got, (w * z[idx]).sum(1) / w.sum(1), rtol=1e-12)
    assert np.array_equal(ba_core.idw_interpolation(pts, z, pts[:7], N=5), z[:7])  # a query on a known point
    assert np.array_equal(ba_core.idw_interpolation(pts, z, q, N=1), z[np.argmin(d, axis=1)])


def test_figure_helpers_write_files(tmp_path):
    """The three figure entry points a default ba_pipeline.run() calls (ref:bundle_adjust/ba_pipeline.py:654-663)."""
    import matplotlib

    matplotlib.use("Agg")
    from satba import ba_core, geo_utils, synth

    rng = np.random.default_rng(1)
    e0, e1 = rng.rayleigh(3.0, 500), rng.rayleigh(0.3, 500)
    path = tmp_path / "ba_figures" / "error_histograms.png"
    ba_core.save_histogram_of_errors(str(path), e0, e1)
    assert path.stat().st_size > 1000

    scene = synth.make_rpc_scene(3, 200, 3, seed=2)
    p = synth.make_params(scene, {"correction_params": ["R"], "reduce": False})
    lat, lon, _ = geo_utils.ecef_to_latlon_custom(*scene.pts3d.T)
    box = [[lon.min(), lat.min()], [lon.max(), lat.min()], [lon.max(), lat.max()], [lon.min(), lat.max()], [lon.min(), lat.min()]]
    half = [[lon.min(), lat.min()], [lon.mean(), lat.min()], [lon.mean(), lat.max()], [lon.min(), lat.max()], [lon.min(), lat.min()]]
    footprints = [{"type": "Polygon", "coordinates": [box]}, {"type": "Polygon", "coordinates": [half]}]
    err = rng.rayleigh(0.5, p.n_obs)
    out = tmp_path / "ba_figures" / "error_after.png"
    ba_core.save_heatmap_of_reprojection_error(str(out), p, err, footprints, aoi_lonlat_roi=footprints[1], smooth=2,
                                               global_transform=np.zeros(3))
    assert out.stat().st_size > 1000


def test_host_elbow_matches_reference_vectors():
    """satba.ba_outliers.get_elbow_value (host, one vector) on vectors run through ref:bundle_adjust/ba_outliers.py:14-58."""
    import cases
    from satba import ba_outliers

    g = cases.golden("outliers")
    for v, (elbow, ok, n) in zip(g["elbow_vecs"], g["elbow_out"]):
        v = v[: int(n)]
        if n < 2:
            continue
        e, s = ba_outliers.get_elbow_value(v)
        assert e == elbow and bool(s) == bool(ok)


def test_output_writers(tmp_path):
    """pts3d_adj.ply, cam_params/*.params and rpcs_adj/*.rpc_adj in the reference's layouts (ref:bundle_adjust/loader.py:232-238,384-406,
    ref:bundle_adjust/ba_pipeline.py:606-620)."""
    from satba import loader
    from satba.rpc_model import RPCModel

    pts = np.array([[1704018.25, -5904285.5, 1201842.125], [1.5, -2.25, 3.0]])
    ply = tmp_path / "pts3d_adj.ply"
    loader.write_point_cloud_ply(str(ply), pts)
    text = ply.read_text().splitlines()
    assert text[:7] == ["ply", "format ascii 1.0", "element vertex 2", "property float x", "property float y", "property float z", "end_header"]
    assert text[7] == "1704018.25 -5904285.5 1201842.125" and len(text) == 9
    assert np.array_equal(loader.read_point_cloud_ply(str(ply)), pts)
    loader.write_point_cloud_ply(str(ply), pts, color=(255, 0, 7))
    text = ply.read_text().splitlines()
    assert text[12] == "end_header" and text[13].endswith(" 255 0 7 255") and np.array_equal(loader.read_point_cloud_ply(str(ply)), pts)
    est = [{"R": np.array([1e-6, -2e-6, 3e-6]), "T": np.array([0.0, 0.0, 0.0]), "C": np.array([1.0, 2.0, 3.0])}]
    loader.save_estimated_params(str(tmp_path), ["img_a"], est)
    lines = (tmp_path / "cam_params" / "img_a.params").read_text().splitlines()
    assert lines[0] == "R" and lines[1] == "0.0000010000000000 -0.0000020000000000 0.0000030000000000" and lines[4] == "C" and len(lines) == 6
    r = RPCModel.from_file(synth.default_rpc_files()[0])
    fn = tmp_path / "rpcs_adj" / "img_a.rpc_adj"
    loader.save_rpcs([str(fn)], [r])
    assert np.allclose(RPCModel.from_file(str(fn)).to_table(), r.to_table(), rtol=1e-10, atol=1e-12)


def test_projection_matrix_json_writer(tmp_path):
    """P_adj/<id>_pinhole_adj.json of affine / perspective runs (ref:bundle_adjust/loader.py:255-268, ba_pipeline.py:370-378)."""
    import json

    from satba import loader

    P = np.array([[1.5, 0.25, -3.0, 1200.5], [0.0, 2.0, 0.125, -77.0], [1e-7, 2e-7, -3e-7, 4.0]])
    off = {"col0": 12, "row0": 34, "width": 5600, "height": 4100}
    fn = tmp_path / "P_adj" / "img_a_pinhole_adj.json"
    loader.save_projection_matrices([str(fn)], [P], [off])
    d = json.loads(fn.read_text())
    assert list(d.keys()) == ["P", "height", "width", "col_offset", "row_offset"]  # the reference's key order
    assert d["P"] == P.tolist() and (d["height"], d["width"], d["col_offset"], d["row_offset"]) == (4100, 5600, 12, 34)
    assert fn.read_text().startswith('{\n  "P": [\n    [\n      1.5,')  # json.dump(..., indent=2)
    P2, off2 = loader.load_projection_matrix(str(fn))
    assert np.allclose(P2, P / P[2, 3], rtol=0, atol=0) and off2 == off


def test_correction_params_the_reference_cannot_build_raise_error():
    """`correction_params` without "R" (the reference dies with AttributeError on `[].ravel()`, ref:bundle_adjust/ba_params.py:152-170)
    and with "K" (it slices the T columns a second time, :163) have no behaviour to reproduce: ba_params.Error, not a silent guess."""
    from satba import ba_params

    scene = synth.make_affine_scene(4, 30, 3, seed=1)
    for cp in ([], ["T"], ["R", "T", "K"], ["R", "T", "K", "COMMON_K"]):
        with pytest.raises(ba_params.Error):
            synth.make_params(scene, {"correction_params": cp})
    p = synth.make_params(scene, {"correction_params": ["R", "T"]})
    assert p.n_params == 5 and p.params_opt.size == 4 * 5 + 3 * 30


def test_track_filter_counts_only_pairs_written_i_lt_j():
    """
    satba.ba_outliers._tracks_with_a_listed_pair against a literal restatement of
    ref:bundle_adjust/feature_tracks/ft_utils.py:37-62: a track survives iff one of ITS camera pairs (i < j) is in
    pairs_to_triangulate -- a pair listed as (j, i) never matches.
    """
    from satba import ba_outliers

    rng = np.random.default_rng(3)
    M, N = 7, 300
    seen = rng.random((N, M)) < 0.4
    pts_ind, cam_ind = np.nonzero(seen)
    for pairs in ([(0, 1), (2, 5), (3, 6)], [(1, 0), (5, 2)], [(0, 1), (4, 2), (2, 4), (6, 9)], []):
        want = np.array([len({(a, b) for a in np.nonzero(seen[q])[0] for b in np.nonzero(seen[q])[0] if a < b} & set(pairs)) > 0
                         for q in range(N)])
        got = ba_outliers._tracks_with_a_listed_pair(pts_ind, cam_ind, N, M, pairs)
        assert np.array_equal(got, want)


def test_device_percentile_index_is_numpys():
    """
    csrc/satba_outliers.h evaluates np.percentile(v, 80) with the virtual index (n - 1) * 0.8 of numpy's "linear" method and
    numpy's two-sided lerp: the same sequence of IEEE operations restated here must reproduce np.percentile bit for bit for every
    length (including n = 1 mod 5, where 0.8 (n - 1) is an integer).
    """
    rng = np.random.default_rng(0)
    for n in range(1, 3000):
        v = np.sort(rng.random(n) * 10)
        vi = (n - 1) * 0.8
        lo = int(np.floor(vi))
        hi = min(lo + 1, n - 1)
        t, a, b = vi - lo, v[lo], v[hi]
        pct = a + (b - a) * t
        if t >= 0.5:
            pct = b - (b - a) * (1 - t)
        assert pct == np.percentile(v, 80), n


def test_asan_build_of_the_abi_shim_raises_no_report():
    """
    `make -C sat-bundleadjust_amd/csrc asan_check` builds the host side of libsatba_hip.so with -fsanitize=address and
    tests/asan/abi_driver.c, which walks the argument checks and error paths of every entry point (on a box without a GPU the calls
    that need one must return SATBA_E_HIP).  The build takes minutes, so this test runs the driver only when it has been built.
    """
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    drv = os.path.join(root, "sat-bundleadjust_amd", "satba", "lib", "asan", "abi_driver")
    if not os.path.exists(drv):
        pytest.skip("ASan build absent (make -C sat-bundleadjust_amd/csrc asan_check)")
    out = subprocess.run([drv], capture_output=True, text=True, timeout=120, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert out.returncode == 0 and "abi_driver ok" in out.stdout and "AddressSanitizer" not in out.stderr, out.stderr[-2000:]


def test_plots_of_run_ba_optimization_are_the_reference_panels(monkeypatch):
    """ref:bundle_adjust/ba_core.py:321-330 (`plots=True`, the reference's default): three panels -- residuals before / after, the two
    histograms of the reprojection error with 40 bins, the second on the range of the first."""
    import matplotlib

    matplotlib.use("Agg")
    from satba import ba_core

    rng = np.random.default_rng(0)
    r0, r1 = rng.normal(size=400), 0.1 * rng.normal(size=400)
    e0, e1 = np.abs(rng.normal(size=200)), 0.1 * np.abs(rng.normal(size=200))
    fig = ba_core.plot_residuals_and_errors(r0, r1, e0, e1, show=False)
    ax = fig.axes
    assert [a.get_title() for a in ax] == ["Residuals before and after BA", "Reprojection error before BA", "Reprojection error after BA"]
    assert len(ax[0].lines) == 2 and np.array_equal(ax[0].lines[0].get_ydata(), r0) and np.array_equal(ax[0].lines[1].get_ydata(), r1)
    assert len(ax[1].patches) == 40 and len(ax[2].patches) == 40
    assert abs(ax[2].patches[0].get_x() - e0.min()) < 1e-12 and abs(ax[2].patches[-1].get_x() + ax[2].patches[-1].get_width() - e0.max()) < 1e-12
    import matplotlib.pyplot as plt

    plt.close(fig)
