"""
Runs ON the GPU box: initial triangulation (satba_init_pts3d) at the sizes of the bench shapes, kernel time from HIP events, and the
CPU baseline beside it -- the reference's own C (oracle/_ref/disp_to_h.so, `kind: reference`) for the RPC method when the snapshot
carries it, else the numpy oracle (`kind: port`); the numpy oracle for the linear method.  One JSON line per shape.
    usage: python tools/tri_bench.py [C5|C3|C2 ...] [--resident]
"""
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sat-bundleadjust_amd"), ROOT]
from oracle import triangulate_oracle as T  # noqa: E402  (cpu_baseline leg only)
from satba import ft_triangulate as FT  # noqa: E402
from satba import synth  # noqa: E402

SHAPES = {"C2": ("affine", 10, 5000, 6), "C3": ("affine", 50, 100000, 10), "C5": ("rpc", 50, 100000, 10), "P3": ("perspective", 50, 100000, 10)}
# float64 instructions of one RPC triangulation are counted from the iteration counts the oracle reports on a sample:
# localisation = 40 (reduction) x 3 + evaluations x 50; projection = 120


def ref_lib():
    path = os.path.join(ROOT, "oracle", "_ref", "disp_to_h.so")
    return ctypes.CDLL(path) if os.path.exists(path) else None


class Rpc(ctypes.Structure):  # struct rpc of ref:c/rpc.h:14-32
    _fields_ = [(n, ctypes.c_double * k) for n, k in (
        ("numx", 20), ("denx", 20), ("numy", 20), ("deny", 20), ("scale", 3), ("offset", 3), ("inumx", 20), ("idenx", 20), ("inumy", 20),
        ("ideny", 20), ("iscale", 3), ("ioffset", 3), ("dmval", 4), ("imval", 4))] + [("delta", ctypes.c_double)]


def fill(r):
    s = Rpc()
    s.inumx[:], s.idenx[:], s.inumy[:], s.ideny[:] = r.col_num, r.col_den, r.row_num, r.row_den
    s.numx[:] = s.denx[:] = s.numy[:] = s.deny[:] = [float("nan")] * 20
    s.ioffset[:] = [r.lon_offset, r.lat_offset, r.alt_offset]; s.iscale[:] = [r.lon_scale, r.lat_scale, r.alt_scale]
    s.offset[:] = [r.col_offset, r.row_offset, r.alt_offset]; s.scale[:] = [r.col_scale, r.row_scale, r.alt_scale]
    s.delta = 0.1
    return s


def cpu_baseline(scene, C, pairs, n_target):
    """seconds per triangulation of the CPU path on a sample of about n_target correspondences"""
    lib = ref_lib() if scene.cam_model == "rpc" else None
    done, t_sum = 0, 0.0
    for c_i, c_j in pairs:
        t = np.where(~np.isnan(C[2 * c_i]) & ~np.isnan(C[2 * c_j]))[0]
        if not len(t):
            continue
        oi, oj = np.ascontiguousarray(C[2 * c_i:2 * c_i + 2, t].T), np.ascontiguousarray(C[2 * c_j:2 * c_j + 2, t].T)
        t0 = time.perf_counter()
        if lib is not None:
            out = np.zeros((len(t), 3)); err = np.zeros((len(t), 1), np.float32)
            a32, b32 = oi.astype(np.float32), oj.astype(np.float32)
            vp = ctypes.c_void_p
            lib.stereo_corresp_to_lonlatalt(out.ctypes.data_as(vp), err.ctypes.data_as(vp), a32.ctypes.data_as(vp), b32.ctypes.data_as(vp),
                                            ctypes.c_int(len(t)), ctypes.byref(fill(scene.cameras[c_i])), ctypes.byref(fill(scene.cameras[c_j])))
            T.latlon_to_ecef(out[:, 1], out[:, 0], out[:, 2])
        elif scene.cam_model == "rpc":
            T.rpc_triangulation(scene.cameras[c_i], scene.cameras[c_j], oi, oj)
        else:
            T.linear_triangulation_multiple_pts(scene.cameras[c_i], scene.cameras[c_j], oi, oj)
        t_sum += time.perf_counter() - t0
        done += len(t)
        if done >= n_target:
            break
    kind = "reference" if lib is not None else "port"
    return t_sum / max(done, 1), done, kind


for name in [a for a in sys.argv[1:] if not a.startswith("--")] or ["C5", "C3"]:
    model, M, N, opp = SHAPES[name]
    scene = synth.make_scene(model, M, N, opp, seed=1)
    ok = (lambda i, j: (i + j) % 2 == 1) if model == "rpc" else (lambda i, j: True)  # the rpc scene alternates two real models
    pairs = [(i, j) for i in range(M) for j in range(i + 1, M) if ok(i, j)]
    t0 = time.perf_counter()
    pts, info = FT.init_pts3d_from_observations(scene.pts_ind, scene.cam_ind, scene.pts2d, N, scene.cameras, model, pairs, reps=1, return_info=True)
    wall_first = time.perf_counter() - t0
    t0 = time.perf_counter()
    pts, info = FT.init_pts3d_from_observations(scene.pts_ind, scene.cam_ind, scene.pts2d, N, scene.cameras, model, pairs, reps=5, return_info=True)
    wall5 = time.perf_counter() - t0
    n_tri = int(info["n_tri"].sum())
    d = np.linalg.norm(pts.astype(np.float64) - scene.pts3d_true, axis=1)
    Cs = None
    if N <= 200000:
        Cs = scene.to_dense_C()
        spt, n_cpu, kind = cpu_baseline(scene, Cs, pairs, 200000 if model != "rpc" else (400000 if ref_lib() else 20000))
    line = {"shape": name, "cam_model": model, "n_cam": M, "n_pts": N, "n_obs": int(scene.pts_ind.size), "pairs": len(pairs),
            "triangulations": n_tri, "kernel_ms": round(info["kernel_ms"], 4), "triangulations_per_s": round(n_tri / (info["kernel_ms"] * 1e-3)),
            "call_s_with_transfers": round((wall5 - 4 * info["kernel_ms"] * 1e-3), 4), "first_call_s": round(wall_first, 3),
            "median_distance_to_truth_m": round(float(np.median(d[info["n_tri"] > 0])), 3)}
    if "--resident" in sys.argv:
        # the same on a problem handle's resident tracks (satba_init_pts3d_resident: what follows the outlier rejection), wall time
        # of the call with a one-percent mask going up and the points coming back
        p = synth.make_params(scene, {"correction_params": ["R"], "n_cam_fix": 1})
        from satba import ba_core
        ba_core.get_engine(p)  # (the handle exists after the first solve)
        rm = np.random.default_rng(1).random(p.n_obs) < 0.01
        FT.init_pts3d_resident(p, pairs, remove=rm)
        t0 = time.perf_counter()
        for _ in range(5):
            pr, ir = FT.init_pts3d_resident(p, pairs, remove=rm, return_info=True)
        line["resident_call_s"] = round((time.perf_counter() - t0) / 5, 4)
        line["resident_kernel_ms"] = round(ir["kernel_ms"], 4)
        keep = ~rm
        t0 = time.perf_counter()
        for _ in range(5):
            FT.init_pts3d_from_observations(p.pts_ind[keep], p.cam_ind[keep], p.pts2d[keep], p.n_pts, p.cameras, model, pairs)
        line["upload_call_s_same_mask"] = round((time.perf_counter() - t0) / 5, 4)
    if Cs is not None:
        line["cpu_baseline"] = {"value": round(1.0 / spt), "unit": "triangulations/s", "cores": 1, "kind": kind,
                                "sample": "%d correspondences, pairs in list order" % n_cpu,
                                "projected_s_for_this_shape": round(spt * n_tri, 2)}
    print(json.dumps(line), flush=True)
