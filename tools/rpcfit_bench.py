"""Runs ON the GPU box: RPC re-fit of M cameras (10 x 10 x 10 grids through the shipped RPCs behind small corrective rotations):
one batched device call against the numpy oracle camera by camera.   usage: python tools/rpcfit_bench.py [M]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sat-bundleadjust_amd"), ROOT]
from oracle import rpcfit_oracle as F  # noqa: E402  (cpu_baseline leg only)
from satba import ba_rpcfit, cam_utils, geo_utils, synth  # noqa: E402
from satba.ba_core import adjust_pts3d  # noqa: E402
from satba.rpc_model import RPCModel  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(0)
base = [RPCModel.from_file(f) for f in synth.default_rpc_files()]
targets, locs = [], []
t0 = time.perf_counter()
for k in range(M):
    r = base[k % 2]
    cols, rows, alts = cam_utils.generate_point_mesh([-10, 2 * r.col_scale + 10, 10], [-10, 2 * r.row_scale + 10, 10],
                                                     [r.alt_offset - r.alt_scale, r.alt_offset + r.alt_scale, 10])
    lon, lat = r.localization(cols, rows, alts)
    X = np.stack(geo_utils.latlon_to_ecef_custom(lat, lon, alts), 1)
    c = X.mean(0)
    Rt = np.concatenate([rng.normal(0, 5e-6, 3), np.zeros(3), c + 5e5 * c / np.linalg.norm(c)]).reshape(1, 9)
    targets.append(cam_utils.apply_rpc_projection(r, adjust_pts3d(X, Rt))); locs.append(np.stack([lon, lat, alts], 1))
t_grid = time.perf_counter() - t0
targets, locs = np.stack(targets), np.stack(locs)
ba_rpcfit.weighted_lsq_batch(targets[:1], locs[:1])
t0 = time.perf_counter()
rpcs, info = ba_rpcfit.weighted_lsq_batch(targets, locs, return_info=True)
t_dev = time.perf_counter() - t0
t0 = time.perf_counter()
n_cpu = min(M, 10)
errs = []
for k in range(n_cpu):
    m, _ = F.weighted_lsq(targets[k], locs[k])
    p_ref = np.stack(F.project(m, *locs[k].T), 1); p_dev = np.stack(rpcs[k].projection(*locs[k].T), 1)
    errs.append(np.abs(p_ref - p_dev).max())
t_cpu = (time.perf_counter() - t0) / n_cpu
fit = [ba_rpcfit.check_errors(rpcs[k], locs[k], targets[k]).max() for k in range(M)]
# round 3: the whole output step of the pipeline for all cameras in one device-resident call (meshes, localisation, corrected
# projection, fit, errors, coverage test; satba_rpc_refit)
crops = [{"col0": 0, "row0": 0, "width": int(2 * base[k % 2].col_scale), "height": int(2 * base[k % 2].row_scale)} for k in range(M)]
rts = []
for k in range(M):
    r = base[k % 2]
    c = np.array(geo_utils.latlon_to_ecef_custom(r.lat_offset, r.lon_offset, r.alt_offset))
    rts.append(np.concatenate([rng.normal(0, 5e-6, 3), np.zeros(3), c + 5e5 * c / np.linalg.norm(c)]).reshape(1, 9))
ba_rpcfit.fit_Rt_corrected_rpcs(rts[:2], None, [base[0], base[1]], crops[:2])
t0 = time.perf_counter()
refit = ba_rpcfit.fit_Rt_corrected_rpcs(rts, None, [base[k % 2] for k in range(M)], crops)
t_refit = time.perf_counter() - t0
print(json.dumps({"cameras": M, "fit_Rt_corrected_rpcs_s": round(t_refit, 5), "refit_max_err_px": round(float(max(f[1].max() for f in refit)), 4),
                  "refit_margins": sorted(set(int(f[2]) for f in refit)), "samples_per_camera": int(targets.shape[1]), "device_call_s": round(t_dev, 5), "grids_and_localisation_s": round(t_grid, 4),
                  "passes": [int(info["iters"].min()), int(info["iters"].max())], "max_fit_error_px": round(float(max(fit)), 4),
                  "projection_vs_oracle_px_max": float(max(errs)),
                  "cpu_baseline": {"value": round(t_cpu, 5), "unit": "s per camera", "cores": 1, "kind": "port",
                                   "sample": "%d cameras, numpy restatement of weighted_lsq" % n_cpu, "projected_s_for_all": round(t_cpu * M, 3)}}))
