"""Tight-protocol metrics of the default path (and, with SATBA_DETERMINISTIC=1 in the environment, of the camera-major route) against
the reference vectors of tests/golden/solve_*.npz: what DESIGN.md section 6 tabulates."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sat-bundleadjust_amd"), ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import cases
from oracle import ba_oracle as O
from satba import ba_core
for name in cases.SOLVE_CASES:
    _, make_p, g, losses = cases.solve_case(name)
    for loss in losses:
        p = make_p()
        rpc = p.cam_model == "rpc"
        out = ba_core.run_ba_optimization(p, {"loss": loss, "ftol": 1e-15, "xtol": 1e-15, "gtol": 1e-15, "max_iter": 300, "verbose": 0,
                                              "return_result": True, "rpc_store_f32": not rpc}, False, False)
        x, err, res = out[1], out[3], out[5]
        key = "tight3_" if rpc else "tight_"
        xt, ft, st = g[key + "x_" + loss], g[key + "fun_" + loss], g[key + "stats_" + loss]
        n_c = p.n_cam * p.n_params
        et = O.reprojection_error(ft, p.pts2d_w)
        print(name, loss, "status", res.status, "nfev", res.nfev, "cost rel %.1e" % (abs(res.cost - st[0]) / st[0]),
              "cam rel %.1e" % (np.abs(x[:n_c] - xt[:n_c]).max() / np.abs(xt[:n_c]).max()),
              "res rel %.1e" % (np.linalg.norm(res.fun - ft) / np.linalg.norm(ft)),
              "err max/mean %.1e" % (np.abs(err - et).max() / et.mean()), "mean err diff %.1e" % abs(err.mean() - et.mean()))
