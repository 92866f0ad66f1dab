"""Tight-protocol metrics of the default path (and, with SATBA_DETERMINISTIC=1 in the environment, of the camera-major route) against
the reference vectors of tests/golden/solve_*.npz: what DESIGN.md section 6 tabulates."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sat-bundleadjust_amd"), ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import cases
from oracle import ba_oracle as O
from satba import ba_core
g3 = cases.golden("solve_tight3")
print("# columns: against the reference's 3-point run (tight3) | against its forward-difference run (tight)")
for name in cases.SOLVE_CASES:
    _, make_p, g, losses = cases.solve_case(name)
    for loss in losses:
        p = make_p()
        rpc = p.cam_model == "rpc"
        out = ba_core.run_ba_optimization(p, {"loss": loss, "ftol": 1e-15, "xtol": 1e-15, "gtol": 1e-15, "max_iter": 300, "verbose": 0,
                                              "return_result": True, "rpc_store_f32": not rpc}, False, False)
        x, err, res = out[1], out[3], out[5]
        n_c = p.n_cam * p.n_params
        cols = []
        for tag in ("3pt", "2pt"):
            if tag == "2pt":
                xt, ft, st = g["tight_x_" + loss], g["tight_fun_" + loss], g["tight_stats_" + loss]
            elif rpc:
                xt, ft, st = g["tight3_x_" + loss], g["tight3_fun_" + loss], g["tight3_stats_" + loss]
            else:
                xt, ft, st = (g3["{}_{}_{}".format(k, name, loss)] for k in ("x", "fun", "stats"))
            et = O.reprojection_error(ft, p.pts2d_w)
            cols.append("%s: cost rel %.1e cam rel %.1e res rel %.1e err max/mean %.1e mean err diff %.1e" % (
                tag, abs(res.cost - st[0]) / st[0], np.abs(x[:n_c] - xt[:n_c]).max() / np.abs(xt[:n_c]).max(),
                np.linalg.norm(res.fun - ft) / np.linalg.norm(ft), np.abs(err - et).max() / et.mean(), abs(err.mean() - et.mean())))
        print(name, loss, "status", res.status, "nfev", res.nfev, "|", " | ".join(cols), flush=True)
