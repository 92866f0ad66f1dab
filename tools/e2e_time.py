"""Wall clock of the drop-in call: ba_core.run_ba_optimization(p, None) at a benchmark shape, first call (engine creation, module load)
and warm (cached engine), split into its parts (ls_params["timings"]).  What a caller of the reference's API sees; the reference
times the same region at ref:bundle_adjust/ba_core.py:283-299.  One JSON line.

    python tools/e2e_time.py C4 [loss]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sat-bundleadjust_amd"), ROOT]
import numpy as np  # noqa: E402

from satba import ba_core, synth  # noqa: E402


def measure(shape, loss="linear", repeats=2):
    model, corr, n_cam, n_pts, opp = synth.CONFIGS[shape]
    scene = synth.make_scene(model, n_cam, n_pts, opp, seed=1, sigma_theta=1e-4 if model != "rpc" else 1e-6)
    p = synth.make_params(scene, {"correction_params": corr, "n_cam_fix": 1})
    out = {"shape": shape, "loss": loss, "n_obs": int(p.n_obs), "calls": []}
    r = None
    for k in range(1 + repeats):
        tm = {}
        t0 = time.perf_counter()
        r = None  # (the previous call's results go back to the system INSIDE the timed region, as when a caller rebinds the name: reported as free_s)
        tm["free_s"] = time.perf_counter() - t0
        r = ba_core.run_ba_optimization(p, {"loss": loss, "verbose": 0, "timings": tm}, False, False)
        tm["wall_s"] = time.perf_counter() - t0
        tm["host_overhead_frac"] = 1.0 - tm["solve_s"] / tm["wall_s"]
        tm["call"] = "first" if k == 0 else "warm"
        tm["err_mean_px"] = [float(np.mean(r[2])), float(np.mean(r[3]))]
        out["calls"].append(tm)
    return out


if __name__ == "__main__":
    print(json.dumps(measure(sys.argv[1] if len(sys.argv) > 1 else "C4", sys.argv[2] if len(sys.argv) > 2 else "linear")))
