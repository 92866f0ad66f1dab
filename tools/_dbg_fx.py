import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sat-bundleadjust_amd"), ROOT]
import numpy as np
from satba import sharding, synth, trf
from satba.engine_hip import HipEngine
model, corr, n_cam, n_pts, opp = synth.CONFIGS["C5"]
scene = synth.make_scene(model, n_cam, n_pts, opp, seed=1, sigma_theta=1e-6)
p = synth.make_params(scene, {"correction_params": corr, "n_cam_fix": 1})
eng = HipEngine(p, sharding.make_shard(p, 0, 1))
eng.configure("soft_l1", 1.0)
ls = eng.solve_lm(ftol=1e-4, xtol=1e-10, gtol=1e-8, max_nfev=300, loss="soft_l1", f_scale=1.0)
print("nfev", ls.nfev, "status", ls.status, "cost", ls.cost, "init", ls.initial_cost)
info = eng.info()
print({k: info[k] for k in ("fx_fallbacks", "cam_sums_lds", "cm_chunks", "device_loop")})
