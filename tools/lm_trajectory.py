"""Runs ON the GPU box: the LM iterations of the headline shape from x0, one line each (cost before / after, accepted, interior
Newton step, new trust radius, wall time with a synchronisation around the step).   usage: python tools/lm_trajectory.py [sigma_theta]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'sat-bundleadjust_amd'), ROOT]
import torch
from satba import sharding, synth
from satba.engine_hip import HipEngine
model, corr, n_cam, n_pts, opp = synth.CONFIGS["C4"]
scene = synth.make_scene(model, n_cam, n_pts, opp, seed=1, sigma_theta=float(sys.argv[1]) if len(sys.argv) > 1 else 1e-4)  # bench.py default
p = synth.make_params(scene, {"correction_params": corr, "n_cam_fix": 1})
eng = HipEngine(p, sharding.make_shard(p, 0, 1)); eng.configure("linear", 1.0)
st = {"first": True, "accepted": 0, "fail": 0, "cost": None}
for i in range(24):
    torch.cuda.synchronize(); t = time.perf_counter()
    r = eng.lm_step(st["first"], st.get("Delta", -1.0), 1e-14)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    st["first"] = False; st["Delta"] = r["Delta"]
    print(i, "cost %.6f -> %.6f" % (r["cost"], r["cost_new"]), "acc", r["accepted"], "newton", r["newton"], "Delta %.3g" % r["Delta"], "ms %.3f" % (dt * 1e3))
