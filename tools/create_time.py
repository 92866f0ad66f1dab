"""Host time of satba_problem_create (index structures + upload) and device memory at the headline shape."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [os.path.join(ROOT, "sat-bundleadjust_amd"), ROOT]
import torch
from satba import sharding, synth
from satba.engine_hip import HipEngine
model, corr, n_cam, n_pts, opp = synth.CONFIGS["C4"]
scene = synth.make_scene(model, n_cam, n_pts, opp, seed=1, sigma_theta=1e-4)
p = synth.make_params(scene, {"correction_params": corr, "n_cam_fix": 1})
torch.cuda.init()
t = time.perf_counter(); eng = HipEngine(p, sharding.make_shard(p, 0, 1)); dt = time.perf_counter() - t
free, total = torch.cuda.mem_get_info()
print("problem create %.2f s, device memory in use %.2f GB" % (dt, (total - free) / 2**30))
