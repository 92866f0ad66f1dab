"""Where the host time of one LM iteration goes: time blocked in read_header (GPU work + sync) vs Python in between."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sat-bundleadjust_amd"), ROOT]
import torch
import bench
from satba import sharding, synth, trf
from satba.engine_hip import HipEngine

model, corr, n_cam, n_pts, opp = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C4"]
scene = synth.make_scene(model, n_cam, n_pts, opp, seed=1, sigma_theta=1e-4)
p = synth.make_params(scene, {"correction_params": corr, "n_cam_fix": 1})
eng = HipEngine(p, sharding.make_shard(p, 0, 1))
eng.configure("linear", 1.0)
comm = trf.SingleComm()
st = {"first": True, "accepted": 0, "fail": 0, "cost": None}
for _ in range(3):
    bench.lm_step(eng, comm, st, trf)
torch.cuda.synchronize()
acc = {"read_header": 0.0, "n_read": 0}
orig = eng.read_header
def timed():
    t = time.perf_counter(); r = orig(); acc["read_header"] += time.perf_counter() - t; acc["n_read"] += 1; return r
eng.read_header = timed
K = 20
t0 = time.perf_counter()
for _ in range(K):
    bench.lm_step(eng, comm, st, trf)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("per iteration: total %.3f ms, blocked in read_header %.3f ms (%d reads), other host %.3f ms" % (
    1e3 * dt / K, 1e3 * acc["read_header"] / K, acc["n_read"] // K, 1e3 * (dt - acc["read_header"]) / K))
# cost of an empty read (nothing queued)
t = time.perf_counter()
for _ in range(200):
    orig()
print("empty read_header: %.1f us" % ((time.perf_counter() - t) / 200 * 1e6))
