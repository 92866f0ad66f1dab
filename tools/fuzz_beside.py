"""Random mid-size problems (129 .. 340 cameras: the sizes at which the factorisation runs BESIDE the pair kernel): the solve with the
concurrent front against the sequential one (SATBA_CHOL_BESIDE=0), which must give the same bits.  usage: fuzz_beside.py N_CASES [first_seed]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sat-bundleadjust_amd"), ROOT]
import numpy as np
from satba import sharding, synth
from satba.engine_hip import HipEngine

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
for seed in range(seed0, seed0 + n_cases):
    rng = np.random.default_rng(5000 + seed)
    model = ["affine", "perspective"][rng.integers(0, 2)]
    corr = [["R"], ["R", "T"]][rng.integers(0, 2)]
    n_p = 3 if corr == ["R"] else (5 if model == "affine" else 6)
    n_cam = int(rng.integers(129, min(340, 1024 // n_p) + 1))
    n_pts = int(rng.integers(4000, 40000))
    opp = int(rng.integers(3, 11))
    loss = ["linear", "soft_l1", "huber"][rng.integers(0, 3)]
    d = {"correction_params": corr, "n_cam_fix": int(rng.integers(0, 3)), "n_pts_fix": int(rng.integers(0, 2)) * 5, "ref_cam_weight": [1.0, 3.0][rng.integers(0, 2)]}
    scene = synth.make_scene(model, n_cam, n_pts, opp, seed=seed, sigma_theta=2e-6)
    p = synth.make_params(scene, d)
    tag = "{} M{} N{} K{} opp{} {} {} fix{}/{} w{}".format(model, p.n_cam, p.n_pts, p.n_obs, opp, "".join(corr), loss, d["n_cam_fix"], d["n_pts_fix"], d["ref_cam_weight"])
    out = []
    for beside in ("1", "0"):
        os.environ["SATBA_CHOL_BESIDE"] = beside
        e = HipEngine(p, sharding.make_shard(p, 0, 1))
        e.configure(loss, 1.0)
        e.set_x(p.params_opt.copy())
        st = e.solve_lm(ftol=1e-9, xtol=1e-12, gtol=1e-10, max_nfev=40, loss=loss, f_scale=1.0)
        info = e.info()
        out.append((e.get_x(), st.cost, int(st.nfev), int(st.status), int(info["chol_beside"]), int(info["chol_beside_timeouts"])))
    (xa, ca, na, sa, ba, ta), (xb, cb, nb, sb, bb, tb) = out
    same = np.array_equal(xa, xb) and ca == cb and na == nb and sa == sb
    ran = ba == 1 and ta == 0 and bb == 0
    bad += not (same and ran)
    print("{:3d} {:5s} {:66s} cost {:.9e} nfev {} status {} beside {}/{} timeouts {}".format(seed, "same" if same else "DIFF", tag, ca, na, sa, ba, bb, ta), flush=True)
print("cases", n_cases, "bad", bad)
sys.exit(1 if bad else 0)
