"""Seeded scenes shared by the tests and tools/gen_golden.py (keep the two in sync)."""
import os

import numpy as np

from satba import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

FUN_CASES = {
    "affine_RT": ("affine", 4, 50, 3, {"correction_params": ["R", "T"], "n_cam_fix": 0}),
    "affine_R_fix": ("affine", 6, 120, 4, {"correction_params": ["R"], "n_cam_fix": 1, "n_pts_fix": 7,
                                           "ref_cam_weight": 2.5}),
    "persp_RT": ("perspective", 4, 50, 3, {"correction_params": ["R", "T"], "n_cam_fix": 1}),
    "rpc_RT": ("rpc", 4, 60, 3, {"correction_params": ["R", "T"], "n_cam_fix": 0}),
    "rpc_R": ("rpc", 3, 40, 2, {"correction_params": ["R"], "n_cam_fix": 1}),
}

SOLVE_CASES = {
    "affine_small_R": ("affine", 6, 400, 4, 2, {"correction_params": ["R"], "n_cam_fix": 1}, ["linear", "soft_l1"]),
    "affine_small_RT": ("affine", 6, 400, 4, 2, {"correction_params": ["R", "T"], "n_cam_fix": 1}, ["linear"]),
    "persp_small_R": ("perspective", 5, 300, 4, 4, {"correction_params": ["R"], "n_cam_fix": 1}, ["linear"]),
    "affine_C2_R": ("affine", 10, 5000, 6, 1, {"correction_params": ["R"], "n_cam_fix": 1}, ["linear", "soft_l1"]),
    # tight-protocol vectors only (tools/gen_golden.py tight2); rpc: the reference chain in float64 (no float32 store),
    # every camera sees every point (the two shipped RPCs alternate: same-parity-only points have no parallax)
    "rpc_small_R": ("rpc", 4, 300, 4, 5, {"correction_params": ["R"], "n_cam_fix": 1}, ["linear", "soft_l1"]),
    "persp_small_RT": ("perspective", 5, 300, 4, 4, {"correction_params": ["R", "T"], "n_cam_fix": 1}, ["linear"]),
    # round 4 (tools/gen_golden.py solves4): the correction mode bench.py times on a BASELINE shape, and a gauge-free problem (no frozen
    # camera).  Both have flat directions: only gauge-invariant outputs are compared (cost, residual vector, errors)
    "affine_C2_RT": ("affine", 10, 5000, 6, 1, {"correction_params": ["R", "T"], "n_cam_fix": 1}, ["linear"]),
    "affine_small_free": ("affine", 6, 400, 4, 2, {"correction_params": ["R"], "n_cam_fix": 0}, ["linear"]),
}
FLAT_CASES = ("affine_small_RT", "affine_C2_RT", "affine_small_free")  # parameters are not compared
SCENE_KW = {"rpc_small_R": {"sigma_theta": 5e-6}}


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def fun_case(name, dense=False):
    model, M, N, opp, d = FUN_CASES[name]
    scene = synth.make_scene(model, M, N, opp, seed=7)
    return scene, synth.make_params(scene, dict(d, reduce=False), dense=dense), golden("fun_" + name)


def solve_case(name):
    model, M, N, opp, seed, d, losses = SOLVE_CASES[name]
    scene = synth.make_scene(model, M, N, opp, seed=seed, **SCENE_KW.get(name, {}))
    return scene, (lambda: synth.make_params(scene, dict(d, reduce=False))), golden("solve_" + name), losses

# tools/gen_golden.py golden_init_pts3d: (camera model, cameras, tracks, observations per track), scenes with seed 31
TRI_CASES = {"affine": ("affine", 7, 400, 4), "persp": ("perspective", 6, 300, 3), "rpc": ("rpc", 5, 200, 3)}


def tri_case(name):
    """(scene, C, pairs, golden dict) of an init_pts3d fixture; C and the pair list are the stored ones."""
    model, M, N, opp = TRI_CASES[name]
    scene = synth.make_scene(model, M, N, opp, seed=31)
    g = np.load(os.path.join(GOLDEN, "init_pts3d.npz"))
    sub = {k[len(name) + 1:]: g[k] for k in g.files if k.startswith(name + "_")}
    return scene, sub["C"], [tuple(int(v) for v in pr) for pr in sub["pairs"]], sub


def random_case(seed):
    """A small random problem (tools/fuzz_solve.py, test_random_problems_end_where_the_oracle_ends): camera model, sizes, track
    length, correction mode, loss, frozen cameras / points and the reference camera's weight drawn from `seed`.
    Returns (tag, params, loss)."""
    rng = np.random.default_rng(1000 + seed)
    model = ["affine", "perspective", "rpc"][rng.integers(0, 3)]
    n_cam = int(rng.integers(2, 70)) if model != "rpc" else int(rng.integers(2, 12))
    n_pts = int(rng.integers(40, 2500)) if model != "rpc" else int(rng.integers(40, 600))
    opp = int(rng.integers(2, min(n_cam, 12) + 1))
    corr = [["R"], ["R", "T"]][rng.integers(0, 2)]
    loss = ["linear", "soft_l1", "huber", "cauchy"][rng.integers(0, 4)]
    d = {"correction_params": corr, "n_cam_fix": int(rng.integers(0, min(3, n_cam))), "n_pts_fix": int(rng.integers(0, 2)) * 5,
         "ref_cam_weight": [1.0, 3.0][rng.integers(0, 2)]}
    scene = synth.make_scene(model, n_cam, n_pts, opp, seed=seed, sigma_theta=2e-6 if model != "rpc" else 1e-6)
    p = synth.make_params(scene, d)
    tag = "{} M{} N{} K{} opp{} {} {} fix{}/{} w{}".format(model, p.n_cam, p.n_pts, p.n_obs, opp, "".join(corr), loss, d["n_cam_fix"],
                                                          d["n_pts_fix"], d["ref_cam_weight"])
    return tag, p, loss
