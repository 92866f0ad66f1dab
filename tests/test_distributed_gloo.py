"""
The N > 1 path on the CPU: world_size 2 over gloo.  Every rank holds all cameras and a shard of the points; the
same host loop (satba/trf.py) runs on every rank and the exchange buffer is all-reduced after each phase.  The
per-shard arithmetic is done by the CPU oracle engine here (the HIP engine needs a GPU); what is under test is the
product's sharding, the placement of the all-reduces, the rank-0-only camera terms, the header slots and the
assembly of the sharded result.
"""
import os
import socket

import numpy as np
import pytest

import cases


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, name, loss, out_dir):
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    sys.path[:0] = [os.path.join(root, "sat-bundleadjust_amd"), root, here]
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cases as cs
    from oracle import lm_oracle as L
    from satba import sharding, trf

    _, make_p, _, _ = cs.solve_case(name)
    p = make_p()
    comm = trf.TorchComm()
    shard = sharding.make_shard(p, comm.rank, comm.world)
    # a shard is itself a small problem: slice the observation lists, keep all cameras
    import copy

    q = copy.copy(p)
    q.pts_ind = p.pts_ind[shard.o0: shard.o1] - shard.p0
    q.cam_ind = p.cam_ind[shard.o0: shard.o1]
    q.pts2d = p.pts2d[shard.o0: shard.o1]
    q.pts2d_w = p.pts2d_w[shard.o0: shard.o1]
    q.pts3d = p.pts3d[shard.p0: shard.p1]
    q.n_pts, q.n_obs, q.n_pts_fix = shard.n_pts, shard.o1 - shard.o0, shard.n_pts_fix
    q.params_opt = shard.local_x(p, p.params_opt)
    eng = L.OracleEngine(q, rank=comm.rank, world=comm.world)
    eng.n_total = p.params_opt.size
    res = trf.trf_solve(eng, comm, ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=300, loss=loss)
    x = sharding.assemble_x(p, shard, eng.get_x(), comm)
    r = sharding.assemble_residuals(p, shard, eng.residuals(), comm)
    np.savez(os.path.join(out_dir, "rank{}.npz".format(rank)), x=x, r=r, cost=res.cost, nfev=res.nfev, status=res.status)
    dist.destroy_process_group()


@pytest.mark.parametrize("name,loss", [("affine_small_R", "linear"), ("affine_small_R", "soft_l1")])
def test_two_rank_solve_equals_single_rank_and_reference(tmp_path, name, loss):
    import torch.multiprocessing as mp

    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, name, loss, str(tmp_path)), nprocs=world, join=True)
    outs = [np.load(os.path.join(str(tmp_path), "rank{}.npz".format(r))) for r in range(world)]
    # every rank took the same decisions and returns the same full vectors
    assert np.array_equal(outs[0]["x"], outs[1]["x"]) and np.array_equal(outs[0]["r"], outs[1]["r"])
    assert int(outs[0]["nfev"]) == int(outs[1]["nfev"]) and int(outs[0]["status"]) == int(outs[1]["status"])

    from oracle import lm_oracle as L
    from satba import trf

    _, make_p, g, _ = cases.solve_case(name)
    p = make_p()
    eng = L.OracleEngine(p)
    res = trf.trf_solve(eng, ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=300, loss=loss)
    n_c = p.n_cam * p.n_params
    x2, x1 = outs[0]["x"], eng.get_x()
    assert abs(float(outs[0]["cost"]) - res.cost) < 1e-10 * res.cost
    assert np.abs(x2[:n_c] - x1[:n_c]).max() < 1e-7 * np.abs(x1[:n_c]).max()
    xt = g["tight_x_" + loss]
    assert np.abs(x2[:n_c] - xt[:n_c]).max() < 1e-6 * np.abs(xt[:n_c]).max()  # and the reference's tight scipy run
    assert np.linalg.norm(outs[0]["r"] - g["tight_fun_" + loss]) < 5e-6 * np.linalg.norm(g["tight_fun_" + loss])


def _ordered_worker(rank, world, port, out_dir):
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    sys.path[:0] = [os.path.join(root, "sat-bundleadjust_amd"), root, here]
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from satba import trf

    class Eng:  # the exchange buffer is all the communicator touches
        pass

    n = 4099
    out = {}
    for ordered in (False, True):
        e = Eng()
        # magnitudes spread over 30 decades: the sum depends on the order of the additions in the last bits
        rng = np.random.default_rng(100 + rank)
        e.xb = torch.from_numpy(rng.standard_normal(n) * 10.0 ** rng.uniform(-15, 15, n))
        trf.TorchComm(ordered=ordered).allreduce(e, n - 3)  # a prefix, like the phases' headers
        out["ordered" if ordered else "plain"] = e.xb.numpy().copy()
    np.savez(os.path.join(out_dir, "rank{}.npz".format(rank)), **out)
    dist.destroy_process_group()


def test_rank_ordered_reduction_is_the_sum_in_rank_order(tmp_path):
    """SATBA_ORDERED_REDUCE / TorchComm(ordered=True): every rank forms ((r0 + r1) + r2) itself -- the bits do not depend on the
    collective's algorithm; three ranks, because with two every order gives the same sum."""
    import torch.multiprocessing as mp

    world, port, n = 3, _free_port(), 4099
    mp.spawn(_ordered_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    outs = [np.load(os.path.join(str(tmp_path), "rank{}.npz".format(r))) for r in range(world)]
    parts = []
    for r in range(world):
        rng = np.random.default_rng(100 + r)
        parts.append(rng.standard_normal(n) * 10.0 ** rng.uniform(-15, 15, n))
    want = (parts[0] + parts[1]) + parts[2]
    for r in range(world):
        got = outs[r]["ordered"]
        assert np.array_equal(got[: n - 3], want[: n - 3])      # bit for bit, on every rank
        assert np.array_equal(got[n - 3:], parts[r][n - 3:])     # beyond the prefix: untouched
        assert np.allclose(outs[r]["plain"][: n - 3], want[: n - 3], rtol=1e-12, atol=0)


def _messages_worker(rank, world, port, out_dir):
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    sys.path[:0] = [os.path.join(root, "sat-bundleadjust_amd"), root, here]
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from satba import trf

    class Engine:  # what TorchComm.solve_in_messages needs of an engine: the packed payload, its message table, the three calls
        def __init__(self):
            self.xp = torch.arange(100, dtype=torch.float64) * (rank + 1)
            self.log = []

        def schur_messages(self):
            return [(0, 40), (40, 65), (65, 100)]

        def pack_schur(self):
            self.log.append(("pack", self.xp.clone()))

        def solve_messages_begin(self, packed_already=False):
            self.log.append(("begin" if not packed_already else "begin-packed", self.xp.clone()))

        def solve_messages_arrived(self, m):
            self.log.append((m, self.xp.clone()))

        def solve_messages_end(self):
            self.log.append(("end", None))

    comm, eng = trf.TorchComm(), Engine()
    comm.pipeline_min = 0  # (the product only cuts payloads of ~500 camera unknowns and more into messages)
    assert comm.solve_in_messages(eng)
    total = torch.arange(100, dtype=torch.float64) * sum(r + 1 for r in range(world))
    mine = torch.arange(100, dtype=torch.float64) * (rank + 1)
    # gloo blocks the host (and, on device tensors, synchronises the device): the payload is packed and every message summed BEFORE the
    # factorisation is launched, then the protocol runs back to back (TorchComm.solve_in_messages; RCCL: begin, then sum / arrived per message)
    assert [e[0] for e in eng.log] == ["pack", "begin-packed", 0, 1, 2, "end"]
    assert torch.equal(eng.log[0][1], mine)  # nothing reduced before the pack
    assert all(torch.equal(e[1], total) for e in eng.log[1:5])  # every message summed over exactly its range: the ranges tile the payload
    comm.pipeline = False
    assert not comm.solve_in_messages(eng)  # the caller falls back to allreduce_schur + solve
    np.save(os.path.join(out_dir, "ok{}.npy".format(rank)), np.ones(1))
    dist.destroy_process_group()


def test_schur_exchange_in_messages_order_and_sums(tmp_path):
    """TorchComm.solve_in_messages (round 6; csrc: satba_solve_messages_*) with a stub engine on two gloo ranks: the messages are summed
    over exactly their index ranges (they tile the payload), every one has landed on every rank when the engine is told so, and with a
    host-blocking backend the factorisation is only launched behind them."""
    import torch.multiprocessing as mp

    mp.spawn(_messages_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), "ok{}.npy".format(r))) for r in range(2))
