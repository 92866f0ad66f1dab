"""
Point sharding of one bundle-adjustment problem across ranks (one process per GPU).

Each residual touches one camera and one point (ref:bundle_adjust/ba_core.py:209-215) and observations are
grouped by point (ref:bundle_adjust/ba_params.py:142-147), so contiguous point ranges, balanced by observation
count, partition the observation stream with no overlap.  Cameras are replicated; the only coupling between
shards is the camera-side sums that satba/trf.py all-reduces.
"""
import numpy as np



class Shard:
    """The slice of a problem one engine holds: all cameras, points [p0, p1) and their observations [o0, o1)."""

    def __init__(self, p, p0=0, p1=None, o0=0, o1=None, rank=0, world=1):
        self.p0, self.p1 = p0, p.n_pts if p1 is None else p1
        self.o0, self.o1 = o0, p.n_obs if o1 is None else o1
        self.rank, self.world = rank, world
        self.n_pts = self.p1 - self.p0
        self.n_pts_fix = int(min(max(int(p.n_pts_fix) - self.p0, 0), self.n_pts))

    def local_x(self, p, v):
        """[all camera variables | this shard's points] of a global variable vector."""
        n_c = p.n_cam * p.n_params
        if self.p0 == 0 and n_c + 3 * self.p1 == v.size:
            return v  # (one rank: the whole vector, not a copy of it -- 24 MB at 1 M points)
        return np.concatenate((v[:n_c], v[n_c + 3 * self.p0: n_c + 3 * self.p1]))


def point_offsets(p):
    """CSR offsets of the point-major observation list: observations of point i are [ofs[i], ofs[i+1])."""
    return np.searchsorted(p.pts_ind, np.arange(p.n_pts + 1), side="left")


def split_points(p, world):
    """
    Boundaries (world + 1 point indices) of contiguous point ranges whose observation counts are as equal as
    whole points allow.
    """
    ofs = point_offsets(p)
    targets = p.n_obs * np.arange(1, world) / world
    cuts = np.searchsorted(ofs, targets, side="left")
    bounds = np.concatenate(([0], cuts, [p.n_pts])).astype(np.int64)
    return np.maximum.accumulate(bounds), ofs


def make_shard(p, rank, world):
    if world == 1:
        return Shard(p)
    bounds, ofs = split_points(p, world)
    p0, p1 = int(bounds[rank]), int(bounds[rank + 1])
    return Shard(p, p0, p1, int(ofs[p0]), int(ofs[p1]), rank, world)


def assemble_x(p, shard, x_local, comm):
    """
    Global variable vector from per-shard solutions: cameras are identical on all ranks, points are disjoint contiguous ranges, so the
    point parts are all-gathered (each rank contributes its 3 (p1 - p0) doubles; round 2 all-reduced a zero-padded full-size array).
    """
    n_c = p.n_cam * p.n_params
    if comm.world == 1:
        return x_local
    return np.concatenate((x_local[:n_c], comm.gather_array(x_local[n_c:])))


def assemble_residuals(p, shard, r_local, comm):
    """Residual vector in the caller's observation order: the shards hold contiguous observation ranges in rank order."""
    if comm.world == 1:
        return r_local
    return comm.gather_array(r_local)
