"""
ctypes binding of libsatba_hip.so (include/satba.h) and the device engine the solver loop drives.

There is no CPU fallback: if the shared library is missing, or no HIP device is usable, constructing a
HipEngine raises.  PyTorch is used only as plumbing -- the exchange buffer is a torch CUDA tensor so that
torch.distributed (RCCL) can all-reduce it in place, and kernels are launched on torch's current stream so
they order naturally with those collectives.  With no torch CUDA runtime the library's own buffer and the
default stream are used (single GPU only).
"""
import ctypes as C
import os

import numpy as np

from .rpc_model import rpc_to_table
from .sharding import Shard

_LIB = None
LIB_PATH = os.environ.get("SATBA_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libsatba_hip.so")

CAM_MODELS = {"affine": 0, "perspective": 1, "rpc": 2}
LOSSES = {"linear": 0, "soft_l1": 1, "huber": 2, "cauchy": 3, "arctan": 4}
HDR_FIXED = 16

# every symbol include/satba.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "satba_last_error", "satba_version", "satba_problem_create", "satba_problem_destroy", "satba_set_stream",
    "satba_exchange_len", "satba_header_len", "satba_bind_exchange", "satba_configure", "satba_set_x", "satba_get_x",
    "satba_residuals", "satba_reprojection_errors", "satba_reprojection_errors_begin", "satba_reprojection_errors_fetch", "satba_linearize", "satba_prepare", "satba_schur", "satba_schur_auto", "satba_solve", "satba_subspace", "satba_subspace_products", "satba_trial", "satba_trial_gn",
    "satba_accept", "satba_camera_sums_fallback", "satba_read_header", "satba_get_blocks", "satba_get_jacobian", "satba_get_exchange",
    "satba_set_exchange", "satba_get_vector", "satba_time_kernel",
    "satba_packed_schur_len", "satba_pack_schur", "satba_solve_messages", "satba_solve_messages_bind", "satba_solve_messages_begin", "satba_solve_messages_arrived", "satba_solve_messages_end", "satba_unpack_schur",
    "satba_solve_lm", "satba_lm_step", "satba_lm_run", "satba_lm_state", "satba_lm_begin", "satba_lm_part", "satba_lm_poll", "satba_profile_linearize", "satba_profile_read", "satba_outliers", "satba_layout_len", "satba_get_layout", "satba_get_info",
    "satba_triangulate_pairwise", "satba_init_pts3d", "satba_init_pts3d_resident", "satba_snapshot_x",
    "satba_rpc_fit", "satba_rpc_localization", "satba_rpc_refit",
]

FLAG_DETERMINISTIC = 1
LAYOUT = {"perm": 0, "rank": 1, "pt_cnt": 2, "slice_base": 3, "e_cam": 4, "obs_pos": 5, "cam_ofs": 6, "cm_pt": 7, "cm_pos": 8,
          "pair_ofs": 9, "pair_pts": 10, "pair_pi": 11, "pair_pj": 12, "pair_ij": 13, "cm_io": 14, "ipt_ofs": 15,
          # the merged records of the weighted / robust runs (exist after the first such linearisation)
          "w_fix": 16, "sc_ofs": 17, "pair_rec": 18, "pair_kk": 19, "cm_rec": 20, "cm_sc": 21, "dg_ofs": 22}


class LmOpts(C.Structure):
    _fields_ = [("ftol", C.c_double), ("xtol", C.c_double), ("gtol", C.c_double), ("f_scale", C.c_double),
                ("max_nfev", C.c_int64), ("loss", C.c_int32), ("verbose", C.c_int32)]


class LmStats(C.Structure):
    _fields_ = [("cost", C.c_double), ("initial_cost", C.c_double), ("optimality", C.c_double),
                ("nfev", C.c_int64), ("njev", C.c_int64), ("iterations", C.c_int64), ("status", C.c_int32), ("reserved", C.c_int32)]

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class ProblemDesc(C.Structure):
    _fields_ = [
        ("cam_model", C.c_int32), ("n_cam", C.c_int32), ("n_pts", C.c_int32), ("n_params", C.c_int32),
        ("cam_param_len", C.c_int32), ("n_cam_fix", C.c_int32), ("n_pts_fix", C.c_int32), ("rank", C.c_int32),
        ("world", C.c_int32), ("rpc_store_f32", C.c_int32), ("device", C.c_int32), ("flags", C.c_int32),
        ("n_obs", C.c_int64), ("n_total", C.c_int64),
        ("cam_params", _dp), ("rpc_tables", _dp), ("cam_ind", _ip), ("pts_ind", _ip), ("pts2d", _dp), ("weights", _dp),
    ]


def load_library(path=None):
    """dlopen libsatba_hip.so and declare the prototypes.  Raises OSError if it has not been built."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise OSError("libsatba_hip.so not found at {}: build it with `make -C sat-bundleadjust_amd/csrc` "
                      "(or __graft_entry__.build()); there is no CPU fallback".format(path))
    try:
        # torch bundles its own libamdhip64 under the same soname: when torch is going to be used in this process
        # it must be loaded first so that this library binds to the same HIP runtime (two runtimes in one
        # process do not both see the device)
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    h = C.c_void_p
    lib.satba_last_error.restype = C.c_char_p
    lib.satba_version.restype = C.c_int
    lib.satba_problem_create.argtypes = [C.POINTER(ProblemDesc), C.POINTER(h)]
    lib.satba_problem_destroy.argtypes = [h]
    lib.satba_problem_destroy.restype = None
    lib.satba_set_stream.argtypes = [h, C.c_void_p, C.c_int32]
    lib.satba_exchange_len.argtypes = [h]
    lib.satba_exchange_len.restype = C.c_int64
    lib.satba_header_len.argtypes = [h]
    lib.satba_header_len.restype = C.c_int64
    lib.satba_bind_exchange.argtypes = [h, C.c_void_p, C.c_int64]
    lib.satba_configure.argtypes = [h, C.c_int32, C.c_double]
    lib.satba_set_x.argtypes = [h, _dp]
    lib.satba_get_x.argtypes = [h, _dp]
    lib.satba_residuals.argtypes = [h, _dp, _dp]
    lib.satba_reprojection_errors.argtypes = [h, _dp, _dp]
    lib.satba_reprojection_errors_begin.argtypes = [h]
    lib.satba_reprojection_errors_fetch.argtypes = [h, _dp]
    for name in ("satba_linearize", "satba_solve", "satba_accept", "satba_subspace_products", "satba_camera_sums_fallback"):
        getattr(lib, name).argtypes = [h]
    lib.satba_prepare.argtypes = [h, C.c_int32]
    lib.satba_schur.argtypes = [h, C.c_double]
    lib.satba_schur_auto.argtypes = [h, C.c_double, C.c_double]
    lib.satba_subspace.argtypes = [h, C.c_double, C.c_double]
    lib.satba_trial.argtypes = [h, C.c_double, C.c_double]
    lib.satba_trial_gn.argtypes = [h, C.c_double, C.c_double]
    lib.satba_read_header.argtypes = [h, _dp]
    lib.satba_get_blocks.argtypes = [h, _dp, _dp, _dp, _dp]
    lib.satba_get_jacobian.argtypes = [h, _dp, _dp]
    lib.satba_get_exchange.argtypes = [h, C.c_int64, C.c_int64, _dp]
    lib.satba_set_exchange.argtypes = [h, C.c_int64, C.c_int64, _dp]
    lib.satba_packed_schur_len.argtypes = [h]
    lib.satba_packed_schur_len.restype = C.c_int64
    lib.satba_solve_messages.argtypes = [h, C.POINTER(C.c_int64), C.c_int32]
    lib.satba_solve_messages.restype = C.c_int32
    lib.satba_solve_messages_bind.argtypes = [h, C.c_void_p]
    lib.satba_solve_messages_begin.argtypes = [h, C.c_void_p, C.c_int32]
    lib.satba_solve_messages_arrived.argtypes = [h, C.c_void_p, C.c_int32]
    lib.satba_solve_messages_end.argtypes = [h]
    lib.satba_pack_schur.argtypes = [h, C.c_void_p]
    lib.satba_unpack_schur.argtypes = [h, C.c_void_p]
    lib.satba_get_vector.argtypes = [h, C.c_int32, _dp]
    lib.satba_time_kernel.argtypes = [h, C.c_int32, C.c_int32, C.POINTER(C.c_float)]
    lib.satba_solve_lm.argtypes = [h, C.POINTER(LmOpts), C.POINTER(LmStats)]
    lib.satba_lm_step.argtypes = [h, C.c_int32, C.c_double, C.c_double, C.POINTER(C.c_double)]
    lib.satba_lm_run.argtypes = [h, C.c_int64, C.c_int32, C.c_double, C.POINTER(C.c_double), C.c_int32]
    lib.satba_lm_begin.argtypes = [h, C.POINTER(LmOpts), C.c_int32, C.c_int64, C.c_int32]
    lib.satba_lm_part.argtypes = [h, C.c_int32, C.c_double]
    lib.satba_lm_poll.argtypes = [h, C.POINTER(C.c_int64), C.c_int32]
    lib.satba_lm_state.argtypes = [h, C.POINTER(C.c_double), C.c_int32]
    lib.satba_profile_linearize.argtypes = [h, C.c_int32]
    lib.satba_profile_read.argtypes = [h, C.POINTER(C.c_int64), C.POINTER(C.c_double)]
    lib.satba_outliers.argtypes = [h, _dp, C.c_double, C.c_double, _dp, C.POINTER(C.c_uint8), C.POINTER(C.c_int64)]
    lib.satba_layout_len.argtypes = [h, C.c_int32]
    lib.satba_layout_len.restype = C.c_int64
    lib.satba_get_layout.argtypes = [h, C.c_int32, C.c_int64, C.c_void_p]
    lib.satba_get_info.argtypes = [h, _dp, C.c_int32]
    lib.satba_snapshot_x.argtypes = [h, C.c_int32]
    lib.satba_rpc_fit.argtypes = [C.c_int32, C.c_int32, _dp, _dp, C.c_double, C.c_double, C.c_int32, _dp, _dp, _ip, C.c_int32]
    lib.satba_rpc_localization.argtypes = [_dp, C.c_int64, _dp, _dp, _dp, _dp, _dp, C.c_int32]
    lib.satba_rpc_refit.argtypes = [C.c_int32, _dp, _dp, _dp, _dp, _dp, C.c_int32, C.c_double, C.c_double, C.c_int32, _dp, _dp, _dp, _dp, _dp, C.c_int32]
    _fp = C.POINTER(C.c_float)
    lib.satba_triangulate_pairwise.argtypes = [C.c_int32, _dp, _dp, C.c_int64, _dp, _dp, _dp, _fp, C.c_int32, _fp]
    lib.satba_init_pts3d.argtypes = [C.c_int32, C.c_int32, C.c_int64, C.POINTER(C.c_int64), _ip, _dp, _dp, C.c_int32, _ip, _fp, _ip,
                                     C.c_int32, C.c_int32, _fp]
    lib.satba_init_pts3d_resident.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), _dp, C.c_int32, _ip, _fp, _ip, _fp]
    if path == LIB_PATH:
        _LIB = lib
    return lib


def _ptr(a, typ=_dp):
    return a.ctypes.data_as(typ)


class SatbaError(RuntimeError):
    pass


def _check(lib, rc):
    if rc == 0:
        return
    msg = (lib.satba_last_error() or b"").decode()
    if rc in (-1, -4):
        raise ValueError(msg)
    raise SatbaError("libsatba_hip: {} (code {})".format(msg, rc))


class HipEngine:
    """Device-resident solver state of one shard (see satba/trf.py for the phase contract)."""

    HDR_FIXED = HDR_FIXED

    def __init__(self, p, shard=None, device=None, rpc_f32=True, use_torch=None, deterministic=False):
        self.lib = load_library()
        self.p = p
        self.shard = shard or Shard(p)
        sh = self.shard
        self.rank, self.world = sh.rank, sh.world
        self.n_cam, self.n_p, self.n_pts = p.n_cam, p.n_params, sh.n_pts
        self.n_c = self.n_cam * self.n_p
        self.n = self.n_c + 3 * self.n_pts
        self.n_total = p.n_cam * p.n_params + 3 * p.n_pts
        self.n_obs = sh.o1 - sh.o0

        self._torch = None
        if use_torch is None or use_torch:
            try:
                import torch

                if torch.cuda.is_available():
                    self._torch = torch
            except ImportError:
                pass
            if use_torch and self._torch is None:
                raise SatbaError("torch with a usable HIP device was requested but is not available")
        if device is None:
            device = self._torch.cuda.current_device() if self._torch else 0
        self.device = int(device)

        cam_params = np.ascontiguousarray(p.cam_params, dtype=np.float64)
        cam_ind = np.ascontiguousarray(p.cam_ind[sh.o0: sh.o1], dtype=np.int32)
        pts_ind = np.ascontiguousarray(p.pts_ind[sh.o0: sh.o1] - sh.p0, dtype=np.int32)
        pts2d = np.ascontiguousarray(p.pts2d[sh.o0: sh.o1], dtype=np.float64)
        w = np.ascontiguousarray(p.pts2d_w[sh.o0: sh.o1], dtype=np.float64)
        rpc = None
        if p.cam_model == "rpc":
            rpc = np.ascontiguousarray(np.stack([rpc_to_table(c) for c in p.cameras]), dtype=np.float64)
        d = ProblemDesc(
            cam_model=CAM_MODELS[p.cam_model], n_cam=p.n_cam, n_pts=sh.n_pts, n_params=p.n_params,
            cam_param_len=cam_params.shape[1], n_cam_fix=int(p.n_cam_fix), n_pts_fix=sh.n_pts_fix, rank=sh.rank,
            world=sh.world, rpc_store_f32=int(bool(rpc_f32)), device=self.device,
            flags=FLAG_DETERMINISTIC if deterministic else 0, n_obs=self.n_obs,
            n_total=self.n_total, cam_params=_ptr(cam_params), rpc_tables=_ptr(rpc) if rpc is not None else None,
            cam_ind=_ptr(cam_ind, _ip), pts_ind=_ptr(pts_ind, _ip), pts2d=_ptr(pts2d), weights=_ptr(w))
        self._h = C.c_void_p()
        _check(self.lib, self.lib.satba_problem_create(C.byref(d), C.byref(self._h)))

        self.hdr = int(self.lib.satba_header_len(self._h))
        self.xb_len = int(self.lib.satba_exchange_len(self._h))
        self.len_lin = self.hdr + self.n_cam * self.n_p ** 2 + self.n_c
        self.len_schur = self.hdr + self.n_c ** 2 + self.n_c
        self.xb = None
        if self._torch is not None:
            torch = self._torch
            self.xb = torch.zeros(self.xb_len, dtype=torch.float64, device="cuda:{}".format(self.device))
            torch.cuda.synchronize(self.device)  # the fill ran on torch's stream; the handle launches on its own
            _check(self.lib, self.lib.satba_bind_exchange(self._h, C.c_void_p(self.xb.data_ptr()), self.xb_len))
            if self.world > 1:
                # several ranks: the kernels must order with the collectives torch.distributed queues on torch's current stream
                self.use_stream(torch.cuda.current_stream(self.device))
        self._hdr_host = np.zeros(self.hdr)
        self._poll = np.zeros(5, dtype=np.int64)
        self.xp = None  # packed Schur payload (header | rhs | lower triangle of S), allocated on first use
        self._messages = None  # schur_messages()
        self.len_schur_packed = int(self.lib.satba_packed_schur_len(self._h))
        self.set_x(sh.local_x(p, np.asarray(p.params_opt, dtype=np.float64)))

    # -- lifetime
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.satba_problem_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- packed exchange of the Schur payload (multi-rank runs): S is symmetric, only its lower triangle is formed
    def pack_schur(self):
        """Copy [header | lower triangle of S | rhs] of the exchange buffer into the tensor `xp` and return it."""
        if self._torch is None:
            raise SatbaError("the packed exchange needs the torch exchange buffer")
        if self.xp is None:
            self.xp = self._torch.zeros(self.len_schur_packed, dtype=self._torch.float64, device="cuda:{}".format(self.device))
        _check(self.lib, self.lib.satba_pack_schur(self._h, C.c_void_p(self.xp.data_ptr())))
        return self.xp

    def unpack_schur(self):
        """Copy the (all-reduced) packed payload back into the exchange buffer."""
        _check(self.lib, self.lib.satba_unpack_schur(self._h, C.c_void_p(self.xp.data_ptr())))

    # -- the Schur exchange in messages, the factorisation beside it (satba_solve_messages_*; several ranks)
    def schur_messages(self):
        """Index ranges [(begin, end), ...] of the messages the packed payload `xp` is all-reduced in; [] when this engine solves its
        reduced system in one piece (pack_schur / all-reduce / unpack_schur / solve)."""
        if self._torch is None:
            return []
        if self._messages is None:
            b = (C.c_int64 * 18)()
            nm = int(self.lib.satba_solve_messages(self._h, b, 18))
            if nm < 0:
                raise SatbaError("satba_solve_messages failed")
            self._messages = [(int(b[m]), int(b[m + 1])) for m in range(nm)]
            if nm:
                if self.xp is None:
                    self.xp = self._torch.zeros(self.len_schur_packed, dtype=self._torch.float64, device="cuda:{}".format(self.device))
                _check(self.lib, self.lib.satba_solve_messages_bind(self._h, C.c_void_p(self.xp.data_ptr())))
        return self._messages

    def solve_messages_begin(self, packed_already=False):
        _check(self.lib, self.lib.satba_solve_messages_begin(self._h, C.c_void_p(self.xp.data_ptr()), 1 if packed_already else 0))

    def solve_messages_arrived(self, m):
        _check(self.lib, self.lib.satba_solve_messages_arrived(self._h, C.c_void_p(self.xp.data_ptr()), int(m)))

    def solve_messages_end(self):
        _check(self.lib, self.lib.satba_solve_messages_end(self._h))

    def use_torch_stream(self):
        """Launch on torch's current stream, so that the kernels order with the collectives torch.distributed queues there."""
        if self._torch is not None:
            self.use_stream(self._torch.cuda.current_stream(self.device))

    def use_stream(self, stream, own=False):
        """Launch on a torch.cuda.Stream (None: the legacy default stream); own=True: back to the handle's own stream."""
        handle = stream.cuda_stream if stream is not None else 0
        _check(self.lib, self.lib.satba_set_stream(self._h, C.c_void_p(handle), 1 if own else 0))

    # -- state transfer
    def configure(self, loss, f_scale):
        if loss not in LOSSES:
            raise ValueError("`loss` must be one of {}".format(list(LOSSES)))
        _check(self.lib, self.lib.satba_configure(self._h, LOSSES[loss], float(f_scale)))

    def set_x(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        if x.size != self.n:
            raise ValueError("x has {} entries, expected {}".format(x.size, self.n))
        _check(self.lib, self.lib.satba_set_x(self._h, _ptr(x)))

    def get_x(self):
        x = np.empty(self.n)
        _check(self.lib, self.lib.satba_get_x(self._h, _ptr(x)))
        return x

    def residuals(self, with_cost=False):
        r = np.empty(2 * self.n_obs)
        cost = C.c_double()
        _check(self.lib, self.lib.satba_residuals(self._h, _ptr(r), C.byref(cost)))
        return (r, cost.value) if with_cost else r

    def reprojection_errors(self):
        """satba_reprojection_errors: compute_reprojection_error(fun(x)) on the device, (n_obs,) float64 in the caller's order."""
        e = np.empty(self.n_obs)
        _check(self.lib, self.lib.satba_reprojection_errors(self._h, _ptr(e), None))
        return e

    def reprojection_errors_begin(self):
        """Queue the error kernels at the current x (satba_reprojection_errors_begin); reprojection_errors_fetch brings them over."""
        _check(self.lib, self.lib.satba_reprojection_errors_begin(self._h))

    def reprojection_errors_fetch(self):
        """The errors of the x at reprojection_errors_begin.  May run in another thread while this engine solves (ctypes releases the
        GIL; the transfer uses streams and pinned buffers of its own)."""
        e = np.empty(self.n_obs)
        _check(self.lib, self.lib.satba_reprojection_errors_fetch(self._h, _ptr(e)))
        return e

    def read_header(self):
        _check(self.lib, self.lib.satba_read_header(self._h, _ptr(self._hdr_host)))
        return self._hdr_host.copy()

    # -- phases
    def linearize(self):
        _check(self.lib, self.lib.satba_linearize(self._h))

    def prepare(self, first):
        _check(self.lib, self.lib.satba_prepare(self._h, int(bool(first))))

    def schur(self, lam):
        _check(self.lib, self.lib.satba_schur(self._h, float(lam)))

    def schur_auto(self, Delta, lam_floor=0.0):
        _check(self.lib, self.lib.satba_schur_auto(self._h, float(Delta), float(lam_floor)))

    def solve(self):
        _check(self.lib, self.lib.satba_solve(self._h))

    def subspace(self, alpha, inv_norm_g):
        _check(self.lib, self.lib.satba_subspace(self._h, float(alpha), float(inv_norm_g)))

    def subspace_products(self):
        _check(self.lib, self.lib.satba_subspace_products(self._h))

    def trial(self, p0, p1):
        _check(self.lib, self.lib.satba_trial(self._h, float(p0), float(p1)))

    def trial_gn(self, ca, cb):
        _check(self.lib, self.lib.satba_trial_gn(self._h, float(ca), float(cb)))

    def accept(self):
        _check(self.lib, self.lib.satba_accept(self._h))

    def camera_sums_fallback(self):
        """
        Switch the per-camera sums of the linearisation from the fixed-point LDS table to the camera-major pass (header slot
        K_FX_BAD was raised).  Returns False when the engine is already on that route.
        """
        if not self.info()["cam_sums_lds"]:
            return False
        _check(self.lib, self.lib.satba_camera_sums_fallback(self._h))
        return True

    # -- whole solve below the ABI (single rank)
    def profile_linearize(self, on=True):
        """Bracket every k_linearize launch of the following linearize calls with HIP events (satba_profile_linearize)."""
        _check(self.lib, self.lib.satba_profile_linearize(self._h, 1 if on else 0))

    def profile_read(self):
        """(number of bracketed launches, sum of their durations in ms) since the last read."""
        n, ms = C.c_int64(), C.c_double()
        _check(self.lib, self.lib.satba_profile_read(self._h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def snapshot_x(self, restore=False):
        """satba_snapshot_x: keep / return to a device-side copy of the current point."""
        _check(self.lib, self.lib.satba_snapshot_x(self._h, 1 if restore else 0))

    def lm_step(self, first, Delta, lam_floor=0.0):
        """satba_lm_step: one fixed-work LM iteration, host side in C++ (single rank).  Returns a dict of the eight scalars."""
        out = np.zeros(8)
        _check(self.lib, self.lib.satba_lm_step(self._h, 1 if first else 0, float(Delta), float(lam_floor), _ptr(out)))
        keys = ["cost", "cost_new", "Delta", "accepted", "newton", "predicted", "actual", "lam"]
        return dict(zip(keys, out))

    LM_KEYS = ["cost", "cost_new", "Delta", "accepted", "newton", "predicted", "actual", "lam", "phase", "status", "nfev", "njev",
               "iterations", "ticks", "host_reason", "g_norm", "initial_cost"]

    def lm_run(self, n_iterations, cycle_len=0, lam_floor=0.0):
        """
        satba_lm_run: n fixed-work LM iterations on the device-resident loop (decisions taken by one-thread kernels on the device, no
        host round trip inside an iteration); cycle_len > 0: back to the point kept by snapshot_x every cycle_len iterations.
        Returns the loop's scalars as a dict once the device is done.
        """
        out = np.zeros(16)
        _check(self.lib, self.lib.satba_lm_run(self._h, int(n_iterations), int(cycle_len), float(lam_floor), _ptr(out), 16))
        return dict(zip(self.LM_KEYS, out))

    def lm_begin(self, ftol=1e-8, xtol=1e-8, gtol=1e-8, max_nfev=None, loss="linear", f_scale=1.0, never_stop=False, max_iterations=0,
                 cycle_len=0):
        """satba_lm_begin: reset the device-resident loop (several ranks: the parts of a tick are queued with lm_part)."""
        o = LmOpts(ftol=ftol, xtol=xtol, gtol=gtol, f_scale=f_scale, max_nfev=-1 if max_nfev is None else int(max_nfev),
                   loss=LOSSES[loss], verbose=0)
        _check(self.lib, self.lib.satba_lm_begin(self._h, C.byref(o), 1 if never_stop else 0, int(max_iterations), int(cycle_len)))

    def lm_part(self, part, lam_floor=0.0):
        _check(self.lib, self.lib.satba_lm_part(self._h, int(part), float(lam_floor)))

    def lm_poll(self):
        """satba_lm_poll (no wait): (patterns executed, phase, pauses for the subspace pattern, tick of the latest pause, tick at which
        the loop left the running phase or 0)."""
        _check(self.lib, self.lib.satba_lm_poll(self._h, self._poll.ctypes.data_as(C.POINTER(C.c_int64)), 5))
        return tuple(int(v) for v in self._poll)

    def lm_state(self):
        """satba_lm_state: wait for the stream and return the scalars of the device-resident loop as a dict."""
        out = np.zeros(17)
        _check(self.lib, self.lib.satba_lm_state(self._h, _ptr(out), 17))
        return dict(zip(self.LM_KEYS, out))

    def solve_lm(self, ftol=1e-8, xtol=1e-8, gtol=1e-8, max_nfev=None, loss="linear", f_scale=1.0, verbose=0):
        """satba_solve_lm: the loop of satba/trf.py in C++; returns the LmStats structure."""
        if loss not in LOSSES:
            raise ValueError("`loss` must be one of {}".format(list(LOSSES)))
        o = LmOpts(ftol=ftol, xtol=xtol, gtol=gtol, f_scale=f_scale, max_nfev=-1 if max_nfev is None else int(max_nfev),
                   loss=LOSSES[loss], verbose=int(verbose))
        st = LmStats()
        _check(self.lib, self.lib.satba_solve_lm(self._h, C.byref(o), C.byref(st)))
        return st

    def outliers(self, err=None, predef_thr=None, min_thr=1.0):
        """
        Per-camera elbow thresholds and the observations above them (ref:bundle_adjust/ba_outliers.py:112-155) for the errors
        `err` (caller's observation order; None: the reprojection errors at the current x).  Returns (cam_thr (M,),
        remove (K,) bool, n_removed).
        """
        thr = np.empty(self.n_cam)
        rm = np.zeros(max(self.n_obs, 1), dtype=np.uint8)
        n = C.c_int64()
        e = None if err is None else np.ascontiguousarray(err, dtype=np.float64)
        if e is not None and e.size != self.n_obs:
            raise ValueError("err has {} entries, expected {}".format(e.size, self.n_obs))
        _check(self.lib, self.lib.satba_outliers(self._h, _ptr(e) if e is not None else None, -1.0 if predef_thr is None else float(predef_thr),
                                                 float(min_thr), _ptr(thr), rm.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(n)))
        return thr, rm[: self.n_obs].astype(bool), int(n.value)

    def init_pts3d(self, camera_table, pairs, remove=None):
        """
        ft_triangulate.init_pts3d on the handle's RESIDENT observations (ref:bundle_adjust/ba_outliers.py:89-93 re-triangulates right
        after the outlier rejection; the tracks are not uploaded again).  camera_table: (n_cam, 12 | 90) float64 (ft_triangulate's
        table of projection matrices / RPC records), pairs: (n_pairs, 2), remove: (K,) bool in the caller's observation order or None.
        Returns (pts3d (N, 3) float32, n_tri (N,) int32, kernel_ms).
        """
        tab = np.ascontiguousarray(camera_table, dtype=np.float64)
        if tab.shape[0] != self.n_cam:
            raise ValueError("camera table has {} rows, expected {}".format(tab.shape[0], self.n_cam))
        pr = np.ascontiguousarray(np.asarray(pairs, dtype=np.int32).reshape(-1, 2))
        rm = None
        if remove is not None:
            rm = np.ascontiguousarray(remove, dtype=np.uint8)
            if rm.size != self.n_obs:
                raise ValueError("remove has {} entries, expected {}".format(rm.size, self.n_obs))
        out = np.zeros((self.n_pts, 3), dtype=np.float32)
        n_tri = np.zeros(max(self.n_pts, 1), dtype=np.int32)
        ms = C.c_float(0.0)
        _check(self.lib, self.lib.satba_init_pts3d_resident(self._h, rm.ctypes.data_as(C.POINTER(C.c_uint8)) if rm is not None else None,
                                                            _ptr(tab), pr.shape[0], _ptr(pr, _ip), out.ctypes.data_as(C.POINTER(C.c_float)),
                                                            _ptr(n_tri, _ip), C.byref(ms)))
        return out, n_tri[: self.n_pts], ms.value

    # -- inspection (parity tests)
    def get_layout(self, name):
        """One of the index structures satba_problem_create built on the device (csrc/satba_layout.h), as a numpy array."""
        which = LAYOUT[name]
        n = int(self.lib.satba_layout_len(self._h, which))
        if n < 0:
            raise ValueError("layout array {} does not exist (yet)".format(name))
        out = np.empty(n, dtype=np.int64 if name == "pair_ofs" else np.int32)
        _check(self.lib, self.lib.satba_get_layout(self._h, which, n, out.ctypes.data_as(C.c_void_p)))
        return out

    def info(self):
        v = np.zeros(20)
        _check(self.lib, self.lib.satba_get_info(self._h, _ptr(v), 20))
        keys = ["ms_uploads", "ms_sizes", "ms_ell", "ms_pairs", "ms_create", "ell_len", "pair_entries", "pair_chunks", "unit_weights",
                "camc_lds", "rpc_lds", "cam_sums_lds", "deterministic", "cm_chunks", "lin_grid", "fx_fallbacks", "device_loop", "chol_beside", "chol_beside_timeouts", "diag_items_per_chunk"]
        return dict(zip(keys, v[: len(keys)]))

    def get_blocks(self):
        U = np.empty((self.n_cam, self.n_p, self.n_p))
        gc = np.empty((self.n_cam, self.n_p))
        V = np.empty((self.n_pts, 6))
        gp = np.empty((self.n_pts, 3))
        _check(self.lib, self.lib.satba_get_blocks(self._h, _ptr(U), _ptr(gc), _ptr(V), _ptr(gp)))
        return U, gc, V, gp

    def get_jacobian(self):
        Jc = np.empty((self.n_obs, 2, self.n_p))
        Jp = np.empty((self.n_obs, 2, 3))
        _check(self.lib, self.lib.satba_get_jacobian(self._h, _ptr(Jc), _ptr(Jp)))
        return Jc, Jp

    def get_exchange(self, offset, n):
        out = np.empty(n)
        _check(self.lib, self.lib.satba_get_exchange(self._h, int(offset), int(n), _ptr(out)))
        return out

    def set_exchange(self, offset, values):
        values = np.ascontiguousarray(values, dtype=np.float64)
        _check(self.lib, self.lib.satba_set_exchange(self._h, int(offset), values.size, _ptr(values)))

    VECTORS = {"g": 0, "scale_inv": 1, "gn_h": 2, "q1": 3, "w": 4, "x_new": 5, "g_h": 6}

    def get_vector(self, name):
        out = np.empty(self.n)
        _check(self.lib, self.lib.satba_get_vector(self._h, self.VECTORS[name], _ptr(out)))
        return out

    KERNELS = {"residual": 0, "linearize": 1, "schur": 2, "cholesky": 3, "backsub": 4, "jvp": 5}

    def time_kernel(self, name, reps=10):
        """Average milliseconds per launch, HIP events on the engine's stream (see satba_time_kernel)."""
        ms = C.c_float()
        _check(self.lib, self.lib.satba_time_kernel(self._h, self.KERNELS[name], int(reps), C.byref(ms)))
        return ms.value
