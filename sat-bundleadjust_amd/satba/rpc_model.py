"""
Minimal RPC camera model with the attribute names of `rpcm.RPCModel` (the third-party class the
reference passes around as `cameras[i]` when cam_model == "rpc"; call site
ref:bundle_adjust/cam_utils.py:228-230).  Only what the bundle-adjustment path needs:
reading the `KEY: value` text format (ref:tests/data/images/*.rpc, parser mirror
ref:c/rpc.c:60-92), the forward projection, and flattening to the 90-double table the HIP
kernels consume.

Term order of the 20-coefficient cubic (RPC00B), with L = normalised lon, P = lat, H = alt
(ref:bundle_adjust/ba_rpcfit.py:17-44, ref:c/rpc.c:279-298):
    1, L, P, H, LP, LH, PH, L2, P2, H2, PLH, L3, LP2, LH2, L2P, P3, PH2, L2H, P2H, H3
"""
import numpy as np

_SCALARS = [
    ("LINE_OFF", "row_offset"), ("SAMP_OFF", "col_offset"), ("LAT_OFF", "lat_offset"),
    ("LONG_OFF", "lon_offset"), ("HEIGHT_OFF", "alt_offset"), ("LINE_SCALE", "row_scale"),
    ("SAMP_SCALE", "col_scale"), ("LAT_SCALE", "lat_scale"), ("LONG_SCALE", "lon_scale"),
    ("HEIGHT_SCALE", "alt_scale"),
]
_POLYS = [("LINE_NUM_COEFF", "row_num"), ("LINE_DEN_COEFF", "row_den"),
          ("SAMP_NUM_COEFF", "col_num"), ("SAMP_DEN_COEFF", "col_den")]

# layout of the flat table handed to the device (include/satba.h: SATBA_RPC_TABLE_LEN)
RPC_TABLE_LEN = 90


def rpc_monomials(L, P, H):
    """The 20 cubic monomials in RPC00B order, stacked on the first axis."""
    one = np.ones_like(L)
    return np.stack([one, L, P, H, L * P, L * H, P * H, L * L, P * P, H * H, P * L * H, L * L * L, L * P * P,
                     L * H * H, L * L * P, P * P * P, P * H * H, L * L * H, P * P * H, H * H * H])


class RPCModel:
    def __init__(self, **kw):
        for _, attr in _SCALARS:
            setattr(self, attr, float(kw[attr]))
        for _, attr in _POLYS:
            v = np.asarray(kw[attr], dtype=np.float64)
            assert v.shape == (20,)
            setattr(self, attr, v.tolist())

    @classmethod
    def from_file(cls, path):
        vals = {}
        with open(path) as f:
            for line in f:
                if ":" not in line:
                    continue
                key, rest = line.split(":", 1)
                tok = rest.split()
                if tok:
                    try:
                        vals[key.strip()] = float(tok[0])
                    except ValueError:
                        pass
        kw = {attr: vals[key] for key, attr in _SCALARS}
        for key, attr in _POLYS:
            kw[attr] = [vals["{}_{}".format(key, i)] for i in range(1, 21)]
        return cls(**kw)

    def copy(self):
        return RPCModel(**{a: getattr(self, a) for _, a in _SCALARS + _POLYS})

    def projection(self, lon, lat, alt):
        """(lon, lat [deg], alt [m]) -> (col, row) pixels; same signature as rpcm.RPCModel.projection."""
        L = (np.asarray(lon, dtype=np.float64) - self.lon_offset) / self.lon_scale
        P = (np.asarray(lat, dtype=np.float64) - self.lat_offset) / self.lat_scale
        H = (np.asarray(alt, dtype=np.float64) - self.alt_offset) / self.alt_scale
        m = rpc_monomials(L, P, H)
        col = np.tensordot(self.col_num, m, 1) / np.tensordot(self.col_den, m, 1)
        row = np.tensordot(self.row_num, m, 1) / np.tensordot(self.row_den, m, 1)
        return col * self.col_scale + self.col_offset, row * self.row_scale + self.row_offset

    @classmethod
    def from_table(cls, t):
        """inverse of to_table"""
        t = np.asarray(t, dtype=np.float64)
        return cls(col_num=t[0:20], col_den=t[20:40], row_num=t[40:60], row_den=t[60:80], lon_offset=t[80], lon_scale=t[81],
                   lat_offset=t[82], lat_scale=t[83], alt_offset=t[84], alt_scale=t[85], col_offset=t[86], col_scale=t[87],
                   row_offset=t[88], row_scale=t[89])

    def localization(self, col, row, alt):
        """(col, row [px], alt [m]) -> (lon, lat) degrees by inverting the projection on the device (satba_rpc_localization); same
        signature as rpcm.RPCModel.localization for a model without inverse coefficients (call sites ref:bundle_adjust/ba_rpcfit.py:245,323)."""

        from . import engine_hip as E

        lib = E.load_library()
        col, row, alt = np.broadcast_arrays(np.asarray(col, dtype=np.float64), np.asarray(row, dtype=np.float64), np.asarray(alt, dtype=np.float64))
        shape = col.shape
        c, r, a = [np.ascontiguousarray(v.reshape(-1)) for v in (col, row, alt)]
        lon, lat = np.zeros_like(c), np.zeros_like(c)
        tab = np.ascontiguousarray(self.to_table())
        import os
        E._check(lib, lib.satba_rpc_localization(E._ptr(tab), c.size, E._ptr(c), E._ptr(r), E._ptr(a), E._ptr(lon), E._ptr(lat),
                                                 int(os.environ.get("LOCAL_RANK", "0"))))
        return lon.reshape(shape), lat.reshape(shape)

    def write_to_file(self, path):
        """The `KEY: value unit` text format of the reference's outputs (ref:tests/data/outdir/ba_bruteforce/rpcs_adj/*.rpc_adj,
        written there by rpcm.RPCModel.write_to_file from ref:bundle_adjust/ba_pipeline.py:379-427); read back by from_file."""
        units = {"LINE_OFF": "pixels", "SAMP_OFF": "pixels", "LAT_OFF": "degrees", "LONG_OFF": "degrees", "HEIGHT_OFF": "meters",
                 "LINE_SCALE": "pixels", "SAMP_SCALE": "pixels", "LAT_SCALE": "degrees", "LONG_SCALE": "degrees", "HEIGHT_SCALE": "meters"}
        with open(path, "w") as f:
            for key, attr in _SCALARS:
                f.write("{}: {:.12f} {}\n".format(key, getattr(self, attr), units[key]))
            for key, attr in _POLYS:
                for i, v in enumerate(getattr(self, attr)):
                    f.write("{}_{}: {:.12f}\n".format(key, i + 1, v))

    def to_table(self):
        """[col_num(20) col_den(20) row_num(20) row_den(20) lon_off lon_scale lat_off lat_scale alt_off alt_scale
        col_off col_scale row_off row_scale] -- the per-camera record of include/satba.h."""
        return np.concatenate([
            self.col_num, self.col_den, self.row_num, self.row_den,
            [self.lon_offset, self.lon_scale, self.lat_offset, self.lat_scale, self.alt_offset, self.alt_scale,
             self.col_offset, self.col_scale, self.row_offset, self.row_scale]]).astype(np.float64)


def rpc_to_table(rpc):
    """Flatten any object exposing the rpcm attribute names (rpcm.RPCModel or RPCModel above)."""
    return RPCModel.to_table(rpc)
