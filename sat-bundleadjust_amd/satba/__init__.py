"""
satba -- MI355X-native bundle-adjustment core (host side).

Python mirror of the reference's `bundle_adjust.ba_core` / `bundle_adjust.ba_params`
call surface (ref:bundle_adjust/ba_core.py, ref:bundle_adjust/ba_params.py).  All numerics of
the least-squares hot path run in hand-written HIP kernels behind the C ABI declared in
include/satba.h (libsatba_hip.so); this package holds host logic only:

    ba_params   BundleAdjustmentParameters (variable packing / unpacking)
    ba_core     fun, run_ba_optimization, ... (drop-in names, device backed)
    trf         trust-region-reflective / Levenberg-Marquardt outer loop (scipy semantics)
    engine_hip  ctypes binding of the C ABI
    sharding    point sharding of a problem across ranks (one process per GPU)
    synth       seeded synthetic scenes
"""

__all__ = ["ba_core", "ba_params", "ba_rotate", "cam_utils", "geo_utils", "rpc_model", "trf", "synth"]
