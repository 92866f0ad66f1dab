"""
WGS-84 geocentric <-> geodetic conversions used inside every RPC residual
(ref:bundle_adjust/geo_utils.py:218-255).  Host (numpy) versions; the device version of
`ecef_to_latlon_custom` and its 3x3 Jacobian live in csrc/camera_models.h.
"""
import numpy as np

WGS84_A = 6378137.0
WGS84_E = 8.1819190842622e-2  # first eccentricity as hard-coded at ref:bundle_adjust/geo_utils.py:241
WGS84_FINV = 298.257223563


def latlon_to_ecef_custom(lat, lon, alt):
    """(lat, lon [deg], alt [m]) -> ECEF (x, y, z)  (ref:bundle_adjust/geo_utils.py:218-233)."""
    phi, lam = np.deg2rad(lat), np.deg2rad(lon)
    f = 1.0 / WGS84_FINV
    e2 = 1.0 - (1.0 - f) ** 2
    s = np.sin(phi)
    nu = WGS84_A / np.sqrt(1.0 - e2 * s * s)
    x = (nu + alt) * np.cos(phi) * np.cos(lam)
    y = (nu + alt) * np.cos(phi) * np.sin(lam)
    z = (nu * (1.0 - e2) + alt) * s
    return x, y, z


def ecef_to_latlon_custom(x, y, z):
    """ECEF -> (lat, lon [deg], alt [m]) by the closed-form Bowring step (ref:bundle_adjust/geo_utils.py:236-255)."""
    a, esq = WGS84_A, WGS84_E ** 2
    b = np.sqrt(a * a * (1.0 - esq))
    ep2 = (a * a - b * b) / (b * b)
    p = np.sqrt(x * x + y * y)
    th = np.arctan2(a * z, b * p)
    lon = np.arctan2(y, x)
    lat = np.arctan2(z + ep2 * b * np.sin(th) ** 3, p - esq * a * np.cos(th) ** 3)
    nu = a / np.sqrt(1.0 - esq * np.sin(lat) ** 2)
    alt = p / np.cos(lat) - nu
    return np.rad2deg(lat), np.rad2deg(lon), alt
