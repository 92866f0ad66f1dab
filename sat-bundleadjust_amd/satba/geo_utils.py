"""
WGS-84 geocentric <-> geodetic conversions used inside every RPC residual
(ref:bundle_adjust/geo_utils.py:218-255).  Host (numpy) versions; the device version of
`ecef_to_latlon_custom` and its 3x3 Jacobian live in csrc/satba_models.h (`geodetic<JAC>`).
The UTM helpers serve the figure functions of satba/ba_core.py only (ref:bundle_adjust/geo_utils.py:15-97 does the
same through pyproj / utm, which are not dependencies of this package).
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


# ----------------------------------------------------------------------------- UTM (figures only)

def utm_zone_from_lonlat(lon, lat):
    """Zone number of the standard 6-degree UTM grid (no Norway / Svalbard exceptions)."""
    return int(np.floor((float(lon) + 180.0) / 6.0)) % 60 + 1


def utm_from_lonlat(lons, lats, zone=None):
    """
    (easting, northing) [m] in the UTM zone of the first point (as ref:bundle_adjust/geo_utils.py:22-30 picks it), by the
    Krueger n-series of the transverse Mercator projection to n^4 (sub-millimetre inside a zone).  Southern northings
    are negative, as pyproj's "+proj=utm" without "+south" returns them.
    """
    lons, lats = np.atleast_1d(np.asarray(lons, dtype=np.float64)), np.atleast_1d(np.asarray(lats, dtype=np.float64))
    zone = utm_zone_from_lonlat(lons[0], lats[0]) if zone is None else zone
    lam0 = np.deg2rad(6.0 * zone - 183.0)
    f = 1.0 / WGS84_FINV
    n = f / (2.0 - f)
    A = WGS84_A / (1.0 + n) * (1.0 + n ** 2 / 4.0 + n ** 4 / 64.0)
    al = [n / 2.0 - 2.0 * n ** 2 / 3.0 + 5.0 * n ** 3 / 16.0 + 41.0 * n ** 4 / 180.0,
          13.0 * n ** 2 / 48.0 - 3.0 * n ** 3 / 5.0 + 557.0 * n ** 4 / 1440.0,
          61.0 * n ** 3 / 240.0 - 103.0 * n ** 4 / 140.0,
          49561.0 * n ** 4 / 161280.0]
    phi, lam = np.deg2rad(lats), np.deg2rad(lons)
    e = np.sqrt(f * (2.0 - f))
    t = np.sinh(np.arctanh(np.sin(phi)) - e * np.arctanh(e * np.sin(phi)))
    xi = np.arctan2(t, np.cos(lam - lam0))
    eta = np.arctanh(np.sin(lam - lam0) / np.sqrt(1.0 + t * t))
    x, y = eta.copy(), xi.copy()
    for j, a in enumerate(al, start=1):
        x += a * np.cos(2 * j * xi) * np.sinh(2 * j * eta)
        y += a * np.sin(2 * j * xi) * np.cosh(2 * j * eta)
    k0 = 0.9996
    return 500000.0 + k0 * A * x, k0 * A * y


def epsg_code_from_utm_zone(zone, north=True):
    """EPSG code of a WGS-84 UTM zone (ref:bundle_adjust/geo_utils.py:43-54)."""
    return (32600 if north else 32700) + int(zone)
