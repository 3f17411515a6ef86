"""
Map projections for shapes that come in another coordinate system than the cutout (``shapes_crs`` of
``convert_and_aggregate`` / ``Cutout.indicatormatrix``, atlite/convert.py:235-240 -> cutout.py:492-515 -> gis.py:128-133).

The reference hands this to pyproj (``dest = reproject_shapes(dest, dest_crs, orig_crs)``, gis.py:130: the VERTICES OF THE
SHAPES are moved from their crs into the cutout's, the overlaps are taken there, against the rectangular grid cells).  pyproj
is not part of this image, so the transforms of the projections energy-system shapes usually come in are written out here,
to and from geographic coordinates on the GRS80 / WGS84 ellipsoid (EPSG:4326 / 4258 - what ERA5 and SARAH cutouts use;
``inverse`` is what the indicator matrix needs, ``forward`` is its check):

* EPSG:3035  ETRS89-extended / LAEA Europe  (Lambert azimuthal equal area, EPSG method 9820)
* EPSG:3857  WGS 84 / Pseudo-Mercator       (EPSG method 1024, spherical formulas on the semi-major axis)
* EPSG:326zz / 327zz  WGS 84 / UTM zone zz N / S,  EPSG:258zz  ETRS89 / UTM zone zz N  (transverse Mercator, EPSG method
  9807, Krueger's series in the third flattening to n^4: sub-millimetre inside a zone and far beyond)
* EPSG:4326 / 4258 / "OGC:CRS84"  geographic: the identity

Formulas: IOGP Guidance Note 7-2 (Coordinate Conversions and Transformations including Formulas).  Parity note: unlike the
oracle this module cannot be pinned against the reference's dependency here; it is checked against the guidance note's worked
examples, against closed-form properties (equal area, conformality, scale on the central meridian) and by round trips.
"""
from __future__ import annotations

import re

import numpy as np

GRS80 = (6378137.0, 1.0 / 298.257222101)
WGS84 = (6378137.0, 1.0 / 298.257223563)
GEOGRAPHIC = {4326, 4258, 4979}


def epsg_of(crs):
    """EPSG code of ``crs``: an int, 'EPSG:3035' / 'epsg:3035' / '3035', 'OGC:CRS84', or anything with ``to_epsg()``."""
    if isinstance(crs, (int, np.integer)):
        return int(crs)
    if hasattr(crs, "to_epsg"):
        code = crs.to_epsg()
        if code is None:
            raise NotImplementedError(f"crs {crs!r} has no EPSG code")
        return int(code)
    if isinstance(crs, str):
        s = crs.strip()
        if s.upper() in ("OGC:CRS84", "CRS84", "WGS84", "WGS 84"):
            return 4326
        m = re.fullmatch(r"(?:(?:urn:ogc:def:crs:)?EPSG:{1,2})?(\d+)", s, flags=re.IGNORECASE)
        if m:
            return int(m.group(1))
    if isinstance(crs, dict) and str(crs.get("init", "")).lower().startswith("epsg:"):
        return int(str(crs["init"]).split(":")[1])
    raise NotImplementedError(f"cannot read the coordinate system {crs!r}: give an EPSG code (3035, 'EPSG:32632', ...)")


def same_crs(a, b):
    """Do the two describe the same system (every geographic code on GRS80 / WGS84 counts as one: they agree to < 1 m)?"""
    if a is b or a == b:
        return True
    try:
        ea, eb = epsg_of(a), epsg_of(b)
    except NotImplementedError:
        return False
    return ea == eb or (ea in GEOGRAPHIC and eb in GEOGRAPHIC)


def _laea(lon, lat, a, f, lon0, lat0, fe, fn):
    """Lambert azimuthal equal area, oblique aspect (EPSG 9820)."""
    e2 = f * (2.0 - f)
    e = np.sqrt(e2)
    phi, lam = np.radians(lat), np.radians(lon)
    phi0, lam0 = np.radians(lat0), np.radians(lon0)

    def q_of(p):
        s = np.sin(p)
        return (1.0 - e2) * (s / (1.0 - e2 * s * s) - (0.5 / e) * np.log((1.0 - e * s) / (1.0 + e * s)))

    qp = q_of(np.pi / 2.0)
    q, q0 = q_of(phi), q_of(phi0)
    beta, beta0 = np.arcsin(np.clip(q / qp, -1.0, 1.0)), np.arcsin(q0 / qp)
    rq = a * np.sqrt(qp / 2.0)
    d = a * (np.cos(phi0) / np.sqrt(1.0 - e2 * np.sin(phi0) ** 2)) / (rq * np.cos(beta0))
    dl = lam - lam0
    b = rq * np.sqrt(2.0 / (1.0 + np.sin(beta0) * np.sin(beta) + np.cos(beta0) * np.cos(beta) * np.cos(dl)))
    x = fe + b * d * np.cos(beta) * np.sin(dl)
    y = fn + (b / d) * (np.cos(beta0) * np.sin(beta) - np.sin(beta0) * np.cos(beta) * np.cos(dl))
    return x, y


def _tmerc(lon, lat, a, f, lon0, k0, fe, fn):
    """Transverse Mercator (EPSG 9807), Krueger series; latitude of origin 0 (UTM)."""
    n = f / (2.0 - f)
    n2, n3, n4 = n * n, n ** 3, n ** 4
    B = a / (1.0 + n) * (1.0 + n2 / 4.0 + n4 / 64.0)
    h1 = n / 2.0 - 2.0 / 3.0 * n2 + 5.0 / 16.0 * n3 + 41.0 / 180.0 * n4
    h2 = 13.0 / 48.0 * n2 - 3.0 / 5.0 * n3 + 557.0 / 1440.0 * n4
    h3 = 61.0 / 240.0 * n3 - 103.0 / 140.0 * n4
    h4 = 49561.0 / 161280.0 * n4
    e = np.sqrt(f * (2.0 - f))
    phi, dl = np.radians(lat), np.radians(lon) - np.radians(lon0)
    Q = np.arcsinh(np.tan(phi)) - e * np.arctanh(e * np.sin(phi))
    beta = np.arctan(np.sinh(Q))
    eta0 = np.arctanh(np.cos(beta) * np.sin(dl))
    xi0 = np.arcsin(np.clip(np.sin(beta) * np.cosh(eta0), -1.0, 1.0))
    xi = xi0 + h1 * np.sin(2 * xi0) * np.cosh(2 * eta0) + h2 * np.sin(4 * xi0) * np.cosh(4 * eta0) \
        + h3 * np.sin(6 * xi0) * np.cosh(6 * eta0) + h4 * np.sin(8 * xi0) * np.cosh(8 * eta0)
    eta = eta0 + h1 * np.cos(2 * xi0) * np.sinh(2 * eta0) + h2 * np.cos(4 * xi0) * np.sinh(4 * eta0) \
        + h3 * np.cos(6 * xi0) * np.sinh(6 * eta0) + h4 * np.cos(8 * xi0) * np.sinh(8 * eta0)
    return fe + k0 * B * eta, fn + k0 * B * xi


def forward(crs, lon, lat):
    """(x, y) in ``crs`` of geographic coordinates in degrees (arrays broadcast)."""
    code = epsg_of(crs)
    lon, lat = np.asarray(lon, dtype=np.float64), np.asarray(lat, dtype=np.float64)
    if code in GEOGRAPHIC:
        return lon + 0.0 * lat, lat + 0.0 * lon
    if code == 3035:
        return _laea(lon, lat, *GRS80, 10.0, 52.0, 4321000.0, 3210000.0)
    if code == 3857:
        a = WGS84[0]
        return a * np.radians(lon) + 0.0 * lat, a * np.log(np.tan(np.pi / 4.0 + np.radians(lat) / 2.0)) + 0.0 * lon
    for base, ell, south in ((32600, WGS84, False), (32700, WGS84, True), (25800, GRS80, False)):
        zone = code - base
        if 1 <= zone <= 60:
            return _tmerc(lon, lat, *ell, 6.0 * zone - 183.0, 0.9996, 500000.0, 10000000.0 if south else 0.0)
    raise NotImplementedError(
        f"EPSG:{code} is not among the projections written out in atlite_amd.crs (3035, 3857, 326zz / 327zz / 258zz UTM, "
        "4326 / 4258); reproject the shapes first (pyproj is not part of this image)")


def _laea_inv(x, y, a, f, lon0, lat0, fe, fn):
    """Inverse of ``_laea`` (EPSG 9820): the authalic latitude in closed form, the geodetic one from it by the guidance
    note's series as a start and Newton steps on q(phi) = q_p sin(beta) to the last bit."""
    e2 = f * (2.0 - f)
    e = np.sqrt(e2)
    phi0, lam0 = np.radians(lat0), np.radians(lon0)

    def q_of(p):
        s = np.sin(p)
        return (1.0 - e2) * (s / (1.0 - e2 * s * s) - (0.5 / e) * np.log((1.0 - e * s) / (1.0 + e * s)))

    qp = q_of(np.pi / 2.0)
    beta0 = np.arcsin(q_of(phi0) / qp)
    rq = a * np.sqrt(qp / 2.0)
    d = a * (np.cos(phi0) / np.sqrt(1.0 - e2 * np.sin(phi0) ** 2)) / (rq * np.cos(beta0))
    xp, yp = x - fe, y - fn
    rho = np.sqrt((xp / d) ** 2 + (d * yp) ** 2)
    c = 2.0 * np.arcsin(np.clip(rho / (2.0 * rq), -1.0, 1.0))
    with np.errstate(invalid="ignore", divide="ignore"):
        sb = np.where(rho > 0.0, np.cos(c) * np.sin(beta0) + d * yp * np.sin(c) * np.cos(beta0) / np.where(rho > 0.0, rho, 1.0),
                      np.sin(beta0))
    beta = np.arcsin(np.clip(sb, -1.0, 1.0))
    lam = lam0 + np.arctan2(xp * np.sin(c), d * rho * np.cos(beta0) * np.cos(c) - d * d * yp * np.sin(beta0) * np.sin(c))
    e4, e6 = e2 * e2, e2 ** 3
    phi = beta + (e2 / 3 + 31 * e4 / 180 + 517 * e6 / 5040) * np.sin(2 * beta) + (23 * e4 / 360 + 251 * e6 / 3780) * np.sin(4 * beta) \
        + (761 * e6 / 45360) * np.sin(6 * beta)
    q = qp * np.sin(beta)
    for _ in range(4):  # dq/dphi = 2 (1 - e2) cos(phi) / (1 - e2 sin^2 phi)^2
        sp_ = np.sin(phi)
        dq = 2.0 * (1.0 - e2) * np.cos(phi) / (1.0 - e2 * sp_ * sp_) ** 2
        with np.errstate(invalid="ignore", divide="ignore"):
            step = np.where(np.abs(dq) > 1e-12, (q - q_of(phi)) / np.where(np.abs(dq) > 1e-12, dq, 1.0), 0.0)
        phi = phi + step
    return np.degrees(lam), np.degrees(phi)


def _tmerc_inv(x, y, a, f, lon0, k0, fe, fn):
    """Inverse of ``_tmerc`` (EPSG 9807, Krueger series to n^4)."""
    n = f / (2.0 - f)
    n2, n3, n4 = n * n, n ** 3, n ** 4
    B = a / (1.0 + n) * (1.0 + n2 / 4.0 + n4 / 64.0)
    h1 = n / 2.0 - 2.0 / 3.0 * n2 + 37.0 / 96.0 * n3 - 1.0 / 360.0 * n4
    h2 = 1.0 / 48.0 * n2 + 1.0 / 15.0 * n3 - 437.0 / 1440.0 * n4
    h3 = 17.0 / 480.0 * n3 - 37.0 / 840.0 * n4
    h4 = 4397.0 / 161280.0 * n4
    e = np.sqrt(f * (2.0 - f))
    eta, xi = (x - fe) / (B * k0), (y - fn) / (B * k0)
    xi0 = xi - (h1 * np.sin(2 * xi) * np.cosh(2 * eta) + h2 * np.sin(4 * xi) * np.cosh(4 * eta)
                + h3 * np.sin(6 * xi) * np.cosh(6 * eta) + h4 * np.sin(8 * xi) * np.cosh(8 * eta))
    eta0 = eta - (h1 * np.cos(2 * xi) * np.sinh(2 * eta) + h2 * np.cos(4 * xi) * np.sinh(4 * eta)
                  + h3 * np.cos(6 * xi) * np.sinh(6 * eta) + h4 * np.cos(8 * xi) * np.sinh(8 * eta))
    beta = np.arcsin(np.clip(np.sin(xi0) / np.cosh(eta0), -1.0, 1.0))
    q1 = np.arcsinh(np.tan(beta))
    q2 = q1
    for _ in range(8):  # Q'' = Q' + e atanh(e tanh Q''): contracts by e^2 per step
        q2 = q1 + e * np.arctanh(e * np.tanh(q2))
    phi = np.arctan(np.sinh(q2))
    lam = np.radians(lon0) + np.arcsin(np.clip(np.tanh(eta0) / np.cos(beta), -1.0, 1.0))
    return np.degrees(lam), np.degrees(phi)


def inverse(crs, x, y):
    """Geographic (lon, lat) in degrees of coordinates ``x``, ``y`` in ``crs`` (arrays broadcast) - what
    ``reproject_shapes(shapes, shapes_crs, cutout.crs)`` (atlite/gis.py:86-101, 130) does to every vertex of a shape."""
    code = epsg_of(crs)
    x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
    if code in GEOGRAPHIC:
        return x + 0.0 * y, y + 0.0 * x
    if code == 3035:
        return _laea_inv(x, y, *GRS80, 10.0, 52.0, 4321000.0, 3210000.0)
    if code == 3857:
        a = WGS84[0]
        return np.degrees(x / a) + 0.0 * y, np.degrees(2.0 * np.arctan(np.exp(y / a)) - np.pi / 2.0) + 0.0 * x
    for base, ell, south in ((32600, WGS84, False), (32700, WGS84, True), (25800, GRS80, False)):
        zone = code - base
        if 1 <= zone <= 60:
            return _tmerc_inv(x, y, *ell, 6.0 * zone - 183.0, 0.9996, 500000.0, 10000000.0 if south else 0.0)
    raise NotImplementedError(
        f"EPSG:{code} is not among the projections written out in atlite_amd.crs (3035, 3857, 326zz / 327zz / 258zz UTM, "
        "4326 / 4258); reproject the shapes first (pyproj is not part of this image)")

