"""
CPU: shapes in another coordinate system than the cutout (``shapes_crs``; atlite/convert.py:235-240 -> cutout.py:492-515 ->
gis.py:130: ``dest = reproject_shapes(dest, dest_crs, orig_crs)`` - the shapes' vertices are moved into the cutout's crs and
the overlaps are taken against the rectangular cells there).  ``atlite_amd.crs`` writes the projections out (pyproj is not
in this image - parity with it is unpinned): the forward transforms are checked against the worked examples of IOGP
Guidance Note 7-2, against an independent series (Snyder) and a numerically integrated meridian arc for UTM, and against
the defining properties (equal area, conformality); the inverse transforms - what the indicator matrix uses - against the
guidance note's reverse examples and by round trips to 1e-11 degree; the indicator matrix of projected shapes against the
matrix of the same vertices given in geographic coordinates.
"""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

from atlite_amd import _lib, crs, gis


def test_guidance_note_examples():
    x, y = crs.forward(3035, 5.0, 50.0)  # GN 7-2, Lambert azimuthal equal area: ETRS89-extended / LAEA Europe
    assert abs(x - 3962799.45) < 0.01 and abs(y - 2999718.85) < 0.01
    x, y = crs.forward("EPSG:3035", 10.0, 52.0)  # the projection's origin
    assert abs(x - 4321000.0) < 1e-6 and abs(y - 3210000.0) < 1e-6
    x, y = crs.forward(3857, -(100 + 20 / 60), 24 + 22 / 60 + 54.433 / 3600)  # GN 7-2, popular visualisation pseudo-Mercator
    assert abs(x + 11169055.58) < 0.01 and abs(y - 2800000.00) < 0.01
    for code in (4326, 4258, "OGC:CRS84"):
        x, y = crs.forward(code, [3.5, -7.25], [40.0, 61.5])
        assert np.array_equal(x, [3.5, -7.25]) and np.array_equal(y, [40.0, 61.5])


def _snyder_utm(lon, lat, lon0, a=6378137.0, f=1 / 298.257223563):
    """Snyder (1987), eqs. 8-9 / 8-10 and 3-21: an independent series for the transverse Mercator."""
    e2 = f * (2 - f)
    ep2 = e2 / (1 - e2)
    phi, lam = np.radians(lat), np.radians(lon)
    N = a / np.sqrt(1 - e2 * np.sin(phi) ** 2)
    T, Cc, A = np.tan(phi) ** 2, ep2 * np.cos(phi) ** 2, (lam - np.radians(lon0)) * np.cos(phi)
    M = a * ((1 - e2 / 4 - 3 * e2**2 / 64 - 5 * e2**3 / 256) * phi - (3 * e2 / 8 + 3 * e2**2 / 32 + 45 * e2**3 / 1024) * np.sin(2 * phi)
             + (15 * e2**2 / 256 + 45 * e2**3 / 1024) * np.sin(4 * phi) - (35 * e2**3 / 3072) * np.sin(6 * phi))
    k0 = 0.9996
    x = k0 * N * (A + (1 - T + Cc) * A**3 / 6 + (5 - 18 * T + T**2 + 72 * Cc - 58 * ep2) * A**5 / 120)
    y = k0 * (M + N * np.tan(phi) * (A**2 / 2 + (5 - T + 9 * Cc + 4 * Cc**2) * A**4 / 24 + (61 - 58 * T + T**2 + 600 * Cc - 330 * ep2) * A**6 / 720))
    return 500000.0 + x, y


def test_utm_against_an_independent_series_and_the_meridian_arc():
    rng = np.random.default_rng(0)
    lat = rng.uniform(-80, 84, 400)
    dl = rng.uniform(-3.5, 3.5, 400)
    for zone in (29, 32, 33, 60):
        lon0 = 6 * zone - 183
        x, y = crs.forward(32600 + zone, lon0 + dl, lat)
        xs, ys = _snyder_utm(lon0 + dl, lat, lon0)
        assert np.abs(x - xs).max() < 2e-3 and np.abs(y - ys).max() < 2e-3  # Snyder's series is good to ~1 mm inside a zone
        xsouth, ysouth = crs.forward(32700 + zone, lon0 + dl, lat)
        assert np.array_equal(xsouth, x) and np.allclose(ysouth - y, 1.0e7, rtol=0, atol=1e-8)
    # on the central meridian: northing = 0.9996 x meridian arc (numerically integrated), easting = 500 km exactly
    a, f = crs.WGS84
    e2 = f * (2 - f)
    for lat1 in (10.0, 48.0, 71.5):
        p = np.linspace(0.0, np.radians(lat1), 200001)
        g = a * (1 - e2) * (1 - e2 * np.sin(p) ** 2) ** -1.5
        arc = np.sum((g[1:] + g[:-1]) / 2 * np.diff(p))
        x, y = crs.forward(32632, 9.0, lat1)
        assert abs(x - 500000.0) < 1e-9 and abs(y - 0.9996 * arc) < 1e-3
    xe, ye = crs.forward(25832, 9.5, 50.0)  # ETRS89 / UTM 32N: GRS80, within a tenth of a millimetre of WGS84's
    xw, yw = crs.forward(32632, 9.5, 50.0)
    assert abs(xe - xw) < 1e-3 and abs(ye - yw) < 1e-3 and (xe, ye) != (xw, yw)


def _jacobian(code, lon, lat, h=1e-6):
    x1, y1 = crs.forward(code, lon + h, lat)
    x0, y0 = crs.forward(code, lon - h, lat)
    x3, y3 = crs.forward(code, lon, lat + h)
    x2, y2 = crs.forward(code, lon, lat - h)
    return np.array([[(x1 - x0), (x3 - x2)], [(y1 - y0), (y3 - y2)]]) / (2 * np.radians(h))


def test_defining_properties():
    a, f = crs.GRS80
    e2 = f * (2 - f)
    for lon, lat in ((-9.0, 38.0), (10.0, 52.0), (25.0, 67.0), (3.0, 45.0)):
        nu = a / np.sqrt(1 - e2 * np.sin(np.radians(lat)) ** 2)
        rho = a * (1 - e2) / (1 - e2 * np.sin(np.radians(lat)) ** 2) ** 1.5
        # metric of the ellipsoid: d(east) = nu cos(lat) d(lon), d(north) = rho d(lat)
        scale = np.diag([1.0 / (nu * np.cos(np.radians(lat))), 1.0 / rho])
        J = _jacobian(3035, lon, lat) @ scale
        assert abs(np.linalg.det(J) - 1.0) < 1e-6  # equal area
        for code in (32600 + int((lon + 180) // 6) + 1, 3857):
            # (the pseudo-Mercator is conformal on the SPHERE of the semi-major axis its formulas assume)
            J = _jacobian(code, lon, lat) @ (scale if code != 3857 else np.diag([1 / (crs.WGS84[0] * np.cos(np.radians(lat))), 1 / crs.WGS84[0]]))
            assert abs(J[0, 0] - J[1, 1]) < 2e-6 * abs(J[0, 0]) and abs(J[0, 1] + J[1, 0]) < 2e-6 * abs(J[0, 0])  # conformal: a similarity


def test_reading_crs_descriptions():
    class P:
        def to_epsg(self):
            return 3035

    assert [crs.epsg_of(c) for c in (3035, "EPSG:3035", "epsg:3035", "3035", P(), {"init": "epsg:3035"}, "urn:ogc:def:crs:EPSG::3035")] == [3035] * 7
    assert crs.same_crs(4326, "EPSG:4258") and crs.same_crs("EPSG:4326", 4326) and not crs.same_crs(4326, 3035)
    with pytest.raises(NotImplementedError, match="not among the projections"):
        crs.forward(27700, 0.0, 52.0)
    with pytest.raises(NotImplementedError, match="cannot read"):
        crs.epsg_of("+proj=laea +lat_0=52")


def test_inverse_guidance_note_examples_and_round_trips():
    lon, lat = crs.inverse(3035, 3962799.45, 2999718.85)  # GN 7-2 reverse case of the LAEA example
    assert abs(lon - 5.0) < 2e-7 and abs(lat - 50.0) < 2e-7
    lon, lat = crs.inverse(3857, -11169055.58, 2800000.00)
    assert abs(lon + (100 + 20 / 60)) < 1e-7 and abs(lat - (24 + 22 / 60 + 54.433 / 3600)) < 1e-7
    lon, lat = crs.inverse(3035, 4321000.0, 3210000.0)  # the projection's origin (rho = 0)
    assert abs(lon - 10.0) < 1e-12 and abs(lat - 52.0) < 1e-12
    rng = np.random.default_rng(1)
    lo, la = rng.uniform(-25, 45, 2000), rng.uniform(28, 72, 2000)
    for code in (3035, 3857, 4326):
        x, y = crs.forward(code, lo, la)
        lo2, la2 = crs.inverse(code, x, y)
        assert np.abs(lo2 - lo).max() < 1e-11 and np.abs(la2 - la).max() < 1e-11, code
    for zone in (29, 32, 36):
        lon0 = 6 * zone - 183
        lo = lon0 + rng.uniform(-4, 4, 2000)
        for base in (32600, 32700, 25800):
            x, y = crs.forward(base + zone, lo, la)
            lo2, la2 = crs.inverse(base + zone, x, y)
            assert np.abs(lo2 - lo).max() < 1e-11 and np.abs(la2 - la).max() < 1e-11, (base, zone)
    with pytest.raises(NotImplementedError, match="not among the projections"):
        crs.inverse(27700, 0.0, 0.0)


@pytest.mark.parametrize("code", [3035, 32632, 3857])
def test_shapes_in_a_projected_crs_are_moved_into_the_cutouts(code):
    """The reference's method (gis.py:130): only the vertices move; the matrix of projected shapes equals the matrix of
    the same vertices given in geographic coordinates, on the host and through Cutout.indicatormatrix."""
    from atlite_amd import Cutout, Dataset

    x, y = np.arange(5.0, 12.01, 0.25), np.arange(47.0, 53.01, 0.25)
    ring = np.array([[6.1, 48.2], [10.7, 47.9], [11.2, 51.8], [8.0, 52.6], [5.6, 50.3]])
    hole = np.array([[8.0, 49.5], [9.0, 49.5], [9.0, 50.5], [8.0, 50.5]])
    outside = np.array([[11.0, 52.0], [14.0, 52.0], [14.0, 55.0], [11.0, 55.0]])  # partly beside the grid
    proj = lambda r: np.stack(crs.forward(code, r[:, 0], r[:, 1]), axis=1)  # noqa: E731
    geo = [dict(exterior=ring, holes=[hole]), ring, outside]
    shapes = [dict(exterior=proj(ring), holes=[proj(hole)]), proj(ring), proj(outside)]
    cut = Cutout(Dataset({"height": np.zeros((len(y), len(x)))}, dict(y=y, x=x)))
    M = cut.indicatormatrix(shapes, shapes_crs=code, where="host")
    M0 = cut.indicatormatrix(geo, where="host")
    assert M.shape == (3, len(x) * len(y)) and M.data.min() > 0 and M.data.max() <= 1.0
    assert np.abs(M - M0).max() < 1e-9 and M.nnz >= M0.nnz - 4  # (vertices round-trip to 1e-11 degree)
    # areas in degrees of the parts on the grid
    cell = 0.25 * 0.25
    shoelace = lambda p: 0.5 * abs(np.sum(p[:, 0] * np.roll(p[:, 1], -1) - np.roll(p[:, 0], -1) * p[:, 1]))  # noqa: E731
    got = np.asarray(M.sum(1)).ravel() * cell
    assert abs(got[1] - shoelace(ring)) < 1e-8 and abs(got[0] - (shoelace(ring) - shoelace(hole))) < 1e-8
    assert got[2] < 0.5 * shoelace(outside)  # only the part on the grid counts
    # a second call with the same shapes comes from the cutout's cache; another crs code is another key
    assert np.array_equal(cut.indicatormatrix(shapes, shapes_crs=code, where="host").toarray(), M.toarray())
    with pytest.raises(NotImplementedError, match="geographic"):
        Cutout(Dataset({"height": np.zeros((len(y), len(x)))}, dict(y=y, x=x)), crs=3035).indicatormatrix(shapes, shapes_crs=32632, where="host")
