"""
CPU: shapes in another coordinate system than the cutout (``shapes_crs``; atlite/convert.py:235-240 -> cutout.py:492-515 ->
gis.py:128-133: the corners of the cell boxes are reprojected into the shapes' crs, the overlaps are taken there).
``atlite_amd.crs`` writes the forward projections out (pyproj is not in this image - parity with it is unpinned); they are
checked against the worked examples of IOGP Guidance Note 7-2, against an independent series (Snyder) and a numerically
integrated meridian arc for UTM, and against the defining properties (equal area, conformality).  The cell-by-cell overlap
with convex quadrilaterals (``atl_indicator_polygons_quads``) is checked against the rectangular-grid clipper and by area
conservation.
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


def _quads_matrix(shapes, quads):
    lib = _lib.load()
    shape_ptr, ring_ptr, holes, xy = [0], [0], [], []
    for s in shapes:
        for ring, is_hole in gis._rings_of(s):
            xy.append(np.asarray(ring, dtype=np.float64))
            ring_ptr.append(ring_ptr[-1] + len(ring))
            holes.append(int(is_hole))
        shape_ptr.append(len(holes))
    shape_ptr, ring_ptr = np.asarray(shape_ptr, np.int64), np.asarray(ring_ptr, np.int64)
    holes, xy = np.asarray(holes, np.uint8), np.ascontiguousarray(np.concatenate(xy))
    quads = np.ascontiguousarray(quads, dtype=np.float64)
    p = [C.c_void_p() for _ in range(3)]
    _lib.check(lib.atl_indicator_polygons_quads(len(shapes), shape_ptr.ctypes.data, len(holes), ring_ptr.ctypes.data, holes.ctypes.data,
                                                xy.ctypes.data, len(quads), quads.ctypes.data, *[C.byref(v) for v in p]))
    N = len(shapes)
    indptr = np.ctypeslib.as_array(C.cast(p[0], C.POINTER(C.c_int64)), (N + 1,)).copy()
    nnz = int(indptr[-1])
    idx = np.ctypeslib.as_array(C.cast(p[1], C.POINTER(C.c_int32)), (max(nnz, 1),))[:nnz].copy()
    dat = np.ctypeslib.as_array(C.cast(p[2], C.POINTER(C.c_double)), (max(nnz, 1),))[:nnz].copy()
    for v in p:
        lib.atl_host_free(v)
    return sp.csr_matrix((dat, idx, indptr), shape=(N, len(quads)))


def _boxes(x, y):
    dx, dy = x[1] - x[0], y[1] - y[0]
    gx, gy = np.meshgrid(x, y)
    cx = np.stack([gx + dx / 2, gx + dx / 2, gx - dx / 2, gx - dx / 2], axis=-1).reshape(-1, 4)
    cy = np.stack([gy - dy / 2, gy + dy / 2, gy + dy / 2, gy - dy / 2], axis=-1).reshape(-1, 4)
    return np.stack([cx, cy], axis=-1)


@pytest.mark.parametrize("seed", range(4))
def test_quadrilateral_cells_against_the_rectangular_clipper_and_under_affine_maps(seed):
    rng = np.random.default_rng(seed)
    X, Y = int(rng.integers(3, 40)), int(rng.integers(3, 40))
    x, y = -3.0 + 0.5 * np.arange(X), 41.0 + 0.25 * np.arange(Y)
    box = (x[0] - 0.25, y[0] - 0.125, x[-1] + 0.25, y[-1] + 0.125)
    shapes = gis.random_star_polygons(7, box, seed=seed) + gis.random_tessellation(5, box, seed=seed)
    hole = np.array([[x[1], y[1]], [x[-2], y[1]], [x[-2], y[-2]], [x[1], y[-2]]])
    inner = hole.mean(0) + 0.3 * (hole - hole.mean(0))
    shapes.append(dict(exterior=hole, holes=[inner]))
    ref = gis.compute_indicatormatrix(x, y, shapes)
    quads = _boxes(x, y)
    got = _quads_matrix(shapes, quads)
    assert np.abs((got - ref)).max() < 1e-12 and got.nnz == ref.nnz
    # area ratios do not change under an affine map of everything; windings may flip (negative determinant)
    A = rng.normal(size=(2, 2)) + 2.0 * np.eye(2) * rng.choice([-1, 1])
    t = rng.normal(size=2) * 100

    def mapped(s):
        if isinstance(s, dict):
            return dict(exterior=s["exterior"] @ A.T + t, holes=[h @ A.T + t for h in s["holes"]])
        return np.asarray(s) @ A.T + t

    got2 = _quads_matrix([mapped(s) for s in shapes], quads @ A.T + t)
    assert np.abs((got2 - ref)).max() < 1e-10


@pytest.mark.parametrize("code", [3035, 32632, 3857])
def test_shapes_in_a_projected_crs_conserve_area(code):
    from atlite_amd import Cutout, Dataset

    x, y = np.arange(5.0, 12.01, 0.25), np.arange(47.0, 53.01, 0.25)
    ring = np.array([[6.1, 48.2], [10.7, 47.9], [11.2, 51.8], [8.0, 52.6], [5.6, 50.3]])
    hole = np.array([[8.0, 49.5], [9.0, 49.5], [9.0, 50.5], [8.0, 50.5]])
    outside = np.array([[11.0, 52.0], [14.0, 52.0], [14.0, 55.0], [11.0, 55.0]])  # partly beside the grid
    proj = lambda r: np.stack(crs.forward(code, r[:, 0], r[:, 1]), axis=1)  # noqa: E731
    shapes = [dict(exterior=proj(ring), holes=[proj(hole)]), proj(ring), proj(outside)]
    cut = Cutout(Dataset({"height": np.zeros((len(y), len(x)))}, dict(y=y, x=x)))
    M = cut.indicatormatrix(shapes, shapes_crs=code, where="host")
    assert M.shape == (3, len(x) * len(y)) and M.data.min() > 0 and M.data.max() <= 1.0
    q = np.stack(crs.forward(code, *np.moveaxis(_boxes(x, y), -1, 0)), axis=-1)
    cell_area = 0.5 * np.abs(np.sum(q[..., 0] * np.roll(q[..., 1], -1, 1) - np.roll(q[..., 0], -1, 1) * q[..., 1], axis=1))
    shoelace = lambda p: 0.5 * abs(np.sum(p[:, 0] * np.roll(p[:, 1], -1) - np.roll(p[:, 0], -1) * p[:, 1]))  # noqa: E731
    got = M @ cell_area
    assert abs(got[1] - shoelace(proj(ring))) < 1e-11 * got[1]
    assert abs(got[0] - (shoelace(proj(ring)) - shoelace(proj(hole)))) < 1e-11 * got[0]
    assert got[2] < 0.5 * shoelace(proj(outside))  # only the part on the grid counts
    # the same shapes described in the cutout's own crs differ only by the curvature of the edges between their vertices
    M0 = cut.indicatormatrix([dict(exterior=ring, holes=[hole]), ring, outside], where="host")
    assert 0 < np.abs(M - M0).max() < 0.12
    with pytest.raises(NotImplementedError, match="geographic"):
        Cutout(Dataset({"height": np.zeros((len(y), len(x)))}, dict(y=y, x=x)), crs=3035).indicatormatrix(shapes, shapes_crs=32632, where="host")


def test_hostile_cells_and_shapes():
    """Degenerate and non-finite cells take no part, a non-convex cell is refused, shapes with non-finite vertices get an
    empty row, offsets are validated before they are followed."""
    quads = _boxes(np.arange(4.0), np.arange(3.0))
    sq = np.array([[0.2, 0.1], [2.6, 0.1], [2.6, 1.7], [0.2, 1.7]])
    ref = _quads_matrix([sq], quads).toarray()
    q = quads.copy()
    q[5] = q[5][0]  # collapsed to a point
    q[6, 2] = np.nan
    got = _quads_matrix([sq], q).toarray()
    keep = np.ones(12, bool)
    keep[[5, 6]] = False
    assert np.array_equal(got[:, keep], ref[:, keep]) and not got[:, ~keep].any()
    bad = quads.copy()
    bad[3, 1] = bad[3].mean(0) - 0.3 * (bad[3, 1] - bad[3].mean(0))  # a dart: the corner pulled through the centre
    with pytest.raises(ValueError, match="convex"):
        _quads_matrix([sq], bad)
    assert _quads_matrix([np.array([[0.2, 0.1], [np.inf, 0.1], [2.6, 1.7]])], quads).nnz == 0
    lib = _lib.load()
    p = [C.c_void_p() for _ in range(3)]
    ptr = np.array([0, 5, 2], np.int64)  # not monotone
    rc = lib.atl_indicator_polygons_quads(2, ptr.ctypes.data, 2, ptr.ctypes.data, None, sq.ctypes.data, 12, quads.ctypes.data,
                                          *[C.byref(v) for v in p])
    assert rc != 0 and "offsets" in lib.atl_last_error().decode()
