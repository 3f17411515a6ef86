#!/usr/bin/env python3
"""Indicator matrix (area(shape n cell) / area(cell)): host polygon clipper against the device kernel.
usage: tools/bench_indicator.py   (on the GPU box)"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from atlite_amd import gis, synthetic  # noqa: E402
from atlite_amd.device import Context  # noqa: E402

ctx = Context(0)
for (Y, X, N, kind) in ((200, 200, 100, "tessellation"), (200, 200, 100, "star"), (800, 800, 500, "tessellation"), (800, 800, 500, "star")):
    x, y = synthetic.grid_coords(Y, X)
    dx, dy = x[1] - x[0], y[1] - y[0]
    box = (x[0] - dx / 2, y[0] - dy / 2, x[-1] + dx / 2, y[-1] + dy / 2)
    polys = gis.random_tessellation(N, box, seed=42) if kind == "tessellation" else gis.random_star_polygons(N, box, seed=1)
    th, td = [], []
    for _ in range(3):
        t0 = time.perf_counter(); H = gis.compute_indicatormatrix(x, y, polys); th.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); D = gis.compute_indicatormatrix(x, y, polys, ctx=ctx); td.append(time.perf_counter() - t0)
    err = abs(H - D).max() if H.nnz else 0.0
    print(f"{Y}x{X} {N:4d} {kind:12s} nnz {H.nnz:7d}/{D.nnz:7d}  host {min(th) * 1e3:8.1f} ms  device {min(td) * 1e3:8.1f} ms  max|diff| {err:.2e}", flush=True)
