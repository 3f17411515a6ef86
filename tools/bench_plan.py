#!/usr/bin/env python3
"""Cold-start pieces of a `Cutout.pv(shapes=...)` call: indicator matrix (host / device) and plan build."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from atlite_amd import gis, synthetic  # noqa: E402
from atlite_amd.device import Context  # noqa: E402

ctx = Context(0)
for (Y, X, N, kind) in ((200, 200, 100, "tessellation"), (800, 800, 500, "tessellation"), (800, 800, 500, "star"), (400, 400, 50, "tessellation")):
    x, y = synthetic.grid_coords(Y, X)
    dx, dy = x[1] - x[0], y[1] - y[0]
    box = (x[0] - dx / 2, y[0] - dy / 2, x[-1] + dx / 2, y[-1] + dy / 2)
    polys = gis.random_tessellation(N, box, seed=42) if kind == "tessellation" else gis.random_star_polygons(N, box, seed=1)
    t0 = time.perf_counter(); M = gis.compute_indicatormatrix(x, y, polys, ctx=ctx); t_ind = time.perf_counter() - t0
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); plan = ctx.plan(M, row_len=X, cache=False) if "cache" in ctx.plan.__code__.co_varnames else ctx.plan(M * (1.0 + 1e-9 * len(ts)), row_len=X); ctx.sync(); ts.append(time.perf_counter() - t0)
    info = plan.info()
    print(f"{Y}x{X} {N:4d} {kind:12s} nnz {M.nnz:7d} P {info['n_partial_rows']:6d}: indicator (device) {t_ind * 1e3:7.1f} ms, plan build {min(ts) * 1e3:7.1f} ms", flush=True)
