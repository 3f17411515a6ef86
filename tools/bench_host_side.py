#!/usr/bin/env python3
"""Host-side setup costs next to the millisecond kernels: tessellation, indicator matrix, plan."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from atlite_amd import gis, synthetic  # noqa: E402
from atlite_amd.device import Context  # noqa: E402

ctx = Context(0)
for Y, X, N in ((200, 200, 100), (400, 400, 100), (800, 800, 500)):
    x, y = synthetic.grid_coords(Y, X)
    dx, dy = x[1] - x[0], y[1] - y[0]
    t0 = time.perf_counter()
    polys = gis.random_tessellation(N, (x[0] - dx / 2, y[0] - dy / 2, x[-1] + dx / 2, y[-1] + dy / 2))
    t1 = time.perf_counter()
    M = gis.compute_indicatormatrix(x, y, polys)
    t2 = time.perf_counter()
    plan = ctx.plan(M, row_len=X)
    ctx.sync()
    t3 = time.perf_counter()
    info = plan.info()
    print(f"{Y}x{X}, {N} shapes: tessellation {1e3*(t1-t0):7.1f} ms | indicator matrix {1e3*(t2-t1):7.1f} ms (nnz {M.nnz}) | "
          f"plan {1e3*(t3-t2):7.1f} ms (tile {info['tile_w']}x{info['tile_h']}, P={info['n_partial_rows']})", flush=True)
