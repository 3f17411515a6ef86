#!/usr/bin/env python3
"""Where the host time of a warm `cutout.pv(..., shapes=polys)` call goes (cProfile, C2 shape, device-resident data)."""
import cProfile
import pstats
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from atlite_amd import Cutout, Dataset, gis, synthetic  # noqa: E402
from atlite_amd.device import default_context  # noqa: E402

ctx = default_context()
T, Y, X, N = 8760, 200, 200, 100
inputs, _ = synthetic.pv_inputs(ctx, T, Y, X)
x, y = synthetic.grid_coords(Y, X)
dx, dy = x[1] - x[0], y[1] - y[0]
polys = gis.random_tessellation(N, (x[0] - dx / 2, y[0] - dy / 2, x[-1] + dx / 2, y[-1] + dy / 2), seed=42)
kw = dict(panel="CSi", orientation={"slope": 30.0, "azimuth": 180.0}, shapes=polys, aggregate_time=None)
# ---- the COLD call (what bench.py's api_e2e_ms.cold times): a fresh Cutout over device-resident data, first pv() with shapes ----
for rep in range(2):
    cut = Cutout(Dataset(dict(inputs), dict(time=synthetic.time_index(T), y=y, x=x), static=True))
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    r = cut.pv(**kw)
    pr.disable()
    print("cold call %d: %.2f ms" % (rep, (time.perf_counter() - t0) * 1e3))
    pstats.Stats(pr).sort_stats("cumulative").print_stats(32)
    del cut
cut = Cutout(Dataset(dict(inputs), dict(time=synthetic.time_index(T), y=y, x=x), static=True))
for _ in range(3):
    r = cut.pv(**kw)
t0 = time.perf_counter()
for _ in range(10):
    r = cut.pv(**kw)
print("warm call: %.2f ms" % ((time.perf_counter() - t0) / 10 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    r = cut.pv(**kw)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
