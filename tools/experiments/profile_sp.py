#!/usr/bin/env python3
"""In-kernel solar position on the C2 shape, night early-out on and off (a target for rocprofv3 / tools/pmc_gpu.sh)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from atlite_amd import gis, synthetic, solar
from atlite_amd.device import Context
from tools.bench_configs import CSI
ctx = Context(0)
T, Y, X, N = 8760, 200, 200, 100
S = Y * X
inputs, _ = synthetic.pv_inputs(ctx, T, Y, X)
x, y = synthetic.grid_coords(Y, X)
dx, dy = x[1] - x[0], y[1] - y[0]
M = gis.compute_indicatormatrix(x, y, gis.random_tessellation(N, (x[0] - dx / 2, y[0] - dy / 2, x[-1] + dx / 2, y[-1] + dy / 2), seed=42))
plan = ctx.plan(M, row_len=X)
five = {k: v for k, v in inputs.items() if not k.startswith("solar_")}
t = synthetic.time_index(T)
h, dec = solar.hour_angle(t, x, "-30min")
lat = np.radians(y)
tables = dict(sin_dec=np.sin(dec), cos_dec=np.cos(dec), h=h, cos_h=np.cos(h), sin_lat=np.sin(lat), cos_lat=np.cos(lat))
for skip in (True, False):
    for _ in range(4):
        out = ctx.pv(five, CSI, T, S, plan=plan, solar_tables=tables, options=dict(night_skip=skip))
    ctx.sync()
