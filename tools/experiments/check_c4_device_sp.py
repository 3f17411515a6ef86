#!/usr/bin/env python3
"""
BASELINE.json configs[3] at FULL size, device-resident, on ONE MI355X: pv over 8760 x 800 x 800 fp64
with 500 shapes.  The seven stored-angle cubes would be 314 GB (> 288 GB of HBM); with the in-kernel
solar position the five radiation / albedo / temperature cubes are 224 GB and fit.  The cubes are
generated shard by shard straight into place (the generator's two solar-angle outputs go to a reusable
shard-sized scratch).  Checks the first shard against the stored-angle kernel and prints the time of
the whole-year launch.
"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from atlite_amd import _lib, gis, solar, synthetic  # noqa: E402
from atlite_amd._lib import check  # noqa: E402
from atlite_amd.device import Context  # noqa: E402

CSI = dict(c_temp_amb=1, c_temp_irrad=0.035, r_tmod=298, r_irradiance=1000, k_1=-0.017162, k_2=-0.040289,
           k_3=-0.004681, k_4=0.000148, k_5=0.000169, k_6=0.000005, inverter_efficiency=0.9)
T, Y, X, N, TS = 8760, 800, 800, 500, 1095
S = Y * X
ctx = Context(0)
x, y = synthetic.grid_coords(Y, X)
five = [k for k in synthetic.PV_VARS if not k.startswith("solar_")]
big = {k: ctx.empty((T, S)) for k in five}  # 5 x 44.8 GB
alt, az = ctx.empty((TS, S)), ctx.empty((TS, S))
first = {}
for r in range(T // TS):
    off = r * TS
    t = synthetic.time_index(TS, "2013-01-01", off)
    h, dec = solar.hour_angle(t, x, "-30min")
    doy, hour = np.asarray(t.dayofyear, float), np.asarray(t.hour, float)
    tseason = 283.15 + 12.0 * np.sin(2 * np.pi * (doy - 110.0) / 365.0) + 5.0 * np.sin(2 * np.pi * (hour - 9.0) / 24.0)
    tabs = [ctx.upload(a) for a in (np.sin(dec), np.cos(dec), h, np.radians(y), tseason)]
    s = _lib.SynthSolar(*[a.ptr for a in tabs], X, Y, 42 + 1000003 * off)
    ptrs = [big[k].ptr + off * S * 8 for k in five] + [alt.ptr, az.ptr]
    check(ctx.lib.atl_synth_pv_inputs(ctx.handle, C.byref(s), TS, S, *ptrs))
    ctx.sync()
    if r == 0:  # keep the first shard's stored angles for the cross-check
        first = dict(solar_altitude=alt, solar_azimuth=az)
        alt, az = ctx.empty((TS, S)), ctx.empty((TS, S))
print(f"generated {5 * T * S * 8 / 1e9:.0f} GB of inputs in HBM", flush=True)
dx, dy = x[1] - x[0], y[1] - y[0]
M = gis.compute_indicatormatrix(x, y, gis.random_tessellation(N, (x[0] - dx / 2, y[0] - dy / 2, x[-1] + dx / 2, y[-1] + dy / 2)))
plan = ctx.plan(M, row_len=X)
time_all = synthetic.time_index(T)
h, dec = solar.hour_angle(time_all, x, "-30min")
lat = np.radians(y)
tables = dict(sin_dec=np.sin(dec), cos_dec=np.cos(dec), h=h, cos_h=np.cos(h), sin_lat=np.sin(lat), cos_lat=np.cos(lat))
tables = {k: ctx.upload(np.ascontiguousarray(v)) for k, v in tables.items()}
scal = dict(CSI, slope=np.radians(30.0), azimuth=np.radians(180.0))
ctx.set_profiling(True)
outs = {}
for skip in (False, True):
    ms = []
    for _ in range(4):
        out = ctx.pv(big, scal, T, S, plan=plan, solar_tables=tables, options=dict(night_skip=skip))
        ms.append(ctx.last_kernel_ms())
    outs[skip] = out.numpy()
    k = float(np.median(ms[1:]))
    print(f"pv {T}x{Y}x{X}, {N} shapes, device-resident, in-kernel solar position, night early-out {'on' if skip else 'off'}: "
          f"fused kernel {k:.2f} ms = {40 * T * S / k / 1e6:.0f} GB/s of 40 B/cell = {T * S / k * 1e3:.3e} cell-timesteps/s", flush=True)
print("night early-out on == off (bits):", bool(np.array_equal(outs[False], outs[True])))
out = outs[False]
shard = {kk: big[kk].slab(0, TS) for kk in five}
shard.update(first)
ref = ctx.pv(shard, scal, TS, S, plan=plan, options=dict(night_skip=False)).numpy()
err = np.abs(out[:, :TS] - ref).max() / np.abs(ref).max()
print(f"first shard vs the stored-angle kernel: max |diff| / max = {err:.2e}")
sys.exit(0 if err < 1e-10 else 1)
