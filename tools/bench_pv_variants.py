#!/usr/bin/env python3
"""pv kernel variants on the C2 shape (8760x200x200, 100 shapes): dominant-kernel time from HIP events."""
import os

os.environ["ATLITE_HIP_NIGHT_SKIP"] = "0"  # lines without an explicit night_skip read every byte
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from atlite_amd import gis, solar, synthetic  # noqa: E402
from atlite_amd.device import Context  # noqa: E402
from atlite_amd.device import interleave_enabled  # noqa: E402

CSI = dict(c_temp_amb=1, c_temp_irrad=0.035, r_tmod=298, r_irradiance=1000, k_1=-0.017162, k_2=-0.040289,
           k_3=-0.004681, k_4=0.000148, k_5=0.000169, k_6=0.000005, inverter_efficiency=0.9)
T, Y, X, N = 8760, 200, 200, 100
ctx = Context(0)
inputs, coords = synthetic.pv_inputs(ctx, T, Y, X, interleaved=interleave_enabled())
x, y = coords["x"], coords["y"]
dx, dy = x[1] - x[0], y[1] - y[0]
M = gis.compute_indicatormatrix(x, y, gis.random_tessellation(N, (x[0] - dx / 2, y[0] - dy / 2, x[-1] + dx / 2, y[-1] + dy / 2)))
plan = ctx.plan(M, row_len=X)
S = Y * X
h, dec = solar.hour_angle(coords["time"], x, "-30min")
lat = np.radians(y)
tables = {k: ctx.upload(np.ascontiguousarray(v)) for k, v in dict(sin_dec=np.sin(dec), cos_dec=np.cos(dec), h=h,
                                                                   cos_h=np.cos(h), sin_lat=np.sin(lat), cos_lat=np.cos(lat)).items()}
scal = dict(CSI, slope=np.radians(30.0), azimuth=np.radians(180.0))
alat = np.abs(lat)
slope = np.where(alat <= np.radians(25), 0.87 * alat, np.where(alat <= np.radians(50), 0.76 * alat + np.radians(0.31), np.radians(40.0)))
percell = dict(CSI, slope=ctx.upload(np.repeat(slope, X)), azimuth=ctx.upload(np.full(S, np.pi)))
five = {k: v for k, v in inputs.items() if not k.startswith("solar_")}
from atlite_amd.resource import get_solarpanelconfig  # noqa: E402

kanena = dict(get_solarpanelconfig("KANENA"), slope=np.radians(30.0), azimuth=np.radians(180.0))
# an `influx` / `outflux` flavoured dataset over the same cubes (the values only need to be plausible here)
influx_ds = dict(influx=inputs["influx_direct"], influx_toa=inputs["influx_toa"], outflux=inputs["influx_diffuse"],
                 temperature=inputs["temperature"], solar_altitude=inputs["solar_altitude"], solar_azimuth=inputs["solar_azimuth"])


FILTER = [f for f in os.environ.get("ATL_VARIANTS", "").split("|") if f]  # substrings of the lines to run (default: all)
REPS = int(os.environ.get("ATL_VARIANT_REPS", "5"))


WARM = int(os.environ.get("ATL_VARIANT_WARMUP", "10"))  # launches before the timed ones: the shader clock needs ~30 ms of load
                                                        # to settle (profiles/r03_clock_per_launch.txt)


def timed(fn, reps=5):
    ctx.set_profiling(True)
    ms = []
    for i in range(reps + WARM):
        out = fn()
        t = ctx.last_kernel_ms()
        if i >= WARM:
            ms.append(t)
    return float(np.median(ms)), out


ref = None
for name, nbytes, fn in (
    ("getter, scalar orientation (bench.py)", 56, lambda: ctx.pv(inputs, scal, T, S, plan=plan, options=dict(night_skip=False))),
    ("getter + night early-out", 56, lambda: ctx.pv(inputs, scal, T, S, plan=plan, options=dict(night_skip=True))),
    ("getter, per-cell orientation (latitude_optimal)", 56, lambda: ctx.pv(inputs, percell, T, S, plan=plan, options=dict(night_skip=False))),
    ("in-kernel solar position (5 cubes + tables)", 40, lambda: ctx.pv(five, scal, T, S, plan=plan, solar_tables=tables)),
    ("in-kernel solar position + night early-out", 40, lambda: ctx.pv(five, scal, T, S, plan=plan, solar_tables=tables, options=dict(night_skip=True))),
    ("pv(tracking='horizontal') - fast family, closed-form tracker", 56, lambda: ctx.pv(inputs, scal, T, S, plan=plan, options=dict(tracking="horizontal"))),
    ("pv(tracking='tilted_horizontal') - fast family", 56, lambda: ctx.pv(inputs, scal, T, S, plan=plan, options=dict(tracking="tilted_horizontal"))),
    ("pv(tracking='vertical') - fast family (48 B/cell: azimuth not read)", 48, lambda: ctx.pv(inputs, scal, T, S, plan=plan, options=dict(tracking="vertical"))),
    ("pv(tracking='dual') - fast family (48 B/cell: azimuth not read)", 48, lambda: ctx.pv(inputs, scal, T, S, plan=plan, options=dict(tracking="dual"))),
    ("irradiation() - fast family, no panel model (48 B/cell: temperature is not read)", 48,
     lambda: ctx.pv(inputs, scal, T, S, plan=plan, options=dict(panel_model="none"))),
    ("solar_thermal() - fast family, collector tail", 56,
     lambda: ctx.pv(inputs, scal, T, S, plan=plan, options=dict(panel_model="solar_thermal", c0=0.8, c1=3.0, t_store_K=353.15))),
    ("pv(trigon_model='other') - fast family, Hay-Davies tail", 56, lambda: ctx.pv(inputs, scal, T, S, plan=plan, options=dict(trigon_model="other"))),
    ("tracking='horizontal' + Hay-Davies - fast family again since round 6 (general kernel in r03-r05: 4.87 ms)", 56, lambda: ctx.pv(inputs, scal, T, S, plan=plan, options=dict(tracking="horizontal", trigon_model="other"))),
    ("tracking='tilted_horizontal', per-cell orientation - fast family (r02)", 56, lambda: ctx.pv(inputs, percell, T, S, plan=plan, options=dict(tracking="tilted_horizontal"))),
    ("pv(panel='KANENA') bofinger - fast family (r02)", 56, lambda: ctx.pv(inputs, kanena, T, S, plan=plan, options=dict(night_skip=False))),
    ("bofinger + Hay-Davies - fast family (r02; was the general kernel)", 56, lambda: ctx.pv(inputs, kanena, T, S, plan=plan, options=dict(trigon_model="other"))),
    ("irradiation(trigon_model='other') - fast family (r02) (48 B/cell: temperature is not read)", 48, lambda: ctx.pv(inputs, scal, T, S, plan=plan, options=dict(panel_model="none", trigon_model="other"))),
    ("solar_thermal(trigon_model='other') - fast family (r02)", 56,
     lambda: ctx.pv(inputs, scal, T, S, plan=plan, options=dict(panel_model="solar_thermal", c0=0.8, c1=3.0, t_store_K=353.15, trigon_model="other"))),
    ("bofinger + tracking='horizontal' - fast family again since round 6 (general kernel in r03-r05: 5.45 ms)", 56, lambda: ctx.pv(inputs, kanena, T, S, plan=plan, options=dict(tracking="horizontal"))),
    ("irradiation(tracking='dual') - fast family again since round 6 (general kernel in r03-r05: 3.62 ms)", 48, lambda: ctx.pv(inputs, scal, T, S, plan=plan, options=dict(panel_model="none", tracking="dual"))),
    ("bofinger + tracking='tilted_horizontal' + Hay-Davies, per-cell orientation - fast family since round 6, atl_kernels_pvka.hip (general kernel in r03-r05: 6.09 ms)", 56,
     lambda: ctx.pv(inputs, dict(kanena, slope=percell["slope"], azimuth=percell["azimuth"]), T, S, plan=plan, options=dict(tracking="tilted_horizontal", trigon_model="other"))),
    ("irradiation(tracking='horizontal', trigon_model='other') - fast family since round 6, atl_kernels_pvka.hip (48 B/cell)", 48,
     lambda: ctx.pv(inputs, scal, T, S, plan=plan, options=dict(panel_model="none", tracking="horizontal", trigon_model="other", night_skip=False))),
    ("bofinger + tracking='vertical', per-cell orientation - fast family since round 6, atl_kernels_pvkc.hip (48 B/cell: azimuth not read)", 48,
     lambda: ctx.pv(inputs, dict(kanena, slope=percell["slope"], azimuth=percell["azimuth"]), T, S, plan=plan, options=dict(tracking="vertical", night_skip=False))),
    ("pv(tracking='horizontal') + night early-out (r02)", 56, lambda: ctx.pv(inputs, scal, T, S, plan=plan, options=dict(tracking="horizontal", night_skip=True))),
    ("pv(tracking='tilted_horizontal') + night early-out (r02)", 56, lambda: ctx.pv(inputs, scal, T, S, plan=plan, options=dict(tracking="tilted_horizontal", night_skip=True))),
    ("pv(tracking='dual') + night early-out (r02)", 48, lambda: ctx.pv(inputs, scal, T, S, plan=plan, options=dict(tracking="dual", night_skip=True))),
    ("tracking='horizontal' + Hay-Davies, per-cell orientation + night early-out - fast family since round 6, atl_kernels_pvkc.hip (general kernel in r03-r05: 4.93 ms)", 56, lambda: ctx.pv(inputs, percell, T, S, plan=plan, options=dict(tracking="horizontal", trigon_model="other", night_skip=True))),
    ("pv(trigon_model='other') + night early-out (r02)", 56, lambda: ctx.pv(inputs, scal, T, S, plan=plan, options=dict(trigon_model="other", night_skip=True))),
    ("pv(panel='KANENA') + night early-out (r02)", 56, lambda: ctx.pv(inputs, kanena, T, S, plan=plan, options=dict(night_skip=True))),
    ("irradiation() + night early-out (r02)", 48, lambda: ctx.pv(inputs, scal, T, S, plan=plan, options=dict(panel_model="none", night_skip=True))),
    ("solar_thermal() + night early-out (r02)", 56,
     lambda: ctx.pv(inputs, scal, T, S, plan=plan, options=dict(panel_model="solar_thermal", c0=0.8, c1=3.0, t_store_K=353.15, night_skip=True))),
    ("influx / outflux dataset (Reindl split, albedo from outflux) - fast family head (r02; was the general kernel)", 48,
     lambda: ctx.pv(influx_ds, scal, T, S, plan=plan)),
    ("influx / outflux dataset + Hay-Davies - fast family head (r04; was the general kernel)", 48, lambda: ctx.pv(influx_ds, scal, T, S, plan=plan, options=dict(trigon_model="other"))),
    ("influx / outflux dataset, enhanced clearsky model (humidity) - fast family head (r04; was the general kernel)", 56,
     lambda: ctx.pv(dict(influx_ds, humidity=inputs["albedo"]), scal, T, S, plan=plan, options=dict(clearsky_model="enhanced"))),
    ("influx / outflux dataset, enhanced clearsky model + Hay-Davies - fast family head (r04)", 56,
     lambda: ctx.pv(dict(influx_ds, humidity=inputs["albedo"]), scal, T, S, plan=plan, options=dict(clearsky_model="enhanced", trigon_model="other"))),
    ("influx / outflux dataset + Hay-Davies + night early-out (r04)", 48, lambda: ctx.pv(influx_ds, scal, T, S, plan=plan, options=dict(trigon_model="other", night_skip=True))),
    ("influx dataset with an albedo variable + Hay-Davies - fast family head since round 6 (the general kernel before: 5.04 ms)", 48,
     lambda: ctx.pv(dict(influx=inputs["influx_direct"], influx_toa=inputs["influx_toa"], albedo=inputs["albedo"], temperature=inputs["temperature"],
                         solar_altitude=inputs["solar_altitude"], solar_azimuth=inputs["solar_azimuth"]), scal, T, S, plan=plan, options=dict(trigon_model="other"))),
    ("per-cell series out (no matrix), no early-out", 64, lambda: ctx.pv(inputs, scal, T, S, options=dict(night_skip=False))),
    ("per-cell series out (no matrix) + night early-out", 64, lambda: ctx.pv(inputs, scal, T, S, options=dict(night_skip=True, row_len=X))),
    ("per-cell time-mean (capacity factor map), no early-out", 56, lambda: ctx.pv(inputs, scal, T, S, time_agg="mean", options=dict(night_skip=False))),
    ("per-cell time-mean (capacity factor map) + night early-out", 56, lambda: ctx.pv(inputs, scal, T, S, time_agg="mean", options=dict(night_skip=True, row_len=X))),
    ("per-cell time-mean, in-kernel solar position + early-out", 40, lambda: ctx.pv(five, scal, T, S, time_agg="mean", solar_tables=tables, options=dict(night_skip=True))),
):
    if FILTER and not any(f in name for f in FILTER):
        continue
    ms, out = timed(fn, REPS)
    gbs = nbytes * T * S / (ms * 1e-3) / 1e9
    extra = ""
    early_out = "early-out" in name
    if ref is None:
        ref = out.numpy()
    elif "in-kernel" in name and "per-cell" not in name and not FILTER:
        o = out.numpy()
        extra = f"  max rel diff vs getter {np.max(np.abs(o - ref) / np.maximum(np.abs(ref), 1e-12 * ref.max())):.1e}"
    if early_out:  # reads fewer bytes than the kernel it replaces (data dependent: PMC traffic in profiles/): no GB/s claim
        rate = f"effective {gbs:6.0f} GB/s-equivalent of the {nbytes} B/cell it replaces (NOT bytes moved)"
    else:
        rate = f"{gbs:6.0f} GB/s on the {nbytes} B/cell it reads"
    print(f"{name:52s} {ms:8.3f} ms  {rate}  {T * S / (ms * 1e-3):.3e} cell-steps/s{extra}", flush=True)
    del out
