#!/usr/bin/env python3
"""
Secondary measurements (not the bench.py line): the other BASELINE.json configs at their
single-GPU sizes, dominant-kernel time from HIP events, GB/s on algorithmic bytes.
  C3  wind V112 per-cell series      8760x400x400           24 B/cell-step (16 in + 8 out)
  C3m wind V112 per-cell time-mean   8760x400x400           16 B
  C3a wind V112 aggregated, 100 shp  8760x400x400           16 B
  C5h heat demand, 50 shapes         4380x400x400 (1/8 of the 35040-step config)  8 B
  C5r runoff, 50 shapes              4380x400x400           8 B
  C4s pv, 500 shapes                 1095x800x800 (1/8 of the 8760-step config)  56 B
"""
import json
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from atlite_amd import gis, synthetic  # noqa: E402
from atlite_amd.device import Context  # noqa: E402
from atlite_amd.device import interleave_enabled  # noqa: E402

V = np.array([0, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 25, 25], dtype=float)
POW = np.array([0.000, 0.000, 0.005, 0.150, 0.300, 0.525, 0.905, 1.375, 1.950, 2.580, 2.960, 3.050, 3.060, 3.060, 0.000])
CSI = dict(c_temp_amb=1, c_temp_irrad=0.035, r_tmod=298, r_irradiance=1000, k_1=-0.017162, k_2=-0.040289,
           k_3=-0.004681, k_4=0.000148, k_5=0.000169, k_6=0.000005, inverter_efficiency=0.9,
           slope=np.radians(30.0), azimuth=np.radians(180.0))


WARM = int(os.environ.get("ATL_CFG_WARMUP", "10"))  # untimed launches first: the shader clock needs ~30 ms of load to settle
REPS = int(os.environ.get("ATL_CFG_REPS", "0"))     # (profiles/r03_clock_per_launch.txt); ATL_CFG_REPS overrides a caller's reps


def timed(ctx, fn, reps=6):
    ctx.set_profiling(True)
    ms = []
    reps = REPS or reps
    for i in range(reps + WARM):
        out = fn()
        t = ctx.last_kernel_ms()
        if i >= WARM:
            ms.append(t)
        del out
    return float(np.median(ms)), float(np.min(ms))


def shapes_matrix(Y, X, n):
    x, y = synthetic.grid_coords(Y, X)
    dx, dy = x[1] - x[0], y[1] - y[0]
    polys = gis.random_tessellation(n, (x[0] - dx / 2, y[0] - dy / 2, x[-1] + dx / 2, y[-1] + dy / 2), seed=42)
    return gis.compute_indicatormatrix(x, y, polys)


def main():
    which = sys.argv[1:] or ["C3", "C3m", "C3a", "C5h", "C5r", "C4s"]
    ctx = Context(0)
    res = {}

    def report(name, bytes_per, cells, med, mn, extra=""):
        gbs = bytes_per * cells / (med * 1e-3) / 1e9
        res[name] = dict(kernel_ms_median=med, kernel_ms_min=mn, GBps=gbs, frac_of_8TBps=gbs / 8000, cells=cells,
                         bytes_per_cell_step=bytes_per)
        print(f"{name:4s} {extra:40s} median {med:8.3f} ms  min {mn:8.3f} ms  {gbs:7.0f} GB/s  "
              f"{100 * gbs / 8000:5.1f}% of 8 TB/s  {cells / (med * 1e-3):.3e} cell-steps/s", flush=True)

    if any(w.startswith("C3") for w in which):
        T, Y, X = 8760, 400, 400
        S = Y * X
        d = synthetic.wind_inputs(ctx, T, Y, X)
        args = (d["wnd100m"], d["roughness"], V, POW / 3.06, 80.0, 100.0, "logarithmic", T, S)
        if "C3" in which:
            report("C3", 24, T * S, *timed(ctx, lambda: ctx.wind(*args)), "wind per-cell series")
        if "C3m" in which:
            report("C3m", 16, T * S, *timed(ctx, lambda: ctx.wind(*args, time_agg="mean")), "wind per-cell time-mean")
        if "C3a" in which:
            plan = ctx.plan(shapes_matrix(Y, X, 100), row_len=X)
            report("C3a", 16, T * S, *timed(ctx, lambda: ctx.wind(*args, plan=plan)),
                   f"wind aggregated 100 shapes {plan.info()['tile_w']}x{plan.info()['tile_h']} P={plan.info()['n_partial_rows']}")
        del d, args
    if any(w.startswith("C5") for w in which):
        T, Y, X = 4380, 400, 400
        S = Y * X
        d = synthetic.heat_runoff_inputs(ctx, T, Y, X)
        plan = ctx.plan(shapes_matrix(Y, X, 50), row_len=X)
        info = plan.info()
        tag = f"50 shapes {info['tile_w']}x{info['tile_h']} P={info['n_partial_rows']}"
        day_ptr = np.arange(0, T + 1, 24)
        if day_ptr[-1] != T:
            day_ptr = np.append(day_ptr, T)
        if "C5h" in which:
            report("C5h", 8, T * S, *timed(ctx, lambda: ctx.heat_demand(d["temperature"], day_ptr, 288.15, 1.0, 0.0, T, S, plan=plan)),
                   "heat demand " + tag)
        if "C5r" in which:
            report("C5r", 8, T * S, *timed(ctx, lambda: ctx.runoff(d["runoff"], d["height"], T, S, plan=plan)),
                   "runoff " + tag)
        del d
    if "C4s" in which:
        T, Y, X = 1095, 800, 800
        S = Y * X
        inputs, _ = synthetic.pv_inputs(ctx, T, Y, X, interleaved=interleave_enabled())
        plan = ctx.plan(shapes_matrix(Y, X, 500), row_len=X)
        info = plan.info()
        report("C4s", 56, T * S, *timed(ctx, lambda: ctx.pv(inputs, CSI, T, S, plan=plan, options=dict(night_skip=False))),
               f"pv 500 shapes {info['tile_w']}x{info['tile_h']} P={info['n_partial_rows']}")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
