"""
Synthetic ERA5-shaped cutouts generated directly in HBM (bench / test tooling).

Grid: ``x = -25 + (70/X) i``, ``y = 30 + (42/Y) j`` (cell centres, degrees, ascending);
time: hourly from ``start``.  Solar altitude/azimuth follow the almanac algorithm with the
ERA5 -30 min shift (atlite/datasets/era5.py:178-188), radiation and temperature are physically
consistent with them, noise is a stateless splitmix64 hash of (seed, variable, linear index).
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import pandas as pd

from . import _lib, solar
from ._lib import check

PV_VARS = ("influx_direct", "influx_diffuse", "influx_toa", "albedo", "temperature", "solar_altitude",
           "solar_azimuth")


def grid_coords(Y, X):
    x = -25.0 + (70.0 / X) * np.arange(X)
    y = 30.0 + (42.0 / Y) * np.arange(Y)
    return x, y


def time_index(T, start="2013-01-01", offset_hours=0):
    return pd.date_range(pd.Timestamp(start) + pd.Timedelta(hours=int(offset_hours)), periods=T, freq="h")


def pv_inputs(ctx, T, Y, X, start="2013-01-01", offset_hours=0, seed=42, interleaved=False):
    """dict name -> DeviceArray (T, Y*X) with the 7 variables convert_pv reads; plus coords.  ``interleaved``: the seven
    cubes slot-interleaved in one allocation (``device.SlotPool``, the layout of the library's own device copies)
    instead of an allocation each."""
    x, y = grid_coords(Y, X)
    t = time_index(T, start, offset_hours)
    h, dec = solar.hour_angle(t, x, "-30min")
    doy = np.asarray(t.dayofyear, dtype=np.float64)
    hour = np.asarray(t.hour, dtype=np.float64)
    tseason = 283.15 + 12.0 * np.sin(2 * np.pi * (doy - 110.0) / 365.0) + 5.0 * np.sin(2 * np.pi * (hour - 9.0) / 24.0)
    S = Y * X
    tabs = dict(
        sin_dec=ctx.upload(np.sin(dec)),
        cos_dec=ctx.upload(np.cos(dec)),
        h=ctx.upload(h),
        lat=ctx.upload(np.radians(y)),
        tseason=ctx.upload(tseason),
    )
    ld = S
    if interleaved:
        from .device import SlotPool, pitch_for

        pool = SlotPool(ctx, T, S, PV_VARS, pitch_for(S))
        out, ld = {k: pool.view(k) for k in PV_VARS}, pool.ld
    else:
        out = {k: ctx.empty((T, S)) for k in PV_VARS}
    # the hash is indexed by the GLOBAL linear index so time shards of one cutout are consistent
    s = _lib.SynthSolar(tabs["sin_dec"].ptr, tabs["cos_dec"].ptr, tabs["h"].ptr, tabs["lat"].ptr,
                        tabs["tseason"].ptr, X, Y, int(seed) + 1000003 * int(offset_hours), 0 if ld == S else ld)
    check(ctx.lib.atl_synth_pv_inputs(ctx.handle, C.byref(s), T, S, *[out[k].ptr for k in PV_VARS]))
    ctx.sync()
    return out, dict(x=x, y=y, time=t)


def wind_inputs(ctx, T, Y, X, seed=42, static_roughness=False):
    S = Y * X
    wnd = ctx.synth_field(_lib.SYN_RAYLEIGH, seed, 5, 8.0, 0.0, T, S)
    rough = ctx.synth_field(_lib.SYN_EXPLOG, seed, 6, 1e-3, 1.5e3, T, S, per_cell_static=static_roughness)
    ctx.sync()
    return dict(wnd100m=wnd, roughness=rough)


def heat_runoff_inputs(ctx, T, Y, X, seed=42):
    S = Y * X
    temp = ctx.synth_field(_lib.SYN_UNIFORM, seed, 4, 263.15, 303.15, T, S)
    runoff = ctx.synth_field(_lib.SYN_NEGLOG, seed, 7, 1e-4, 0.0, T, S)
    height = ctx.synth_field(_lib.SYN_UNIFORM, seed, 8, 0.0, 2000.0, 1, S, per_cell_static=True)
    ctx.sync()
    return dict(temperature=temp, runoff=runoff, height=height)
