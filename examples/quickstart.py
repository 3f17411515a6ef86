#!/usr/bin/env python3
"""
Quickstart: the atlite calls you know, on an MI355X.

Builds a small synthetic ERA5-shaped cutout in host memory, then runs pv / wind / heat demand /
runoff with shapes, layouts and the usual keyword arguments.  Needs the library
(`python -c "import __graft_entry__ as g; g.build()"`) and a gfx950 GPU.
"""
import sys
from pathlib import Path

import numpy as np
import pandas as pd

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from atlite_amd import Cutout, Dataset, LabeledArray, gis, solar  # noqa: E402

T, Y, X = 24 * 14, 40, 60
time = pd.date_range("2013-06-01", periods=T, freq="h")
x = np.linspace(-10.0, 20.0, X)
y = np.linspace(36.0, 60.0, Y)
rng = np.random.default_rng(0)

# solar angles with ERA5's -30 min shift, consistent radiation, temperature, wind, runoff
h, dec = solar.hour_angle(time, x, "-30min")
lat = np.radians(y)[None, :, None]
s = np.clip(np.sin(dec)[:, None, None] * np.sin(lat) + np.cos(dec)[:, None, None] * np.cos(lat) * np.cos(h)[:, None, :], -1, 1)
alt = np.arcsin(s)
az = np.arccos(np.clip((np.sin(dec)[:, None, None] * np.cos(lat) - np.cos(dec)[:, None, None] * np.sin(lat) * np.cos(h)[:, None, :]) / np.cos(alt), -1, 1))
az = np.where(h[:, None, :] <= 0, az, 2 * np.pi - az)
toa = 1361.0 * np.maximum(s, 0)
kt, fd = 0.2 + 0.55 * rng.random((T, Y, X)), 0.3 + 0.5 * rng.random((T, Y, X))
ds = Dataset(
    dict(influx_toa=toa, influx_direct=toa * kt * fd, influx_diffuse=toa * kt * (1 - fd),
         albedo=0.05 + 0.3 * rng.random((T, Y, X)), temperature=285 + 8 * rng.standard_normal((T, Y, X)),
         solar_altitude=alt, solar_azimuth=az,
         wnd100m=8 * np.sqrt(-np.log1p(-rng.random((T, Y, X)))) * 2 / np.sqrt(np.pi),
         roughness=np.exp(np.log(1e-3) + rng.random((T, Y, X)) * np.log(1.5e3)),
         runoff=-1e-4 * np.log1p(-rng.random((T, Y, X))), height=2000 * rng.random((Y, X))),
    coords=dict(time=time, y=y, x=x), chunked=True)   # chunked=True: results as (time, <index>) like a file-loaded cutout
cutout = Cutout(ds)
print(cutout)

# 12 "bus regions" tiling the domain (any polygons work: vertex arrays, dict(exterior=, holes=), shapely)
regions = pd.Series(gis.random_tessellation(12, cutout.bounds[[0, 1, 2, 3]], seed=1), index=pd.Index([f"bus{i}" for i in range(12)], name="bus"))

pv = cutout.pv(panel="CSi", orientation="latitude_optimal", shapes=regions, per_unit=True, aggregate_time=None)
print("pv      ", pv.dims, pv.shape, pv.attrs, "mean CF", float(np.mean(pv.values)))

layout = LabeledArray(rng.random((Y, X)) * 10, ("y", "x"), {"y": y, "x": x})   # MW installed per cell
wind, cap = cutout.wind(turbine="Vestas_V112_3MW", shapes=regions, layout=layout, return_capacity=True, aggregate_time="mean")
print("wind    ", wind.dims, wind.shape, wind.attrs, "capacity", cap.values.round(1)[:4], "...")

cf = cutout.wind(turbine="Vestas_V112_3MW", smooth=True, aggregate_time="mean")        # per-cell capacity factor map
print("wind cf ", cf.dims, cf.shape, "max", float(cf.values.max()))

hd = cutout.heat_demand(threshold=15.0, hour_shift=1.0, shapes=regions, aggregate_time=None)
print("heat    ", hd.dims, hd.shape, "days", pd.DatetimeIndex(hd.coords["time"])[[0, -1]].strftime("%Y-%m-%d").tolist())

ro = cutout.runoff(shapes=regions, smooth=True, aggregate_time=None)
print("runoff  ", ro.dims, ro.shape)

irr = cutout.irradiation(orientation={"slope": 30.0, "azimuth": 180.0}, tracking="horizontal", aggregate_time="mean")
print("irradiation (1-axis tracking) mean W/m2", float(irr.values.mean()))

# the same calls straight from a prepared NetCDF-4 cutout file: parsed natively (no xarray / netCDF4 /
# libhdf5), chunks inflated on host threads, un-shuffled and widened to fp64 on the GPU, slab by slab
nc = Path(__file__).resolve().parent.parent / "tests" / "golden" / "nc" / "cutout_small_f32.nc"
filecut = Cutout(nc)
print(filecut.data)
few = pd.Series(gis.random_tessellation(3, filecut.bounds[[0, 1, 2, 3]], seed=2), index=pd.Index(["a", "b", "c"], name="bus"))
pv_f = filecut.pv(panel="CSi", orientation={"slope": 30.0, "azimuth": 180.0}, shapes=few, aggregate_time=None)
print("pv(file)", pv_f.dims, pv_f.shape)
