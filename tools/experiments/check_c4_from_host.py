#!/usr/bin/env python3
"""
BASELINE.json configs[3] (pv, 8760 x 800 x 800 fp64, 500 shapes: 314 GB of inputs - more than the
288 GB of HBM) on ONE GPU from host memory through the slab pipeline.  Needs ~320 GB of host RAM:
the script checks the cgroup memory limit first and scales the time axis down if it has to.
Inputs: one synthetic year of a 1095-step shard generated on the device, downloaded and tiled along
time on the host (the point is the pipeline at full footprint, not fresh random numbers).
Checks the result of the first shard against a device-resident run of that shard.
"""
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from atlite_amd import Cutout, Dataset, gis, synthetic  # noqa: E402
from atlite_amd.device import default_context  # noqa: E402


def mem_limit():
    lim = 1 << 62
    for p in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
        try:
            v = open(p).read().strip()
            if v != "max":
                lim = min(lim, int(v))
        except Exception:
            pass
    try:
        avail = int([l for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0].split()[1]) * 1024
        lim = min(lim, avail)
    except Exception:
        pass
    return lim


Y = X = 800
S = Y * X
Tshard = 1095
lim = mem_limit()
per_step = 7 * S * 8
T = min(8760, int(0.6 * lim / per_step) // Tshard * Tshard)
print(f"host memory usable: {lim / 1e9:.0f} GB -> T = {T} steps ({T * per_step / 1e9:.0f} GB of inputs)", flush=True)
if T < Tshard:
    sys.exit("not enough host memory for even one shard")
ctx = default_context()
dev, coords = synthetic.pv_inputs(ctx, Tshard, Y, X)
x, y = coords["x"], coords["y"]
dx, dy = x[1] - x[0], y[1] - y[0]
M = gis.compute_indicatormatrix(x, y, gis.random_tessellation(500, (x[0] - dx / 2, y[0] - dy / 2, x[-1] + dx / 2, y[-1] + dy / 2)))
kw = dict(panel="CSi", orientation={"slope": 30.0, "azimuth": 180.0}, matrix=M, aggregate_time=None)
import pandas as pd  # noqa: E402

time_shard = pd.date_range("2013-01-01", periods=Tshard, freq="h")
ref = Cutout(Dataset(dict(dev), dict(time=time_shard, y=y, x=x))).pv(**kw).values  # device-resident shard
t0 = time.perf_counter()
host = {}
for k, v in dev.items():
    a = np.empty((T, Y, X))
    blk = v.numpy().reshape(Tshard, Y, X)
    for r in range(T // Tshard):
        a[r * Tshard:(r + 1) * Tshard] = blk
    host[k] = a
del dev
print(f"host cube built in {time.perf_counter() - t0:.1f} s", flush=True)
tt = pd.date_range("2013-01-01", periods=T, freq="h")
os.environ["ATLITE_HIP_STREAM"] = "1"
ds = Dataset(host, dict(time=tt, y=y, x=x))
t0 = time.perf_counter()
ds.pin()
t_pin = time.perf_counter() - t0
for rep in range(2):
    t0 = time.perf_counter()
    out = Cutout(ds).pv(**kw).values
    dt = time.perf_counter() - t0
    print(f"pv {T}x{Y}x{X}, 500 shapes, from pinned host memory (pin {t_pin:.1f} s): {dt:.2f} s = {T * per_step / dt / 1e9:.1f} GB/s "
          f"= {T * S / dt:.3e} cell-timesteps/s", flush=True)
# the tiled cube repeats the shard's radiation but the solar geometry is stored, so shard 0 must match
ok = np.array_equal(out[:, :Tshard] if out.shape[1] == T else out[:Tshard], ref)
print("first shard == device-resident run of that shard:", ok)
sys.exit(0 if ok else 1)
