#!/usr/bin/env python3
"""End-to-end rate for a HOST-resident cutout: whole-variable upload + launch vs the slab pipeline
(pinned host memory, copy stream overlapped with the kernels).  pv, T x 200 x 200, 100 shapes."""
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from atlite_amd import Cutout, Dataset, gis, synthetic  # noqa: E402
from atlite_amd.device import default_context  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 2190
Y = X = 200
ctx = default_context()
dev, coords = synthetic.pv_inputs(ctx, T, Y, X)
host = {k: v.numpy() for k, v in dev.items()}
del dev
x, y = coords["x"], coords["y"]
dx, dy = x[1] - x[0], y[1] - y[0]
M = gis.compute_indicatormatrix(x, y, gis.random_tessellation(100, (x[0] - dx / 2, y[0] - dy / 2, x[-1] + dx / 2, y[-1] + dy / 2)))
nbytes = sum(a.nbytes for a in host.values())
kw = dict(panel="CSi", orientation={"slope": 30.0, "azimuth": 180.0}, matrix=M, aggregate_time=None)
res = {}
for mode in ("0", "1", "1", "pinned"):
    os.environ["ATLITE_HIP_STREAM"] = "1" if mode == "pinned" else mode
    c = Cutout(Dataset(host, dict(time=coords["time"], y=y, x=x)))  # fresh dataset: no cached uploads
    if mode == "pinned":
        c.data.pin()
    t0 = time.perf_counter()
    r = c.pv(**kw).values
    dt = time.perf_counter() - t0
    label = {"0": "whole-upload", "1": "slab-pipeline", "pinned": "slab+Dataset.pin"}[mode]
    print(f"{label:14s} {dt:7.3f} s  {nbytes / dt / 1e9:7.2f} GB/s end-to-end  {T * Y * X / dt:.3e} cell-steps/s", flush=True)
    res.setdefault(label, r)
print("identical:", np.array_equal(res["whole-upload"], res["slab-pipeline"]), f"({nbytes / 1e9:.1f} GB of inputs)")
