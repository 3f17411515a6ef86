#!/usr/bin/env python3
"""Per-cell time reductions (capacity-factor maps): blocks per CU of the slot chunking (ATLITE_HIP_CELL_BLOCKS_PER_CU)."""
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from atlite_amd import synthetic  # noqa: E402
from atlite_amd.device import Context  # noqa: E402
from tools.bench_configs import CSI, POW, V, timed  # noqa: E402

ctx = Context(0)
T, Y, X = 8760, 200, 200
S = Y * X
pv, _ = synthetic.pv_inputs(ctx, T, Y, X, interleaved=True)
w = synthetic.wind_inputs(ctx, T, 400, 400)
wargs = (w["wnd100m"], w["roughness"], V, POW / 3.06, 80.0, 100.0, "logarithmic", T, 160000)
for per_cu in (16, 32, 64, 128, 256):
    os.environ["ATLITE_HIP_CELL_BLOCKS_PER_CU"] = str(per_cu)
    a = timed(ctx, lambda: ctx.pv(pv, CSI, T, S, time_agg="mean", options=dict(night_skip=False, row_len=X)), reps=8)[0]
    b = timed(ctx, lambda: ctx.pv(pv, CSI, T, S, time_agg="mean", options=dict(night_skip=True, row_len=X)), reps=8)[0]
    c = timed(ctx, lambda: ctx.wind(*wargs, time_agg="mean"), reps=8)[0]
    print(f"blocks per CU {per_cu:4d}: pv map {a:.3f} ms  pv map early-out {b:.3f} ms  wind C3 map {c:.3f} ms", flush=True)
