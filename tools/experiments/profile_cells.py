#!/usr/bin/env python3
"""Per-cell pv kernels on the C2 shape (capacity-factor map and per-cell series, night early-out on and off): a target
for rocprofv3 (tools/r02_job_cells_prof.sh)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from atlite_amd import synthetic  # noqa: E402
from atlite_amd.device import Context  # noqa: E402
from tools.bench_configs import CSI  # noqa: E402

ctx = Context(0)
T, Y, X = 8760, 200, 200
S = Y * X
inputs, _ = synthetic.pv_inputs(ctx, T, Y, X)
for skip in (True, False):
    for _ in range(4):
        out = ctx.pv(inputs, CSI, T, S, time_agg="mean", options=dict(night_skip=skip, row_len=X))
    for _ in range(4):
        out = ctx.pv(inputs, CSI, T, S, options=dict(night_skip=skip, row_len=X))
    ctx.sync()
    del out
