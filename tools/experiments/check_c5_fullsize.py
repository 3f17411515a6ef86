#!/usr/bin/env python3
"""
BASELINE.json configs[4] at FULL size on one GPU: heat demand + runoff over 35040 x 400 x 400 fp64
(5.6e9 cells per variable = 44.8 GB each - beyond 2^32 elements, so every index path runs in its
64-bit range), 50 shapes.  The full launch must equal, bit for bit, the same cube processed as four
8760-step shards (whose indices stay below 2^31); also prints the full-size kernel times.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from atlite_amd import gis, synthetic  # noqa: E402
from atlite_amd.device import Context  # noqa: E402

T, Y, X, N = 35040, 400, 400, 50
ctx = Context(0)
S = Y * X
inp = synthetic.heat_runoff_inputs(ctx, T, Y, X)
x, y = synthetic.grid_coords(Y, X)
dx, dy = x[1] - x[0], y[1] - y[0]
M = gis.compute_indicatormatrix(x, y, gis.random_tessellation(N, (x[0] - dx / 2, y[0] - dy / 2, x[-1] + dx / 2, y[-1] + dy / 2)))
plan = ctx.plan(M, row_len=X)
ctx.set_profiling(True)

full = ctx.runoff(inp["runoff"], inp["height"], T, S, plan=plan)
ms = ctx.last_kernel_ms()
full = full.numpy()
parts = [ctx.runoff(inp["runoff"].slab(t0, t0 + 8760), inp["height"], 8760, S, plan=plan).numpy() for t0 in range(0, T, 8760)]
ok_r = np.array_equal(full, np.concatenate(parts, axis=1))
print(f"runoff  {T}x{Y}x{X}: fused kernel {ms:.3f} ms = {8 * T * S / ms / 1e6:.0f} GB/s; full == 4 shards: {ok_r}")

day_ptr = np.arange(0, T + 1, 24, dtype=np.int64)
full = ctx.heat_demand(inp["temperature"], day_ptr, 288.15, 1.0, 0.0, T, S, plan=plan)
ms = ctx.last_kernel_ms()
full = full.numpy()
parts = [ctx.heat_demand(inp["temperature"].slab(t0, t0 + 8760), np.arange(0, 8761, 24, dtype=np.int64), 288.15, 1.0, 0.0,
                         8760, S, plan=plan).numpy() for t0 in range(0, T, 8760)]
ok_h = np.array_equal(full, np.concatenate(parts, axis=1))
print(f"heat    {T}x{Y}x{X}: fused kernel {ms:.3f} ms = {8 * T * S / ms / 1e6:.0f} GB/s; full == 4 shards: {ok_h}")

# per-cell time mean over the whole cube (k_cells_timered + k_chunk_reduce at > 2^32 elements)
m_full = ctx.runoff(inp["runoff"], inp["height"], T, S, time_agg="sum").numpy()
m_parts = sum(ctx.runoff(inp["runoff"].slab(t0, t0 + 8760), inp["height"], 8760, S, time_agg="sum").numpy() for t0 in range(0, T, 8760))
ok_m = np.allclose(m_full, m_parts, rtol=1e-12)
print(f"per-cell time sum: full vs shards allclose(1e-12): {ok_m}")
sys.exit(0 if (ok_r and ok_h and ok_m) else 1)
