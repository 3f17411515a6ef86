#!/usr/bin/env python3
"""
Dense / overlapping aggregation matrices: kernel time against partial rows per tile.

  R dense rows   - R shapes that each cover the whole grid (what matrix x layout products and country-wide
                   availability matrices look like): every tile carries exactly R partial rows.
  k layers       - k independent tessellations of 100 // k shapes stacked (overlapping polygons).

C2 shape (8760 x 200 x 200), pv (56 B/cell, heavy converter) and runoff (8 B/cell, nothing to hide behind).
usage: tools/bench_dense.py [pv|runoff|wind ...]
"""
import json
import sys
from pathlib import Path

import numpy as np
import scipy.sparse as sp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from atlite_amd import gis, synthetic  # noqa: E402
from atlite_amd.device import Context  # noqa: E402
from atlite_amd.device import interleave_enabled  # noqa: E402
from tools.bench_configs import CSI, POW, V, timed  # noqa: E402


def layers(Y, X, k, n_total=100):
    x, y = synthetic.grid_coords(Y, X)
    dx, dy = x[1] - x[0], y[1] - y[0]
    box = (x[0] - dx / 2, y[0] - dy / 2, x[-1] + dx / 2, y[-1] + dy / 2)
    mats = [gis.compute_indicatormatrix(x, y, gis.random_tessellation(max(2, n_total // k), box, seed=100 + j)) for j in range(k)]
    return sp.vstack(mats).tocsr()


def dense_rows(S, R, seed=3):
    rng = np.random.default_rng(seed)
    return sp.csr_matrix(0.5 + rng.random((R, S)))


def main():
    which = sys.argv[1:] or ["pv", "runoff"]
    ctx = Context(0)
    T, Y, X = 8760, 200, 200
    S = Y * X
    res = {}
    runs = {}
    if "pv" in which:
        inputs, _ = synthetic.pv_inputs(ctx, T, Y, X, interleaved=interleave_enabled())
        runs["pv"] = (56, lambda plan: ctx.pv(inputs, CSI, T, S, plan=plan, options=dict(night_skip=False)))
        runs["pv_night"] = (56, lambda plan: ctx.pv(inputs, CSI, T, S, plan=plan, options=dict(night_skip=True)))
    if "runoff" in which:
        d = synthetic.heat_runoff_inputs(ctx, T, Y, X)
        runs["runoff"] = (8, lambda plan: ctx.runoff(d["runoff"], d["height"], T, S, plan=plan))
    if "wind" in which:
        w = synthetic.wind_inputs(ctx, T, Y, X)
        runs["wind"] = (16, lambda plan: ctx.wind(w["wnd100m"], w["roughness"], V, POW / 3.06, 80.0, 100.0, "logarithmic", T, S, plan=plan))
    import os

    only = [int(v) for v in os.environ.get("ATL_DENSE_R", "").split(",") if v]  # e.g. "16,32": just these row counts
    mats = [(f"dense R={R}", dense_rows(S, R)) for R in (only or (1, 2, 3, 4, 8, 16, 32))]
    if not only:
        mats += [(f"layers k={k}", layers(Y, X, k)) for k in (1, 2, 4, 8)]
    chunk_sweep = [int(v) for v in os.environ.get("ATL_DENSE_CHUNKS", "").split(",") if v]  # e.g. "16,32,64": ATLITE_HIP_CHUNK A/B in one process
    for mname, M in mats:
        plan = ctx.plan(M, row_len=X)
        info = plan.info()
        rpt = info["n_partial_rows"] / max(1, info["n_segments"])
        for name, (nbytes, fn) in runs.items():
            if chunk_sweep:
                row = []
                for ch in chunk_sweep:
                    os.environ["ATLITE_HIP_CHUNK"] = str(ch)
                    row.append(f"chunk {ch}: {timed(ctx, lambda: fn(plan), reps=5)[0]:.3f} ms")
                del os.environ["ATLITE_HIP_CHUNK"]
                row.append(f"default: {timed(ctx, lambda: fn(plan), reps=5)[0]:.3f} ms")
                print(f"{name:9s} {mname:14s} ({rpt:5.1f} rows/tile)  " + "   ".join(row), flush=True)
                continue
            med, mn = timed(ctx, lambda: fn(plan), reps=4)
            res[f"{name} {mname}"] = dict(ms=med, rows_per_tile=rpt, P=info["n_partial_rows"])
            print(f"{name:9s} {mname:14s} shapes {M.shape[0]:4d}  partial rows {info['n_partial_rows']:6d} ({rpt:5.1f}/tile)  "
                  f"{med:8.3f} ms  {nbytes * T * S / (med * 1e-3) / 1e9:6.0f} GB/s", flush=True)
        del plan
    print(json.dumps(res))


if __name__ == "__main__":
    main()
