#!/usr/bin/env python3
"""
Padded slots against contiguous cubes on grids whose cell count is not a multiple of 16 (real-world ERA5 cutouts): the same
synthetic pv cubes once as (T, S) contiguous and once with every slot padded to a 128-byte line (ld_cells, what
Dataset.device() does for the library's own device copies), pv convert + aggregate and the per-cell capacity-factor map.
usage: tools/bench_pitch.py [Y X]...
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from atlite_amd import synthetic  # noqa: E402
from atlite_amd._lib import check  # noqa: E402
from atlite_amd.device import Context, pitch_for  # noqa: E402
from tools.bench_configs import CSI, shapes_matrix, timed  # noqa: E402


def main():
    grids = [(int(a), int(b)) for a, b in zip(sys.argv[1::2], sys.argv[2::2])] or [(189, 157), (201, 201), (241, 321)]
    ctx = Context(0)
    T = 8760
    for Y, X in grids:
        S = Y * X
        inputs, _ = synthetic.pv_inputs(ctx, T, Y, X)
        ld = pitch_for(S)
        M = shapes_matrix(Y, X, 100)
        res = {}
        for name, lay in (("contiguous", None), ("aligned", None), ("padded", ld)):
            if name != "contiguous" and ld is None:
                continue
            cubes = inputs
            if lay:
                cubes = {}
                for k, v in inputs.items():
                    p = ctx.empty_pitched((T, S), lay)
                    check(ctx.lib.atl_copy_2d(ctx.handle, p.ptr, lay * 8, v.ptr, S * 8, S * 8, T, 2, 0))
                    cubes[k] = p
            # "aligned": the contiguous cubes through the line-aligned plan (atl_agg_create_aligned)
            plan = ctx.plan(M, row_len=X, aligned=True) if name == "aligned" else ctx.plan(M, row_len=X, ld=lay)
            info = plan.info()
            for skip in (False, True):
                fn = lambda: ctx.pv(cubes, CSI, T, S, plan=plan, options=dict(night_skip=skip))  # noqa: E731
                med, mn = timed(ctx, fn, reps=6)
                res[(name, skip)] = fn().numpy()
                print(f"grid {Y} x {X} (S % 16 = {S % 16}) {name:10s} ld={lay or S} tile {info['tile_w']}x{info['tile_h']} P={info['n_partial_rows']} "
                      f"night_skip={int(skip)}: {med:.3f} ms  {T * S / (med * 1e-3):.3e} cell-steps/s", flush=True)
            if name == "aligned":
                continue
            fn = lambda: ctx.pv(cubes, CSI, T, S, time_agg="mean", options=dict(night_skip=True, row_len=X))  # noqa: E731
            med, mn = timed(ctx, fn, reps=6)
            res[(name, "map")] = fn().numpy()
            print(f"grid {Y} x {X} {name:10s} capacity-factor map (early-out): {med:.3f} ms", flush=True)
            del cubes
        if ld:
            for key in ((False), (True), ("map")):
                a, b = res[("contiguous", key)], res[("padded", key)]
                print(f"   padded vs contiguous [{key}]: max rel diff {np.max(np.abs(a - b) / np.maximum(np.abs(a), 1e-300)):.2e}")
                if ("aligned", key) in res:
                    b = res[("aligned", key)]
                    print(f"   aligned plan vs contiguous [{key}]: max rel diff {np.max(np.abs(a - b) / np.maximum(np.abs(a), 1e-300)):.2e}")
        del inputs


if __name__ == "__main__":
    main()
