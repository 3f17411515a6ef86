"""Per-cell series on contiguous cubes of odd and even grids (wind, pv with / without the early-out): HIP-event medians.\nRun on the GPU box from the repo root; ATLITE_HIP_SERIES_NO_SHIFT=1 switches the per-slot line-grid placement off."""
import sys
sys.path.insert(0, ".")
import numpy as np
from atlite_amd import synthetic
from atlite_amd.device import Context
from tools.bench_configs import V, POW, CSI, timed
ctx = Context(0)
T = 8760
for (Y, X) in ((400, 400), (401, 401), (201, 201), (200, 200)):
    S = Y * X
    w = synthetic.wind_inputs(ctx, T, Y, X)
    fn = lambda: ctx.wind(w["wnd100m"], w["roughness"], V, POW / 3.06, 80.0, 100.0, "logarithmic", T, S)
    med, mn = timed(ctx, fn, reps=6)
    print(f"wind per-cell series {Y}x{X} contiguous (S%16={S%16}): {med:.3f} ms  {24*T*S/med/1e6/8000:.3f} of peak", flush=True)
    del w
    if S < 100000:
        inputs, _ = synthetic.pv_inputs(ctx, T, Y, X)
        for skip in (False, True):
            fn = lambda: ctx.pv(inputs, CSI, T, S, options=dict(night_skip=skip, row_len=X))
            med, mn = timed(ctx, fn, reps=6)
            print(f"pv per-cell series {Y}x{X} contiguous night_skip={int(skip)}: {med:.3f} ms  {T*S/(med*1e-3):.3e} cell-steps/s", flush=True)
        del inputs
