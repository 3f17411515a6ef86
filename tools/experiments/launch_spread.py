#!/usr/bin/env python3
"""
Launch-to-launch spread of the early-out pv kernel and the aggregated wind kernel (2.08-2.71 ms
resp. 3.74-4.55 ms over seven launches under rocprofv3).  Runs N back-to-back launches of each kernel - after an idle
gap, so that the clock / power state at the start of a burst is part of the picture - and prints every launch's
HIP-event time.  Under  rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace  the per-dispatch counter gives each launch's
effective shader clock (GRBM_GUI_ACTIVE / 8 XCDs / duration; MI355X_MICROARCH.md "DVFS give-back"):
tools/rocpd_clock_per_launch.py turns the database into the table kept under profiles/.
  usage: launch_spread.py [n_launches] [which: night,base,wind,runoff]
"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from atlite_amd import gis, synthetic  # noqa: E402
from atlite_amd.device import Context  # noqa: E402
from atlite_amd.device import interleave_enabled  # noqa: E402
from tools.bench_configs import CSI, POW, V, shapes_matrix  # noqa: E402


def burst(ctx, fn, n, idle_s=0.5):
    ctx.sync()
    time.sleep(idle_s)  # let the chip fall back to its idle state: the burst starts cold
    ctx.set_profiling(max(2, n))
    for _ in range(n):
        out = fn()
        del out
    ctx.sync()
    return ctx.kernel_times()[-n:]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    which = (sys.argv[2] if len(sys.argv) > 2 else "night,base,wind").split(",")
    ctx = Context(0)
    res = {}

    def show(name, ms):
        ms = np.asarray(ms)
        res[name] = dict(ms=[round(float(v), 4) for v in ms], min=float(ms.min()), median=float(np.median(ms)), mean=float(ms.mean()),
                         max=float(ms.max()), first5_mean=float(ms[:5].mean()), last5_mean=float(ms[-5:].mean()))
        print(f"{name:28s} n={len(ms)} min {ms.min():.3f} median {np.median(ms):.3f} mean {ms.mean():.3f} max {ms.max():.3f} ms | "
              f"first 5 {ms[:5].mean():.3f}  last 5 {ms[-5:].mean():.3f}", flush=True)
        print("   " + " ".join(f"{v:.3f}" for v in ms), flush=True)

    if "night" in which or "base" in which:
        T, Y, X = 8760, 200, 200
        S = Y * X
        inputs, _ = synthetic.pv_inputs(ctx, T, Y, X, interleaved=interleave_enabled())
        plan = ctx.plan(shapes_matrix(Y, X, 100), row_len=X)
        for rep in range(2):  # two bursts each: is the pattern inside a burst reproducible?
            if "night" in which:
                show(f"pv C2 night early-out #{rep}", burst(ctx, lambda: ctx.pv(inputs, CSI, T, S, plan=plan, options=dict(night_skip=True)), n))
            if "base" in which:
                show(f"pv C2 every byte read #{rep}", burst(ctx, lambda: ctx.pv(inputs, CSI, T, S, plan=plan, options=dict(night_skip=False)), n))
        del inputs, plan
    if "wind" in which:
        T, Y, X = 8760, 400, 400
        S = Y * X
        d = synthetic.wind_inputs(ctx, T, Y, X)
        plan = ctx.plan(shapes_matrix(Y, X, 100), row_len=X)
        args = (d["wnd100m"], d["roughness"], V, POW / 3.06, 80.0, 100.0, "logarithmic", T, S)
        for rep in range(2):
            show(f"wind C3 aggregated #{rep}", burst(ctx, lambda: ctx.wind(*args, plan=plan), n))
        del d, plan
    if "runoff" in which:
        T, Y, X = 4380, 400, 400
        S = Y * X
        d = synthetic.heat_runoff_inputs(ctx, T, Y, X)
        plan = ctx.plan(shapes_matrix(Y, X, 50), row_len=X)
        show("runoff C5 shard", burst(ctx, lambda: ctx.runoff(d["runoff"], d["height"], T, S, plan=plan), n))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
