#!/usr/bin/env python3
"""
Does the per-cell wind series (BASELINE configs[2]: 16 B read + 8 B written per cell-step, 8760 x 400 x 400) depend on WHERE
its three cubes lie?  bench.py's c3_series leg is bimodal from process to process on one box (5.30 / 5.6 / 6.0-6.2 ms with
the same library, profiles/r06_bench_* and gpurun_out/r06_final1): this probe varies, inside ONE process, the output cube's
offset within a larger allocation, the inputs' layout (slot-interleaved pool / one allocation each) and the order of the
allocations, and prints the kernel's HIP-event time for each.
"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from atlite_amd import _lib, synthetic  # noqa: E402
from atlite_amd.device import Context, SlotPool, pitch_for  # noqa: E402
from atlite_amd.resource import get_windturbineconfig  # noqa: E402

ctx = Context(0)
T, Y, X = 8760, 400, 400
S = Y * X
turb = get_windturbineconfig("Vestas_V112_3MW")
V, POW, P, hub = turb["V"], turb["POW"], turb["P"], turb["hub_height"]


def fill(dst_w, dst_z):
    tmp = ctx.empty((T, S))
    for var, kind, p0, p1, dst in ((5, _lib.SYN_RAYLEIGH, 8.0, 0.0, dst_w), (6, _lib.SYN_EXPLOG, 1e-3, 1.5e3, dst_z)):
        _lib.check(ctx.lib.atl_synth_field(ctx.handle, kind, 42, var, p0, p1, 0, T, S, tmp.ptr))
        _lib.check(ctx.lib.atl_copy_2d(ctx.handle, dst.ptr, (dst.ld or S) * 8, tmp.ptr, S * 8, S * 8, T, 2, 0))
    ctx.sync()


def run(wnd, z0, out_ptr, reps=8, warm=6):
    ctx.set_profiling(True)
    ms = []
    for i in range(warm + reps):
        ctx.wind(wnd, z0, V, POW / P, hub, 100.0, "logarithmic", T, S, out=(out_ptr, S))
        ctx.sync()
        if i >= warm:
            ms.append(ctx.last_kernel_ms())
    return float(np.mean(ms)), float(np.min(ms))


big = ctx.empty((T * S + (64 << 20) // 8,))  # the output cube somewhere inside this
pool = SlotPool(ctx, T, S, ["wnd100m", "roughness"], pitch_for(S))
wi, zi = pool.view("wnd100m"), pool.view("roughness")
fill(wi, zi)
ws, zs = ctx.empty((T, S)), ctx.empty((T, S))
fill(ws, zs)
print(f"allocations: out {big.ptr:#x}  pool {wi.ptr:#x}  separate {ws.ptr:#x} {zs.ptr:#x}", flush=True)
for off in (0, 128, 4096, 65536, 1 << 20, (2 << 20) + 4096, 16 << 20, 48 << 20):
    a, b = run(wi, zi, big.ptr + off)
    c, d = run(ws, zs, big.ptr + off)
    print(f"out + {off:>9d} B   interleaved inputs: mean {a:.3f} min {b:.3f} ms   separate inputs: mean {c:.3f} min {d:.3f} ms", flush=True)
# a fresh output allocation made AFTER the inputs (what bench.py's leg gets from ctx.wind): a few of them
held = []
for k in range(4):
    o = ctx.empty((T, S))
    held.append(o)
    a, b = run(wi, zi, o.ptr)
    print(f"fresh output {k} at {o.ptr:#x}: mean {a:.3f} min {b:.3f} ms", flush=True)
