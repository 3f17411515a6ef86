#!/usr/bin/env python3
"""
Does what a block HOLDS matter to how fast it is read, or only where it lies?  One arena over most of the device memory; the C2 block
(slot-interleaved pv cubes) is placed every 16 GiB and timed with the fused kernel; at the fastest and at the slowest position the same
bytes are then read (a) by the fused kernel and the per-cell pv map on the real cubes, (b) by the one-cube runoff sum on the real
cubes, on zeros and on random doubles, (c) by the per-cell pv map on zeros.  (profiles/r03_vram_map.txt: the placement experiment's
probes read zeros / stale bytes and did not predict the fused kernel.)
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from atlite_amd import _lib, synthetic  # noqa: E402
from atlite_amd._lib import check  # noqa: E402
from atlite_amd.device import Context, DeviceArray  # noqa: E402
from tools.bench_configs import CSI, shapes_matrix, timed  # noqa: E402

ctx = Context(0)
T, Y, X = 8760, 200, 200
S = Y * X
ld = 7 * S
names = list(synthetic.PV_VARS)
M = shapes_matrix(Y, X, 100)
plan = ctx.plan(M, row_len=X, ld=ld)
sep, _ = synthetic.pv_inputs(ctx, T, Y, X)
GiB = 1 << 30
total = 248
arena = None
while arena is None:
    try:
        arena = ctx.empty((total * GiB // 8,))
    except Exception:  # noqa: BLE001
        total -= 8
need = T * ld * 8


def cubes_at(off):
    return {k: DeviceArray(ctx, arena.ptr + off + v * S * 8, (T, S), owner=arena, ld=ld) for v, k in enumerate(names)}


def fill_real(off):
    for k, d in cubes_at(off).items():
        check(ctx.lib.atl_copy_2d(ctx.handle, d.ptr, ld * 8, sep[k].ptr, S * 8, S * 8, T, 2, 0))
    ctx.sync()


def fused(off):
    c = cubes_at(off)
    return timed(ctx, lambda: ctx.pv(c, CSI, T, S, plan=plan, options=dict(night_skip=False)), reps=6)[0]


def pvmap(off):
    c = cubes_at(off)
    return timed(ctx, lambda: ctx.pv(c, CSI, T, S, time_agg="mean", options=dict(night_skip=False)), reps=6)[0]


def onecube(off):
    one = DeviceArray(ctx, arena.ptr + off, (T, ld), owner=arena)
    return timed(ctx, lambda: ctx.runoff(one, None, T, ld, time_agg="sum"), reps=6)[0]


scan = {}
off = 0
while off + need <= total * GiB:
    fill_real(off)
    scan[off] = fused(off)
    off += 16 * GiB
print("fused kernel on the real cubes by position:", {int(o / GiB): round(t, 3) for o, t in scan.items()}, flush=True)
fast, slow = min(scan, key=scan.get), max(scan, key=scan.get)
for tag, o in (("FAST position", fast), ("SLOW position", slow)):
    fill_real(o)
    a, b, c = fused(o), pvmap(o), onecube(o)
    check(ctx.lib.atl_memset(ctx.handle, arena.ptr + o, 0, need))
    z1, z7 = onecube(o), pvmap(o)
    check(ctx.lib.atl_synth_field(ctx.handle, _lib.SYN_UNIFORM, 7, 1, 0.0, 1000.0, 0, T, ld, arena.ptr + o))
    r1 = onecube(o)
    fill_real(o)
    a2 = fused(o)
    print(f"{tag} (+{o / GiB:.0f} GiB): real cubes: fused {a:.3f} ms, pv map {b:.3f}, one-cube sum {c:.3f} | zeros: one-cube {z1:.3f}, pv map {z7:.3f} | "
          f"random doubles: one-cube {r1:.3f} | real cubes again: fused {a2:.3f}", flush=True)
