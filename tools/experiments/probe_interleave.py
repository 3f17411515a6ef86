#!/usr/bin/env python3
"""
Does it matter WHERE the seven cubes of a pv dataset lie relative to one another?  The kernels address cube v, slot t as
ptr_v + t * ld, so one allocation of T slots of 7 x S cells (ptr_v = base + v * S, ld = 7 * S: the seven variables of a time
step side by side, "slot-interleaved") runs through the shipped kernels unchanged.  Times the headline workload both ways
(bit-identical results), plus the capacity-factor map and the wind / heat converters where they have more than one cube.
usage: tools/probe_interleave.py [Y X [shapes [T]]]
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from atlite_amd import synthetic  # noqa: E402
from atlite_amd._lib import check  # noqa: E402
from atlite_amd.device import Context, DeviceArray  # noqa: E402
from tools.bench_configs import CSI, POW, V, shapes_matrix, timed  # noqa: E402


def interleave(ctx, inputs, T, S, order=None, pad=0):
    """One (T, n*S') allocation; cube v of slot t at base + (t * n + v) * S'."""
    names = order or list(inputs)
    n = len(names)
    Sp = S + pad
    ld = n * Sp
    big = ctx.empty((T, ld))
    out = {}
    for v, k in enumerate(names):
        a = inputs[k]
        check(ctx.lib.atl_copy_2d(ctx.handle, big.ptr + v * Sp * 8, ld * 8, a.ptr, S * 8, S * 8, T, 2, 0))
        out[k] = DeviceArray(ctx, big.ptr + v * Sp * 8, (T, S), owner=big, ld=ld)
    return out, ld


def row_interleave(ctx, inputs, T, Y, X, M, names=None):
    """[slot][y][variable][x]: the seven variables of a grid row side by side.  Emulated with the shipped kernels as a
    grid of 7 * X columns of which only the first X carry weights; variable v starts v * X cells into the allocation."""
    import scipy.sparse as sp
    names = names or list(inputs)
    n = len(names)
    Xf = n * X
    Sf = Y * Xf
    big = ctx.empty((T * Sf + Xf,))
    out = {}
    for v, k in enumerate(names):
        # T * Y rows of X cells, Xf apart
        check(ctx.lib.atl_copy_2d(ctx.handle, big.ptr + v * X * 8, Xf * 8, inputs[k].ptr, X * 8, X * 8, T * Y, 2, 0))
        out[k] = DeviceArray(ctx, big.ptr + v * X * 8, (T, Sf), owner=big)
    m = sp.coo_matrix(M)
    Mf = sp.csr_matrix((m.data, (m.row, (m.col // X) * Xf + m.col % X)), shape=(M.shape[0], Sf))
    return out, Mf, Xf, Sf


def wind(ctx):
    """C3: wind 8760 x 400 x 400, two cubes (wnd100m, roughness)."""
    T, Y, X = 8760, 400, 400
    S = Y * X
    d = synthetic.wind_inputs(ctx, T, Y, X)
    M = shapes_matrix(Y, X, 100)
    res = {}
    for name in ("separate", "interleaved"):
        cubes, ld = (d, None) if name == "separate" else interleave(ctx, d, T, S)
        args = (cubes["wnd100m"], cubes["roughness"], V, POW / 3.06, 80.0, 100.0, "logarithmic", T, S)
        plan = ctx.plan(M, row_len=X, ld=ld)
        for what, kw, bpc in (("aggregated", dict(plan=plan), 16), ("time-mean map", dict(time_agg="mean"), 16), ("series", {}, 24)):
            fn = lambda: ctx.wind(*args, **kw)  # noqa: E731
            med, mn = timed(ctx, fn, reps=8)
            r = fn().numpy() if what != "series" else None
            same = "" if name == "separate" or r is None else ("  bit-identical" if np.array_equal(r, res[what], equal_nan=True) else "  DIFFERENT")
            if name == "separate":
                res[what] = r
            print(f"wind C3 {name:12s} {what:14s}: median {med:.3f} ms min {mn:.3f} ms {bpc * T * S / (med * 1e-3) / 1e9:.0f} GB/s on {bpc} B/cell{same}", flush=True)
        del cubes, args


def order(ctx):
    """Does the gain depend on WHEN the allocation was made?  interleaved first (generated in place), then separate cubes
    copied out of it, then a second interleaved copy, then a second set of separate cubes."""
    T, Y, X = 8760, 200, 200
    S = Y * X
    M = shapes_matrix(Y, X, 100)

    def run(tag, cubes, ld):
        plan = ctx.plan(M, row_len=X, ld=ld)
        fn = lambda: ctx.pv(cubes, CSI, T, S, plan=plan, options=dict(night_skip=False))  # noqa: E731
        med, mn = timed(ctx, fn, reps=10)
        print(f"{tag:34s} base {min(v.ptr for v in cubes.values()):#x}: median {med:.3f} ms min {mn:.3f} ms {56 * T * S / (med * 1e-3) / 1e9:.0f} GB/s", flush=True)

    il, _ = synthetic.pv_inputs(ctx, T, Y, X, interleaved=True)
    ld = next(iter(il.values())).ld
    run("1 interleaved (first allocation)", il, ld)
    sep = {k: ctx._relayout(v, None) for k, v in il.items()}
    run("2 separate (after it)", sep, None)
    il2, _ = interleave(ctx, sep, T, S)
    run("3 interleaved (third)", il2, ld)
    sep2 = {k: ctx._relayout(v, None) for k, v in il.items()}
    run("4 separate (fourth)", sep2, None)
    run("1 again", il, ld)
    run("2 again", sep, None)
    del il
    il3, _ = interleave(ctx, sep, T, S)
    run("5 interleaved (reusing 1's memory?)", il3, ld)


def region(ctx):
    """Layout or memory region?  The same 19.6 GB blocks hold the seven cubes once slot-interleaved and once stacked
    (cube v at v * T * S, i.e. the layout of seven separate allocations inside ONE allocation)."""
    T, Y, X = 8760, 200, 200
    S = Y * X
    M = shapes_matrix(Y, X, 100)
    names = list(synthetic.PV_VARS)

    def run(tag, cubes, ld):
        plan = ctx.plan(M, row_len=X, ld=ld)
        fn = lambda: ctx.pv(cubes, CSI, T, S, plan=plan, options=dict(night_skip=False))  # noqa: E731
        med, mn = timed(ctx, fn, reps=10)
        print(f"{tag:44s} base {min(v.ptr for v in cubes.values()):#x}: median {med:.3f} ms min {mn:.3f} ms {56 * T * S / (med * 1e-3) / 1e9:.0f} GB/s", flush=True)
        return fn().numpy()

    def fill(block, src, stacked):
        out = {}
        for v, k in enumerate(names):
            if stacked:
                d = DeviceArray(ctx, block.ptr + v * T * S * 8, (T, S), owner=block)
            else:
                d = DeviceArray(ctx, block.ptr + v * S * 8, (T, S), owner=block, ld=7 * S)
            check(ctx.lib.atl_copy_2d(ctx.handle, d.ptr, (d.ld or S) * 8, src[k].ptr, (src[k].ld or S) * 8, S * 8, T, 2, 0))
            out[k] = d
        ctx.sync()
        return out

    il, _ = synthetic.pv_inputs(ctx, T, Y, X, interleaved=True)
    A = next(iter(il.values()))._owner
    ref = run("A (first allocation) interleaved", il, 7 * S)
    sep = {k: ctx._relayout(v, None) for k, v in il.items()}
    run("seven separate allocations", sep, None)
    del il
    blocks = [("A", A)] + [(n, ctx.empty((T * 7 * S,))) for n in ("B", "C")]
    for name, blk in blocks:
        for stacked in (True, False, True):
            cubes = fill(blk, sep, stacked)
            r = run(f"{name} {'stacked (7 cubes one after another)' if stacked else 'slot-interleaved'}", cubes, None if stacked else 7 * S)
            assert np.array_equal(r, ref)
            del cubes
    del sep
    D = ctx.empty((T * 7 * S,))
    cubes = fill(D, fill(A, {k: v for k, v in fill(A, fill(D, dict(zip(names, [DeviceArray(ctx, A.ptr + v * T * S * 8, (T, S), owner=A) for v in range(7)])), True), True).items()}, True), False)
    run("D (allocated after the separate cubes were freed) interleaved", cubes, 7 * S)


def offsets(ctx):
    """Base address or physical placement?  The interleaved block at different offsets inside two arenas."""
    T, Y, X = 8760, 200, 200
    S = Y * X
    M = shapes_matrix(Y, X, 100)
    names = list(synthetic.PV_VARS)
    plan = ctx.plan(M, row_len=X, ld=7 * S)
    sep, _ = synthetic.pv_inputs(ctx, T, Y, X)
    plan_s = ctx.plan(M, row_len=X)
    med, mn = timed(ctx, lambda: ctx.pv(sep, CSI, T, S, plan=plan_s, options=dict(night_skip=False)), reps=10)
    print(f"seven separate allocations: median {med:.3f} ms; bases {[hex(v.ptr) for v in sep.values()]}", flush=True)
    MB = 1 << 20
    for an in ("arena 1", "arena 2", "arena 3"):
        arena = ctx.empty((T * 7 * S + 40 * MB // 8,))
        for off in (0, 128, 4096, 65536, 256 * 1024, MB, 2 * MB, 3 * MB, 4 * MB, 8 * MB, 16 * MB, 32 * MB):
            cubes = {}
            for v, k in enumerate(names):
                d = DeviceArray(ctx, arena.ptr + off + v * S * 8, (T, S), owner=arena, ld=7 * S)
                check(ctx.lib.atl_copy_2d(ctx.handle, d.ptr, d.ld * 8, sep[k].ptr, S * 8, S * 8, T, 2, 0))
                cubes[k] = d
            ctx.sync()
            med, mn = timed(ctx, lambda: ctx.pv(cubes, CSI, T, S, plan=plan, options=dict(night_skip=False)), reps=8)
            print(f"{an} base {arena.ptr:#x} + {off:>9d}: median {med:.3f} ms min {mn:.3f} ms", flush=True)
        if an == "arena 1":
            keep = arena  # arena 2 cannot reuse arena 1's memory; arena 3 may reuse arena 2's
        del arena, cubes


def chunks(ctx):
    """How much does the fused kernel's time depend on WHICH slots are in flight together (ATLITE_HIP_CHUNK: slots walked
    by one wave; the ~3000 resident waves cover ~9 chunks x 325 tiles)?"""
    import os

    T, Y, X = 8760, 200, 200
    S = Y * X
    M = shapes_matrix(Y, X, 100)
    il, _ = synthetic.pv_inputs(ctx, T, Y, X, interleaved=True)
    sep, _ = synthetic.pv_inputs(ctx, T, Y, X)
    plans = {"interleaved": ctx.plan(M, row_len=X, ld=7 * S), "separate": ctx.plan(M, row_len=X)}
    for ch in (8, 16, 24, 32, 40, 48, 56, 64, 72, 80, 96, 104, 128, 192, 256, 512):
        os.environ["ATLITE_HIP_CHUNK"] = str(ch)
        row = []
        for name, cubes in (("interleaved", il), ("separate", sep)):
            med, mn = timed(ctx, lambda: ctx.pv(cubes, CSI, T, S, plan=plans[name], options=dict(night_skip=False)), reps=8)
            row.append(f"{name} {med:.3f} (min {mn:.3f})")
        print(f"chunk {ch:4d} slots: " + "   ".join(row), flush=True)
    del os.environ["ATLITE_HIP_CHUNK"]


def strides(ctx):
    """One arena (fixed physical memory), the slot-interleaved block with different paddings between the cubes: does the
    slot stride interact with whatever makes an allocation fast or slow?"""
    T, Y, X = 8760, 200, 200
    S = Y * X
    M = shapes_matrix(Y, X, 100)
    names = list(synthetic.PV_VARS)
    sep, _ = synthetic.pv_inputs(ctx, T, Y, X)
    pads = (0, 16, 32, 256, 2048, 4096, 16384, 131072)  # cells between consecutive cubes of a slot
    for an in ("arena 1", "arena 2"):
        arena = ctx.empty((T * 7 * (S + max(pads)) + 64,))
        for pad in pads:
            Sp = S + pad
            ld = 7 * Sp
            cubes = {}
            for v, k in enumerate(names):
                d = DeviceArray(ctx, arena.ptr + v * Sp * 8, (T, S), owner=arena, ld=ld)
                check(ctx.lib.atl_copy_2d(ctx.handle, d.ptr, ld * 8, sep[k].ptr, S * 8, S * 8, T, 2, 0))
                cubes[k] = d
            ctx.sync()
            plan = ctx.plan(M, row_len=X, ld=ld)
            med, mn = timed(ctx, lambda: ctx.pv(cubes, CSI, T, S, plan=plan, options=dict(night_skip=False)), reps=8)
            print(f"{an} base {arena.ptr:#x} pad {pad * 8:>8d} B between cubes (slot stride {ld * 8} B): median {med:.3f} ms min {mn:.3f} ms", flush=True)
        keep = arena if an == "arena 1" else None  # noqa: F841 - arena 2 must be other memory
        del cubes


def vram_map(ctx):
    """Speed against position in VRAM: one allocation of most of the device memory, the 19.6 GB slot-interleaved block
    placed every 8 GiB inside it."""
    T, Y, X = 8760, 200, 200
    S = Y * X
    M = shapes_matrix(Y, X, 100)
    names = list(synthetic.PV_VARS)
    sep, _ = synthetic.pv_inputs(ctx, T, Y, X)
    ld = 7 * S
    plan = ctx.plan(M, row_len=X, ld=ld)
    GiB = 1 << 30
    total = int(sys.argv[2]) if len(sys.argv) > 2 else 240
    step = float(sys.argv[3]) if len(sys.argv) > 3 else 8.0
    arena = None
    while arena is None:
        try:
            arena = ctx.empty((total * GiB // 8,))
        except Exception:  # noqa: BLE001
            total -= 4
    need = T * ld * 8
    print(f"arena {total} GiB at {arena.ptr:#x}", flush=True)
    off = 0
    while off + need <= total * GiB:
        cubes = {}
        for v, k in enumerate(names):
            d = DeviceArray(ctx, arena.ptr + off + v * S * 8, (T, S), owner=arena, ld=ld)
            check(ctx.lib.atl_copy_2d(ctx.handle, d.ptr, ld * 8, sep[k].ptr, S * 8, S * 8, T, 2, 0))
            cubes[k] = d
        ctx.sync()
        med, mn = timed(ctx, lambda: ctx.pv(cubes, CSI, T, S, plan=plan, options=dict(night_skip=False)), reps=6)
        # a plain one-cube read of the same bytes: the block as (T, 7 S) through the per-cell time sum of runoff
        one = DeviceArray(ctx, arena.ptr + off, (T, ld), owner=arena)
        med1, _ = timed(ctx, lambda: ctx.runoff(one, None, T, ld, time_agg="sum"), reps=6)
        print(f"block at +{off / GiB:6.1f} GiB: fused pv median {med:.3f} ms min {mn:.3f} ms  {56 * T * S / (med * 1e-3) / 1e9:.0f} GB/s   "
              f"one-cube per-cell sum {med1:.3f} ms {8 * T * ld / (med1 * 1e-3) / 1e9:.0f} GB/s", flush=True)
        off += int(step * GiB)


def main():
    import os

    if sys.argv[1:2] == ["vram"]:
        return vram_map(Context(0))
    if sys.argv[1:2] == ["strides"]:
        return strides(Context(0))

    if sys.argv[1:2] == ["chunks"]:
        return chunks(Context(0))

    if os.environ.get("ATL_PROBE_TORCH"):  # does an initialised torch (its allocator, its streams) change the picture?
        import torch

        torch.cuda.set_device(0)
        _keep = torch.empty((100, 8760), dtype=torch.float64, device="cuda")  # noqa: F841
    if sys.argv[1:2] == ["offsets"]:
        return offsets(Context(0))
    if sys.argv[1:2] == ["region"]:
        return region(Context(0))
    if sys.argv[1:2] == ["wind"]:
        return wind(Context(0))
    if sys.argv[1:2] == ["order"]:
        return order(Context(0))
    Y, X = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (200, 200)
    nshapes = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    ctx = Context(0)
    T = int(sys.argv[4]) if len(sys.argv) > 4 else 8760
    S = Y * X
    inputs, _ = synthetic.pv_inputs(ctx, T, Y, X)
    M = shapes_matrix(Y, X, nshapes)
    res = {}
    for name in ("separate", "interleaved", "interleaved+2KiB"):
        if name == "separate":
            cubes, ld = inputs, None
        else:
            cubes, ld = interleave(ctx, inputs, T, S, pad=256 if "2KiB" in name else 0)
        plan = ctx.plan(M, row_len=X, ld=ld)
        info = plan.info()
        for skip in (False, True):
            fn = lambda: ctx.pv(cubes, CSI, T, S, plan=plan, options=dict(night_skip=skip))  # noqa: E731
            med, mn = timed(ctx, fn, reps=10)
            res[(name, skip)] = fn().numpy()
            print(f"pv {Y}x{X} {name:18s} ld={ld or S} tile {info['tile_w']}x{info['tile_h']} night_skip={int(skip)}: median {med:.3f} ms min {mn:.3f} ms "
                  f"{56 * T * S / (med * 1e-3) / 1e9:.0f} GB/s on 56 B/cell", flush=True)
        fn = lambda: ctx.pv(cubes, CSI, T, S, time_agg="mean", options=dict(night_skip=False, row_len=X))  # noqa: E731
        med, mn = timed(ctx, fn, reps=10)
        res[(name, "map")] = fn().numpy()
        print(f"pv {Y}x{X} {name:18s} capacity-factor map (every byte): median {med:.3f} ms min {mn:.3f}", flush=True)
        if name != "separate":
            for key in (False, True, "map"):
                same = np.array_equal(res[("separate", key)], res[(name, key)], equal_nan=True)
                print(f"   {name} vs separate [{key}]: {'bit-identical' if same else 'DIFFERENT'}")
        del cubes
    cubes, Mf, Xf, Sf = row_interleave(ctx, inputs, T, Y, X, M)
    plan = ctx.plan(Mf, row_len=Xf)
    info = plan.info()
    for skip in (False, True):
        fn = lambda: ctx.pv(cubes, CSI, T, Sf, plan=plan, options=dict(night_skip=skip))  # noqa: E731
        med, mn = timed(ctx, fn, reps=10)
        r = fn().numpy()
        a = res[("separate", skip)]
        print(f"pv {Y}x{X} row-interleaved (grid of {Xf} columns) tile {info['tile_w']}x{info['tile_h']} tiles {info['n_segments']} P={info['n_partial_rows']} "
              f"night_skip={int(skip)}: median {med:.3f} ms min {mn:.3f} ms {56 * T * S / (med * 1e-3) / 1e9:.0f} GB/s on 56 B/cell; "
              f"max rel diff vs separate {np.nanmax(np.abs(a - r) / np.maximum(np.abs(a), 1e-300)):.1e}", flush=True)


if __name__ == "__main__":
    main()
