#!/usr/bin/env python3
"""
Is the "fast kind" of allocation (profiles/r03_interleave_probe.txt, section 3) the physically contiguous one?  The C2 workload
from slot-interleaved blocks obtained with hipMalloc and with hipExtMallocWithFlags(hipDeviceMallocContiguous), in a fresh
process and after the VRAM free lists were stirred (many allocations freed in shuffled order).
"""
import ctypes as C
import random
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from atlite_amd import synthetic  # noqa: E402
from atlite_amd._lib import check  # noqa: E402
from atlite_amd.device import Context, DeviceArray  # noqa: E402
from tools.bench_configs import CSI, shapes_matrix, timed  # noqa: E402

hip = C.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipFree.argtypes = [C.c_void_p]
T, Y, X = 8760, 200, 200
S = Y * X
NB = T * 7 * S * 8


def main():
    ctx = Context(0)
    M = shapes_matrix(Y, X, 100)
    plan = ctx.plan(M, row_len=X, ld=7 * S)
    names = list(synthetic.PV_VARS)
    src, _ = synthetic.pv_inputs(ctx, T, Y, X, interleaved=True)
    ref = None

    def views(base):
        return {k: DeviceArray(ctx, base + v * S * 8, (T, S), owned=False, ld=7 * S) for v, k in enumerate(names)}

    def run(tag, cubes):
        nonlocal ref
        med, mn = timed(ctx, lambda: ctx.pv(cubes, CSI, T, S, plan=plan, options=dict(night_skip=False)), reps=10)
        r = ctx.pv(cubes, CSI, T, S, plan=plan, options=dict(night_skip=False)).numpy()
        ref = r if ref is None else ref
        print(f"{tag:58s}: median {med:.3f} ms min {mn:.3f} ms   {'same bits' if np.array_equal(r, ref) else 'DIFFERENT'}", flush=True)

    def alloc(flag):
        p = C.c_void_p()
        rc = hip.hipExtMallocWithFlags(C.byref(p), NB, flag)
        return p.value if rc == 0 else None

    def filled(base):
        first = next(iter(src.values()))
        check(ctx.lib.atl_copy_2d(ctx.handle, base, 7 * S * 8, first.ptr, 7 * S * 8, 7 * S * 8, T, 2, 0))  # the whole block
        return views(base)

    run("hipMalloc, first large allocation of the process", src)
    for rnd in range(2):
        for flag, what in ((4, "hipDeviceMallocContiguous"), (0, "default flags")):
            b = alloc(flag)
            if b is None:
                print(f"{what}: allocation refused", flush=True)
                continue
            run(f"round {rnd}: hipExtMallocWithFlags, {what}", filled(b))
            hip.hipFree(C.c_void_p(b))
        if rnd == 0:  # stir the free lists: 60 blocks of 0.3-6 GB, freed in shuffled order
            rng = random.Random(1)
            blocks = []
            for _ in range(60):
                p = C.c_void_p()
                if hip.hipExtMallocWithFlags(C.byref(p), int(rng.uniform(0.3, 6.0) * 2**30), 0) == 0:
                    blocks.append(p.value)
            rng.shuffle(blocks)
            for b in blocks[::2]:
                hip.hipFree(C.c_void_p(b))
            keep = blocks[1::2]
            print(f"stirred: {len(blocks)} blocks allocated, every other one freed, {len(keep)} still held", flush=True)
    run("the first allocation again", src)


if __name__ == "__main__":
    main()
