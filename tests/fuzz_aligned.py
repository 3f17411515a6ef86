#!/usr/bin/env python3
"""
Randomised differential test of line-aligned plans (atl_agg_create_aligned): the same contiguous cubes through the ordinary
plan and through the aligned one - random grids with S % 16 != 0 (2 / 4 / 8 / 16 alignment classes), 2-d tiles or flat strips,
random matrices (sparse, dense rows that reach the MFMA groups, explicit zeros, negative and NaN weights, empty rows),
converter (plain product, runoff x height, temperature, wind log / power law with a static or time-dependent roughness, pv
with and without the early-out), time reduction, NaN / inf in the cubes, windows of partial rows.  The two results agree to
rounding (rtol 1e-11, atol 1e-12 max: another order of the partial sums) with identical NaN patterns; the plain product and
runoff are also held against the oracle (rtol 1e-10).

    python tests/fuzz_aligned.py [n_cases] [seed]
"""
import os
import sys
from pathlib import Path

import numpy as np
import scipy.sparse as sp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from atlite_amd.device import Context  # noqa: E402
from oracle import atlite_oracle as orc  # noqa: E402
from tests import helpers as H  # noqa: E402

V = np.array([0, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 25, 25], dtype=float)
POW = np.array([0.0, 0.0, 0.005, 0.15, 0.3, 0.525, 0.905, 1.375, 1.95, 2.58, 2.96, 3.05, 3.06, 3.06, 0.0])
PV = dict(H.CSI, slope=np.radians(30.0), azimuth=np.radians(180.0))


def random_matrix(rng, N, S):
    kind = rng.choice(["sparse", "blobs", "dense"])
    if kind == "dense":
        M = sp.csr_matrix(0.5 + rng.random((N, S)))
    else:
        M = sp.random(N, S, density=float(rng.choice([0.01, 0.1, 0.5])), random_state=int(rng.integers(1 << 30)), format="csr")
    if M.nnz and rng.random() < 0.4:
        k = rng.integers(0, M.nnz, size=max(1, M.nnz // 30))
        M.data[k] = rng.choice([0.0, -1.5], size=len(k))
    if M.nnz and rng.random() < 0.2:
        M.data[int(rng.integers(0, M.nnz))] = np.nan
    return M


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    ctx = Context(0)
    worst, worst_case, fails, done = 0.0, "", 0, 0
    while done < n:
        T, Y, X = int(rng.integers(1, 120)), int(rng.integers(1, 24)), int(rng.integers(1, 70))
        S = Y * X
        if S % 16 == 0 or S < 16:
            continue
        done += 1
        N = int(rng.choice([1, 3, 7, 20, 40]))
        M = random_matrix(rng, N, S)
        row_len = X if rng.random() < 0.7 else None
        fam = str(rng.choice(["spmm", "runoff", "thermo", "wind", "pv"]))
        tagg = rng.choice([None, None, "sum", "mean"])
        if rng.random() < 0.2:
            os.environ["ATLITE_HIP_PARTIAL_BUDGET"] = "1"  # windows of 64 virtual slots
        else:
            os.environ.pop("ATLITE_HIP_PARTIAL_BUDGET", None)
        plans = (ctx.plan(M, row_len=row_len, cache=False), ctx.plan(M, row_len=row_len, aligned=True, cache=False))
        ref = None
        bad = rng.random() < 0.5  # NaN / inf in the cubes
        if fam in ("spmm", "runoff", "thermo"):
            a = rng.random((T, S)) * 10.0 + (250.0 if fam == "thermo" else 0.0)
            if bad:
                a[rng.random((T, S)) < 0.01] = rng.choice([np.nan, np.inf, -np.inf])
            d = ctx.upload(a)
            if fam == "spmm":
                run = lambda p: ctx.spmm(p, d, time_agg=tagg)  # noqa: E731
                ref = orc.aggregate_matrix(a, M)
            elif fam == "runoff":
                h = rng.random(S) * 800.0
                dh = ctx.upload(h)
                run = lambda p: ctx.runoff(d, dh, T, S, plan=p, time_agg=tagg)  # noqa: E731
                ref = orc.aggregate_matrix(a * h[None, :], M)
            else:
                run = lambda p: ctx.thermo(d, T, S, plan=p, time_agg=tagg)  # noqa: E731
        elif fam == "wind":
            w = H.wind_dataset(T, Y, X, seed=int(rng.integers(1 << 30)))
            v, z = w["wnd100m"].copy(), w["roughness"].copy()
            method = str(rng.choice(["logarithmic", "power"]))
            aux = z if method == "logarithmic" else w["wnd_shear_exp"].copy()
            if bad:
                v[rng.random((T, S)) < 0.01] = rng.choice([np.nan, np.inf, 0.0, 25.0])
                aux[rng.random((T, S)) < 0.01] = rng.choice([np.nan, 0.0, 100.0, -1.0])
            static = rng.random() < 0.3
            dv, da = ctx.upload(v), ctx.upload(np.ascontiguousarray(aux[0]) if static else aux)
            run = lambda p: ctx.wind(dv, da, V, POW / 3.06, 80.0, 100.0, method, T, S, plan=p, time_agg=tagg)  # noqa: E731
        else:
            ds = H.pv_dataset(T, Y, X, seed=int(rng.integers(1 << 30)))
            if bad:
                for k in ("influx_direct", "temperature", "albedo"):
                    ds[k][rng.random((T, S)) < 0.005] = np.nan
            dev = {k: ctx.upload(v) for k, v in ds.items()}
            opt = dict(night_skip=bool(rng.random() < 0.5))
            if rng.random() < 0.3:
                opt["trigon_model"] = "other"
            run = lambda p: ctx.pv(dev, PV, T, S, plan=p, time_agg=tagg, options=opt)  # noqa: E731
        a, b = run(plans[0]).numpy(), run(plans[1]).numpy()
        case = f"{fam} tagg={tagg} ({T},{Y},{X}) S%16={S % 16} N={N} row_len={row_len} nnz={M.nnz} bad={bad} budget={'ATLITE_HIP_PARTIAL_BUDGET' in os.environ}"
        with np.errstate(invalid="ignore"):
            fin = np.isfinite(a) & np.isfinite(b)
            same_nonfinite = np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~fin & ~np.isnan(a)], b[~fin & ~np.isnan(b)])
            scale = float(np.max(np.abs(a[fin]))) if fin.any() else 1.0
            # sums of mixed-sign weights cancel: the allowance follows the magnitude of the terms, not of the result
            mag = float(np.abs(M.data[np.isfinite(M.data)]).sum() / max(N, 1)) if M.nnz else 1.0
            allow = 1e-11 * np.abs(a[fin]) + 1e-12 * max(scale, 1e-300) + 1e-13 * mag * (1e3 if fam in ("pv", "thermo") else 30.0) * (T if tagg == "sum" else 1)
            err = float(np.max(np.abs(a[fin] - b[fin]) / allow)) if fin.any() else 0.0
        ok = same_nonfinite and err <= 1.0
        if ok and ref is not None and tagg is None and not bad and not np.isnan(M.data).any():
            ok = bool(np.allclose(b, ref, rtol=1e-10, atol=1e-12 * max(float(np.max(np.abs(ref))), 1e-300) + 1e-13 * mag * 30.0))
        if err > worst:
            worst, worst_case = err, case
        if not ok:
            fails += 1
            print("FAIL", case, "err", err, "nonfinite patterns equal:", same_nonfinite, flush=True)
        for p in plans:
            p.close()
    print(f"{n} cases, {fails} failures, worst difference {worst:.3e} of the allowance: {worst_case}")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
