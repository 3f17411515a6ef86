#!/usr/bin/env python3
"""
Randomised differential test of the device-side runoff post-processing (atl_rolling_mean, atl_order_statistic,
atl_zero_below, atl_normalize_rows) against pandas: random shapes, windows, min_periods, NaN / +-inf / constant-run /
sign patterns, pitched inputs.  Run on the GPU box:   python tests/fuzz_post.py [n_cases] [seed]
"""
import sys
from pathlib import Path

import numpy as np
import pandas as pd

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from atlite_amd.device import Context  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    ctx = Context(0)
    worst, fails = 0.0, 0
    for case in range(n):
        rows, T = int(rng.integers(1, 70)), int(rng.integers(1, 3000))
        kind = rng.choice(["gamma", "normal", "steps", "tiny"])
        if kind == "gamma":
            a = rng.gamma(0.4, 2.0, size=(rows, T))
        elif kind == "normal":
            a = rng.normal(size=(rows, T)) * 10.0 ** rng.integers(-3, 6)
        elif kind == "steps":
            a = np.repeat(rng.integers(-3, 4, size=(rows, T // 7 + 1)).astype(float), 7, axis=1)[:, :T] * 0.1
        else:
            a = rng.random((rows, T)) * 1e-300
        a[rng.random((rows, T)) < rng.choice([0.0, 0.02, 0.3])] = np.nan
        if rng.random() < 0.3:
            a[rng.random((rows, T)) < 0.01] = rng.choice([np.inf, -np.inf])
        if rng.random() < 0.3:
            a[rng.integers(rows), :] = np.nan
        w = int(rng.choice([1, 2, 3, 24, 168, 255, 256, 257, 1000, max(T, 1), T + 5]))
        mp = int(rng.choice([0, 1, 1, 1, min(2, w), min(w, 24)]))
        ld = T if rng.random() < 0.5 else T + int(rng.integers(1, 20))
        d = ctx.upload(a, ld=ld) if ld > T else ctx.upload(a)
        ref = pd.DataFrame(a.T).rolling(w, min_periods=mp).mean().values.T
        got = ctx.rolling_mean(d, w, mp).numpy()
        scale = np.nanmax(np.abs(np.where(np.isfinite(a), a, np.nan))) if np.isfinite(a).any() else 1.0
        with np.errstate(all="ignore"):
            err = np.abs(got - ref) / (1e-10 * np.abs(ref) + 1e-12 * scale)
        err = np.where((got == ref) | (np.isnan(got) & np.isnan(ref)), 0.0, err)
        e = float(np.nanmax(np.where(np.isnan(err), np.inf, err))) if err.size else 0.0
        if not e <= 1.0:
            fails += 1
            print(f"case {case}: rolling rows={rows} T={T} w={w} mp={mp} kind={kind} ld={ld}: error {e:.3g} of the allowance")
        worst = max(worst, e if np.isfinite(e) else 1e9)
        q = float(rng.choice([0.0, 5e-3, 0.25, 0.5, 0.999, 1.0, rng.random()]))
        qr = pd.Series(a.ravel()).quantile(q)
        qg = ctx.quantile(d, q)
        okq = (np.isnan(qr) and np.isnan(qg)) or qg == qr or abs(qg - qr) <= 1e-14 * abs(qr)
        if not okq:
            fails += 1
            print(f"case {case}: quantile q={q} rows={rows} T={T} kind={kind}: got {qg!r} ref {qr!r}")
        if np.isfinite(qr):
            z = ctx.zero_below(ctx.upload(a), qr).numpy()
            if not np.array_equal(z, np.where(a >= qr, a, 0.0)):
                fails += 1
                print(f"case {case}: zero_below differs")
        mask = rng.random(T) < 0.5
        refv = rng.normal(size=rows)
        refv[rng.random(rows) < 0.1] = np.nan
        fin = np.where(np.isinf(a), np.nan, a)  # (inf rows: inf / nan either way; compare on finite data)
        with np.errstate(all="ignore"):
            want = fin * (refv / np.nansum(fin[:, mask], axis=1))[:, None]
        gotn = ctx.normalize_rows(ctx.upload(fin), mask, refv).numpy()
        with np.errstate(all="ignore"):
            # rows whose masked sum cancels (|sum| << sum |x|) have no well-defined factor: any summation order is as good
            tot, mag = np.nansum(fin[:, mask], axis=1), np.nansum(np.abs(fin[:, mask]), axis=1)
            cond = (np.abs(tot) > 1e-6 * mag)[:, None]
            bad = cond & ~(np.isclose(gotn, want, rtol=1e-9, atol=0.0, equal_nan=True) | (np.isinf(gotn) & np.isinf(want)))
        if bad.any():
            fails += 1
            print(f"case {case}: normalize_rows differs at {np.argwhere(bad)[0]}: {gotn[bad][0]} vs {want[bad][0]}")
    print(f"{n} cases, {fails} failures, worst rolling error {worst:.3e} of the allowance")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
