#!/usr/bin/env python3
"""Long fuzz of the power-curve table builder and np.interp replacement of the wind kernels (host build of the same
source, atl_wind_interp_host), no GPU: random tables of every padded size - grid-aligned and not, repeated knots at the
ends and inside, one knot, huge / tiny spacings, inf / NaN values and end knots - and query points on knots, one ulp
beside them, outside the range, NaN / inf.  Finite tables: numpy's interval everywhere, numpy's bits on knots and
outside the range, one rounding of slope * dx + F[j] apart inside; non-finite tables: numpy's arr_interp bit for bit.
Run against the sanitizer build like tools/fuzz_reader.py."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from atlite_amd import _lib  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    lib = _lib.load()
    refused = 0
    for case in range(n_cases):
        n = int(rng.choice([1, 2, 3, int(rng.integers(4, 40)), int(rng.integers(40, 200)), int(rng.integers(200, 1000))]))
        kind = int(rng.integers(5))
        if kind == 0:  # grid-aligned (integers / halves), like the shipped turbines
            step = float(rng.choice([1.0, 0.5, 0.25, 0.1]))
            V = np.sort(np.round(rng.integers(0, 120, n) * step, 6))
        elif kind == 1:
            V = np.sort(rng.random(n) * 10.0 ** rng.uniform(-3, 4))
        elif kind == 2:  # clustered: many ties
            V = np.sort(np.round(rng.random(n) * 5, 1))
        elif kind == 3:  # wide dynamic range
            V = np.sort(10.0 ** rng.uniform(-200, 200, n) * rng.choice([-1, 1], n))
        else:
            V = np.sort(rng.standard_normal(n).cumsum())
        F = rng.random(n) * 10.0 ** rng.uniform(-3, 3)
        if rng.random() < 0.3 and n > 1:  # duplicate end knots
            V[1] = V[0]
        if rng.random() < 0.3 and n > 2:
            V[-2] = V[-1]
        finite = True
        if rng.random() < 0.25:
            finite = False
            k = rng.integers(0, n, size=int(rng.integers(1, 4)))
            F[k] = rng.choice([np.inf, -np.inf, np.nan], size=len(k))
            if rng.random() < 0.3:
                V[-1] = np.inf
            if rng.random() < 0.2:
                V[0] = -np.inf
        fin = V[np.isfinite(V)]
        span = (fin[-1] - fin[0]) if len(fin) > 1 else 1.0
        lo, hi = (fin[0], fin[-1]) if len(fin) else (0.0, 1.0)
        x = np.concatenate([fin, np.nextafter(fin, np.inf), np.nextafter(fin, -np.inf), rng.uniform(lo - 0.1 * span - 1, hi + 0.1 * span + 1, 300),
                            [np.nan, np.inf, -np.inf, 0.0, -0.0, 1e300, -1e300]])
        out = np.empty_like(x)
        rc = lib.atl_wind_interp_host(V.ctypes.data, F.ctypes.data, n, x.ctypes.data, len(x), out.ctypes.data)
        if rc != 0:  # only non-monotonic tables may be refused, and these are sorted
            raise AssertionError(("refused a sorted table", case, n, kind, _lib.last_error() if hasattr(_lib, "last_error") else rc))
        with np.errstate(all="ignore"):
            ref = np.interp(x, V, F)
        if n == 1:  # numpy's single-knot special case returns F[0] even for NaN
            ref = np.where(np.isnan(x), np.nan, ref)
        if not finite:
            assert np.array_equal(out, ref, equal_nan=True), ("non-finite table", case, n, kind)
            continue
        assert np.array_equal(np.isnan(out), np.isnan(ref)), ("NaN pattern", case, n, kind)
        on = np.isin(x, V) | (x <= V[0]) | (x >= V[-1]) | ~np.isfinite(x)
        assert np.array_equal(out[on], ref[on], equal_nan=True), ("knots / outside", case, n, kind, x[on][out[on] != ref[on]][:3])
        scale = np.maximum(np.abs(ref), np.abs(F).max())
        bad = ~(np.abs(out - ref) <= 2 * np.spacing(scale)) & ~np.isnan(ref)
        assert not bad.any(), ("inside", case, n, kind, x[bad][:3], out[bad][:3], ref[bad][:3])
    print(f"{n_cases} tables: numpy's interval, knot and boundary values everywhere, no crash")


if __name__ == "__main__":
    main()
