"""
The kernels' lean fp64 math (atlite_amd/csrc/atl_math.h) evaluated on the HOST: the routines are
__host__ __device__, so atl_math_probe_host() runs the very source the kernels compile (only the
hardware reciprocal seed and the hi/lo-word intrinsics are replaced) - polynomials, Cody-Waite
reduction, table-driven log, special cases and the guarded division are checked here without a GPU;
tests/test_gpu_math.py checks the device instantiation with the same bounds.
"""
import numpy as np
import pytest

from atlite_amd import _lib
from atlite_amd._lib import check


@pytest.fixture(autouse=True, params=["grid", "search"])
def wind_table_mode(request, monkeypatch):
    """Power curves with grid-aligned knots use a bucket lookup, the others a binary search: run every
    test through both (ATLITE_HIP_WIND_NO_GRID forces the search for aligned tables too)."""
    if request.param == "search":
        monkeypatch.setenv("ATLITE_HIP_WIND_NO_GRID", "1")
    else:
        monkeypatch.delenv("ATLITE_HIP_WIND_NO_GRID", raising=False)


def probe(fn, x, n_out=1):
    x = np.ascontiguousarray(x, dtype=np.float64)
    n = x.size if fn not in (4, 6) else x.size // 2
    out = np.empty(n_out * n)
    check(_lib.load().atl_math_probe_host(fn, x.ctypes.data, n, out.ctypes.data))
    return out


def ulp_err(got, ref):
    return np.abs(got - ref) / np.spacing(np.abs(ref))


def test_sincos_host():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-np.pi, np.pi, 100000), rng.uniform(-7, 7, 50000), rng.uniform(-1e3, 1e3, 50000),
                        np.linspace(-1.6, 1.6, 20001), [0.0, -0.0, np.pi / 2, np.pi, 1e-300, 5e-324]])
    s, c = np.split(probe(3, x, 2), 2)
    assert np.abs(s - np.sin(x)).max() < 2.3e-16 and np.abs(c - np.cos(x)).max() < 2.3e-16
    small = np.abs(x) < 1.6
    assert ulp_err(s[small], np.sin(x[small])).max() <= 2
    m = small & (np.abs(c) > 1e-3)
    assert ulp_err(c[m], np.cos(x[m])).max() <= 2
    np.testing.assert_array_equal(probe(0, x), s)
    np.testing.assert_array_equal(probe(1, x), c)
    big = np.array([1e6, -3e8, 2.0**29])
    assert np.abs(probe(0, big) - np.sin(big)).max() < 2.3e-16
    assert np.isnan(probe(0, np.array([np.nan, np.inf, -np.inf, 2.0**30, 1e300]))).all()
    assert np.isnan(probe(1, np.array([np.nan, np.inf, -np.inf, 2.0**30, 1e300]))).all()


def test_log_host():
    rng = np.random.default_rng(1)
    x = np.concatenate([np.exp(rng.uniform(-700, 700, 150000)), rng.uniform(0.5, 2.0, 100000),
                        [1.0, 2.0, 0.5, np.sqrt(0.5), 1e-310, 5e-324, 1.7976931348623157e308, 1e-3, 1.361]])
    got, ref = probe(2, x), np.log(x)
    assert ulp_err(got[ref != 0], ref[ref != 0]).max() <= 1.0
    assert got[x == 1.0][0] == 0.0
    with np.errstate(all="ignore"):
        sp = probe(2, np.array([0.0, -0.0, -1.0, np.inf, np.nan]))
    assert sp[0] == -np.inf and sp[1] == -np.inf and np.isnan(sp[2]) and sp[3] == np.inf and np.isnan(sp[4])


def test_table_log_host():
    rng = np.random.default_rng(3)
    x = np.concatenate([np.exp(rng.uniform(-700, 700, 150000)), rng.uniform(0.5, 2.0, 150000), 1 + rng.uniform(-1e-3, 1e-3, 50000),
                        [1.0, 2.0, 0.5, np.sqrt(0.5), np.sqrt(2.0), 2.2250738585072014e-308, 1.7976931348623157e308, 1e-3, 80.0, 100.0]])
    got, ref = probe(5, x), np.log(x)
    nz = ref != 0
    assert ulp_err(got[nz], ref[nz]).max() <= 2.0
    assert got[x == 1.0][0] == 0.0
    # arguments the wind kernel feeds it unselected (negative, zero, NaN, inf, subnormal): any value, no fault
    with np.errstate(all="ignore"):
        probe(5, np.array([-1.0, 0.0, np.nan, np.inf, 5e-324, -np.inf]))


def test_lean_sqrt_host():
    """lean_sqrt (reciprocal-square-root seed, one coupled Newton step, residual correction): <= 1 ulp on [0, 2^500),
    exact zeros, NaN for negative / NaN arguments."""
    rng = np.random.default_rng(11)
    x = np.concatenate([rng.random(20000), 10.0 ** rng.uniform(-300, 150, 20000), [0.0, 1.0, 4.0, 1e-300, 2.0 ** 499],
                        1.0 - 10.0 ** rng.uniform(-16, -1, 2000)])
    got = probe(7, x)
    assert ulp_err(got[x > 0], np.sqrt(x[x > 0])).max() <= 1.0
    assert got[x == 0].tolist() == [0.0]
    bad = probe(7, np.array([-1.0, np.nan, -0.0]))
    assert np.isnan(bad[0]) and np.isnan(bad[1]) and bad[2] == 0.0 and np.signbit(bad[2])


def test_divisions_host():
    rng = np.random.default_rng(2)
    a, b = rng.uniform(0, 2, 100000), rng.uniform(0.0174, 1.0, 100000)  # sin(alt) above the 1 degree cut
    assert ulp_err(probe(4, np.concatenate([a, b])), a / b).max() <= 1.0
    # guarded_div: fast path inside the normal range, IEEE behaviour outside it
    a = np.concatenate([rng.standard_normal(50000) * 10.0 ** rng.uniform(-300, 300, 50000),
                        [0.0, 1.0, -1.0, 0.0, np.inf, np.inf, np.nan, 1.0, 5e-324, 1e-310, 1e308, 3.0]])
    b = np.concatenate([rng.standard_normal(50000) * 10.0 ** rng.uniform(-300, 300, 50000),
                        [0.0, 0.0, 0.0, 1.0, np.inf, 2.0, 1.0, np.nan, 3.0, 1e-5, 1e-5, np.inf]])
    with np.errstate(all="ignore"):
        ref = a / b
        got = probe(6, np.concatenate([a, b]))
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(np.isinf(got), np.isinf(ref))
    fin = np.isfinite(ref) & (ref != 0)
    assert np.array_equal(np.sign(got[np.isinf(ref)]), np.sign(ref[np.isinf(ref)]))
    assert ulp_err(got[fin], ref[fin]).max() <= 1.0
    assert np.array_equal(got[ref == 0], ref[ref == 0])


def test_interp_matches_numpy():
    """np.interp through the padded-table search of the wind kernels (host build of the same source), on
    every shipped turbine curve, a smoothed curve and synthetic tables of every padded size with repeated
    knots: the SAME interval as numpy everywhere, bit-for-bit numpy at knots (the upper one of repeated
    knots), outside the range, at +-inf and for NaN; inside an interval one FMA replaces numpy's
    multiply-add, i.e. at most one ulp of the operands apart."""
    import os

    import pytest
    import yaml

    from atlite_amd.resource import get_windturbineconfig, windturbine_smooth

    lib = _lib.load()
    rng = np.random.default_rng(4)
    root = os.path.dirname(os.path.dirname(__file__))
    names = list(yaml.safe_load(open(f"{root}/atlite_amd/resources/technologies.yaml"))["windturbine"])
    tables = []
    for nme in names:
        tb = get_windturbineconfig(nme)
        tables.append((np.asarray(tb["V"], float), np.asarray(tb["POW"], float) / float(tb["P"])))
    sm = windturbine_smooth(get_windturbineconfig("Vestas_V112_3MW"), params=True)
    tables.append((np.asarray(sm["V"], float), np.asarray(sm["POW"], float) / float(sm["P"])))
    for n in (1, 2, 3, 15, 16, 17, 31, 32, 33, 127, 128, 129, 300, 1023):  # every padded size, repeated knots
        tables.append((np.sort(np.round(rng.random(n) * 30, 1)), rng.random(n)))

    def run(V, F, x):
        out = np.empty_like(x)
        check(lib.atl_wind_interp_host(V.ctypes.data, F.ctypes.data, len(V), x.ctypes.data, len(x), out.ctypes.data))
        return out

    for V, F in tables:
        exact = np.concatenate([V, [V[0] - 1.0, V[0] - 1e-9, V[-1] + 1e-9, V[-1] + 7.0, np.inf, -np.inf, 1e300, -1e300]])
        assert np.array_equal(run(V, F, exact), np.interp(exact, V, F), equal_nan=True), len(V)
        # NaN in, NaN out (numpy does the same except for its single-knot special case, which returns F[0])
        assert np.isnan(run(V, F, np.array([np.nan])))[0]
        inner = np.concatenate([V - 1e-9, V + 1e-9, np.nextafter(V, np.inf), np.nextafter(V, -np.inf),
                                rng.uniform(V[0] - 5, V[-1] + 5, 4000), [0.0, -0.0]])
        got, ref = run(V, F, inner), np.interp(inner, V, F)
        scale = np.maximum(np.abs(ref), np.abs(F).max())  # one rounding of slope * dx + F[j]: an ulp of the operands
        assert (np.abs(got - ref) <= np.spacing(scale)).all(), len(V)
    bad = np.array([0.0, 2.0, 1.0])
    with pytest.raises(ValueError, match="increasing"):
        check(lib.atl_wind_interp_host(bad.ctypes.data, bad.ctypes.data, 3, bad.ctypes.data, 3, bad.ctypes.data))


def test_interp_nonfinite_tables_literal_path():
    """Tables holding inf / NaN take the literal transcription of numpy's arr_interp: identical results,
    including numpy's NaN-repair rules (slope * dx = NaN -> evaluate from the right knot, equal values)."""
    lib = _lib.load()
    rng = np.random.default_rng(8)
    for case in range(200):
        n = int(rng.integers(2, 40))
        V = np.sort(np.round(rng.random(n) * 30, 1))
        F = rng.random(n)
        k = rng.integers(0, n, size=int(rng.integers(1, 4)))
        F[k] = rng.choice([np.inf, -np.inf, np.nan], size=len(k))
        if rng.random() < 0.3:
            V[-1] = np.inf
        if rng.random() < 0.2:
            V[0] = -np.inf
        x = np.concatenate([V[np.isfinite(V)], rng.uniform(-5, 35, 300), [np.nan, np.inf, -np.inf]])
        out = np.empty_like(x)
        check(lib.atl_wind_interp_host(V.ctypes.data, F.ctypes.data, n, x.ctypes.data, len(x), out.ctypes.data))
        with np.errstate(all="ignore"):
            ref = np.interp(x, V, F)
        assert np.array_equal(out, ref, equal_nan=True), (case, V, F, x[~((out == ref) | (np.isnan(out) & np.isnan(ref)))][:5])


def test_lean_sqrt_rsqrt_host():
    """sqrt and 1 / sqrt from one coupled iteration (the trackers' closed forms: 1 / sqrt(1 + q^2)): <= 1 / <= 2 ulp
    on the range the callers use (1 <= x < 2^401) and well beyond."""
    rng = np.random.default_rng(12)
    x = np.concatenate([1.0 + rng.random(20000) * 10.0, 1.0 + 10.0 ** rng.uniform(-16, 120, 20000), 10.0 ** rng.uniform(-140, 140, 20000),
                        [1.0, 2.0, 4.0, 2.0 ** 401]])
    got = probe(8, x, n_out=2)
    n = x.size
    assert ulp_err(got[:n], np.sqrt(x)).max() <= 1.0
    assert ulp_err(got[n:], 1.0 / np.sqrt(x)).max() <= 2.0
