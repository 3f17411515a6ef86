"""GPU: accuracy of the kernels' lean fp64 math (atlite_amd/csrc/atl_math.h) against numpy."""
import ctypes as C

import numpy as np
import pytest

from atlite_amd._lib import check

pytestmark = pytest.mark.gpu


def probe(ctx, fn, x, n_out=1):
    x = np.ascontiguousarray(x, dtype=np.float64)
    n = x.size if fn != 4 else x.size // 2
    d_in = ctx.upload(x)
    d_out = ctx.empty((n_out * n,))
    check(ctx.lib.atl_math_probe(ctx.handle, fn, d_in.ptr, n, d_out.ptr))
    return d_out.numpy()


def ulp_err(got, ref):
    return np.abs(got - ref) / np.spacing(np.abs(ref))


def test_sincos(ctx):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-np.pi, np.pi, 200000), rng.uniform(-7, 7, 100000), rng.uniform(-1e3, 1e3, 100000),
                        np.linspace(-1.6, 1.6, 20001), [0.0, -0.0, np.pi / 2, np.pi, 1e-300, 5e-324]])
    s, c = np.split(probe(ctx, 3, x, 2), 2)
    # absolute error bound holds everywhere; relative (ulp) bound away from the zeros of the result
    assert np.abs(s - np.sin(x)).max() < 2.3e-16 and np.abs(c - np.cos(x)).max() < 2.3e-16
    small = np.abs(x) < 1.6
    assert ulp_err(s[small], np.sin(x[small])).max() <= 2 and ulp_err(c[small & (np.abs(c) > 1e-3)],
                                                                      np.cos(x[small & (np.abs(c) > 1e-3)])).max() <= 2
    np.testing.assert_array_equal(probe(ctx, 0, x), s)
    np.testing.assert_array_equal(probe(ctx, 1, x), c)
    big = np.array([1e6, -3e8, 2.0**29])
    assert np.abs(probe(ctx, 0, big) - np.sin(big)).max() < 2.3e-16
    bad = probe(ctx, 0, np.array([np.nan, np.inf, -np.inf, 2.0**30, 1e300]))
    assert np.isnan(bad).all()


def test_log(ctx):
    rng = np.random.default_rng(1)
    x = np.concatenate([np.exp(rng.uniform(-700, 700, 300000)), rng.uniform(0.5, 2.0, 200000),
                        [1.0, 2.0, 0.5, np.sqrt(0.5), 1e-310, 5e-324, 1.7976931348623157e308, 1e-3, 1.361]])
    got = probe(ctx, 2, x)
    ref = np.log(x)
    assert ulp_err(got[ref != 0], ref[ref != 0]).max() <= 1.0
    assert got[x == 1.0][0] == 0.0
    sp = probe(ctx, 2, np.array([0.0, -0.0, -1.0, np.inf, np.nan]))
    assert sp[0] == -np.inf and sp[1] == -np.inf and np.isnan(sp[2]) and sp[3] == np.inf and np.isnan(sp[4])


def test_fast_div(ctx):
    rng = np.random.default_rng(2)
    a = rng.uniform(0, 2, 200000)
    b = rng.uniform(0.0174, 1.0, 200000)  # sin(alt) above the 1 degree cut
    got = probe(ctx, 4, np.concatenate([a, b]))
    assert ulp_err(got, a / b).max() <= 1.0


def test_table_log(ctx):
    rng = np.random.default_rng(3)
    x = np.concatenate([np.exp(rng.uniform(-700, 700, 300000)), rng.uniform(0.5, 2.0, 300000), 1 + rng.uniform(-1e-3, 1e-3, 50000),
                        [1.0, 2.0, 0.5, np.sqrt(0.5), np.sqrt(2.0), 2.2250738585072014e-308, 1.7976931348623157e308, 1e-3, 80.0, 100.0]])
    got = probe(ctx, 5, x)
    ref = np.log(x)
    nz = ref != 0
    assert ulp_err(got[nz], ref[nz]).max() <= 2.0
    assert got[x == 1.0][0] == 0.0


def test_lean_sqrt(ctx):
    """v_rsq_f64 seed + one coupled Newton step + residual correction: <= 1 ulp on [0, 2^500), zeros exact, NaN for
    negative / NaN arguments (the in-kernel solar position takes cos(altitude) and sin(azimuth) through it)."""
    rng = np.random.default_rng(11)
    x = np.concatenate([rng.random(300000), 10.0 ** rng.uniform(-300, 150, 100000), [0.0, 1.0, 4.0, 1e-300, 2.0 ** 499],
                        1.0 - 10.0 ** rng.uniform(-16, -1, 20000)])
    got = probe(ctx, 7, x)
    assert ulp_err(got[x > 0], np.sqrt(x[x > 0])).max() <= 1.0
    assert (got[x == 0] == 0.0).all()
    bad = probe(ctx, 7, np.array([-1.0, np.nan, -0.0]))
    assert np.isnan(bad[0]) and np.isnan(bad[1]) and bad[2] == 0.0



def test_lean_sqrt_rsqrt(ctx):
    """sqrt and 1 / sqrt from one coupled iteration on the v_rsq_f64 seed (the trackers' 1 / sqrt(1 + q^2))."""
    rng = np.random.default_rng(12)
    x = np.concatenate([1.0 + rng.random(200000) * 10.0, 1.0 + 10.0 ** rng.uniform(-16, 120, 100000), 10.0 ** rng.uniform(-140, 140, 100000),
                        [1.0, 2.0, 4.0, 2.0 ** 401]])
    got = probe(ctx, 8, x, n_out=2)
    n = x.size
    assert ulp_err(got[:n], np.sqrt(x)).max() <= 1.0
    assert ulp_err(got[n:], 1.0 / np.sqrt(x)).max() <= 2.0
