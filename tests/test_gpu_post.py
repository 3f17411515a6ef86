"""
GPU: runoff()'s post-processing of the (shapes x time) result ON THE DEVICE (atl_rolling_mean, atl_order_statistic,
atl_zero_below, atl_normalize_rows; atlite/convert.py:1046-1082) against pandas / the oracle's restatement - the rolling
mean with NaN gaps, windows longer than the series, segment boundaries of the kernel (256 steps) and constant runs; the
quantile at every kind of virtual index (first, last, exact order statistic, repeated values, NaN-thinned arrays); the
whole chain through Cutout.runoff() at the size of BASELINE.json's configs[4] result (50 x 35040).
Tolerance: rtol 1e-10, atol 1e-12 max (north star); the order statistics themselves are exact.
"""
import numpy as np
import pandas as pd
import pytest
import scipy.sparse as sp

from atlite_amd import Cutout, Dataset
from oracle import atlite_oracle as orc

pytestmark = pytest.mark.gpu


def close(got, ref):
    scale = float(np.nanmax(np.abs(ref))) if np.isfinite(ref).any() else 1.0
    np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-12 * scale, equal_nan=True)


@pytest.mark.parametrize("rows,T,window", [(1, 1, 1), (3, 5, 168), (7, 1000, 24), (50, 3000, 168), (2, 777, 256), (4, 600, 257),
                                           (5, 513, 1), (64, 300, 300), (130, 258, 7)])
def test_rolling_mean_matches_pandas(ctx, rows, T, window):
    rng = np.random.default_rng(rows * 1000 + T + window)
    a = rng.gamma(0.7, 3.0, size=(rows, T))
    a[rng.random((rows, T)) < 0.05] = np.nan
    if T > 40:
        a[0, 10:30] = np.nan  # a gap longer than small windows: NaN results inside it
        a[-1, 5:25] = 0.1     # a constant run: pandas returns the value itself
    ref = pd.DataFrame(a.T).rolling(window, min_periods=1).mean().values.T
    got = ctx.rolling_mean(ctx.upload(a), window, 1).numpy()
    close(got, ref)
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    for mp in (0, min(3, window)):
        ref = pd.DataFrame(a.T).rolling(window, min_periods=mp).mean().values.T
        close(ctx.rolling_mean(ctx.upload(a), window, mp).numpy(), ref)


def test_rolling_mean_signs_constants_and_bad_arguments(ctx):
    # negative values (pandas' sign clamps), exact constants, +-inf treated as missing like pandas' rolling does
    a = np.array([[1e16, 1.0, 1.0, 1.0, 1.0, -1e16, 1.0, 0.1, 0.1, 0.1, 0.1, np.inf, 2.0, 3.0, -np.inf, 4.0],
                  [-1.0, -2.0, -3.0, np.nan, -4.0, -5.0, -6.0, -0.0, 0.0, 0.0, 1e-300, 1e-300, 1e-300, 5.0, 5.0, 5.0]])
    for w in (2, 3, 5):
        ref = pd.DataFrame(a.T).rolling(w, min_periods=1).mean().values.T
        got = ctx.rolling_mean(ctx.upload(a), w, 1).numpy()
        np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-12 * 1e16, equal_nan=True)
    assert (ctx.rolling_mean(ctx.upload(np.full((1, 9), 0.1)), 4).numpy() == 0.1).all()
    with pytest.raises(ValueError, match="window must be >= 1"):
        ctx.rolling_mean(ctx.upload(a), 0)
    with pytest.raises(ValueError, match="min_periods"):
        ctx.rolling_mean(ctx.upload(a), 3, 4)


@pytest.mark.parametrize("n,q", [(1, 0.3), (2, 0.5), (10, 0.0), (10, 1.0), (11, 0.5), (1000, 5e-3), (1000, 0.3), (4097, 0.999),
                                 (250000, 5e-3), (250000, 0.77)])
def test_quantile_matches_pandas(ctx, n, q):
    rng = np.random.default_rng(n + int(q * 1000))
    rows = 7 if n % 7 == 0 else (5 if n % 5 == 0 else 1)
    a = rng.gamma(0.5, 2.0, size=n) - 0.3  # negative, tiny and large values
    a[rng.random(n) < 0.2] = 0.0           # heavy repetition (dry shapes)
    if n > 20:
        a[rng.random(n) < 0.1] = np.nan
        a[3], a[7] = -0.0, 1e300
    ref = pd.Series(a).quantile(q)
    got = ctx.quantile(ctx.upload(a.reshape(rows, -1)), q)
    assert (np.isnan(ref) and np.isnan(got)) or got == pytest.approx(ref, rel=1e-14, abs=0.0)


def test_quantile_edge_cases(ctx):
    assert np.isnan(ctx.quantile(ctx.upload(np.full((2, 3), np.nan)), 0.5))
    a = np.array([[np.nan, 2.0, np.nan, -np.inf, np.inf, 2.0, 2.0, -5.0]])
    for q in (0.0, 0.2, 0.5, 0.8, 1.0):
        ref = pd.Series(a.ravel()).quantile(q)
        got = ctx.quantile(ctx.upload(a), q)
        assert (np.isnan(ref) and np.isnan(got)) or got == ref, (q, got, ref)
    with pytest.raises(ValueError, match="percentiles should all be in the interval"):
        ctx.quantile(ctx.upload(a), 1.5)
    # a pitched block (rows ld elements apart): only the rows' own elements count
    b = np.arange(60, dtype=np.float64).reshape(4, 15)
    d = ctx.upload(b, ld=16)
    assert d.ld == 16 and ctx.quantile(d, 0.5) == pd.Series(b.ravel()).quantile(0.5)


def test_zero_below_and_normalize_rows(ctx):
    rng = np.random.default_rng(5)
    a = rng.normal(size=(6, 400))
    a[rng.random(a.shape) < 0.05] = np.nan
    got = ctx.zero_below(ctx.upload(a), 0.25).numpy()
    np.testing.assert_array_equal(got, np.where(a >= 0.25, a, 0.0))
    mask = rng.random(400) < 0.6
    ref = np.array([3.0, np.nan, 0.0, 7.5, -2.0, 1.0])
    want = a * (ref / np.nansum(a[:, mask], axis=1))[:, None]
    close(ctx.normalize_rows(ctx.upload(a), mask, ref).numpy(), want)


@pytest.mark.parametrize("chunked", [False, True])
def test_runoff_chain_at_config5_result_size(ctx, chunked):
    """Cutout.runoff(smooth=True, lower_threshold_quantile=True, normalize_using_yearly=...) on a 4-year hourly series
    of 50 shapes (the (50 x 35040) result of BASELINE configs[4]) over a tiny grid: the post-processing runs on the device
    before the download (the gateway hook) and equals the oracle's pandas restatement."""
    T, Y, X, N = 35040, 3, 4, 50
    rng = np.random.default_rng(8)
    t = pd.date_range("2011-01-01", periods=T, freq="h")
    ro = rng.gamma(0.3, 1e-4, size=(T, Y, X))
    ro[rng.random((T, Y, X)) < 0.01] = np.nan
    height = rng.uniform(0.0, 2000.0, size=(Y, X))
    M = sp.random(N, Y * X, density=0.4, random_state=3, format="csr")
    names = [f"c{i:02d}" for i in range(N)]
    yearly = pd.DataFrame(rng.uniform(1.0, 9.0, size=(5, N)), index=["2010", "2011", "2012", "2013", "2014"], columns=names)
    yearly = yearly.drop(columns=["c07"])  # a shape without a reported total: NaN row, like the reference's reindex
    c = Cutout(Dataset({"runoff": ro, "height": height}, dict(time=t, y=np.arange(Y, dtype=float), x=np.arange(X, dtype=float)),
                       chunked=chunked))
    kw = dict(smooth=True, lower_threshold_quantile=True, normalize_using_yearly=yearly)
    r = c.runoff(matrix=M, index=pd.Index(names, name="countries"), aggregate_time=None, **kw)
    assert r.dims == (("time", "countries") if chunked else ("countries", "time"))
    got = np.asarray(r.values).T if chunked else np.asarray(r.values)
    series = orc.aggregate_matrix(orc.convert_runoff(ro, height[None]).reshape(T, -1), M)
    ref = orc.runoff_postprocess(series, t, names, **kw)
    close(got, ref)
    assert np.isnan(got[7]).all() and np.isfinite(got[8]).all() and (got == 0.0).any()
    # the same through the other door: per_unit results reach the host first, the device routines run on an upload
    r2, cap = c.runoff(matrix=M, index=pd.Index(names, name="countries"), aggregate_time=None, per_unit=True,
                       return_capacity=True, smooth=48)
    pu = series / np.asarray(M.sum(1)).ravel()[:, None]
    pu = np.where(np.isnan(pu), 0.0, pu)
    ref2 = orc.runoff_postprocess(pu, t, names, smooth=48)
    got2 = np.asarray(r2.values).T if chunked else np.asarray(r2.values)
    close(got2, ref2)


@pytest.mark.parametrize("chunked", [False, True])
def test_per_cell_runoff_is_smoothed_and_thresholded_without_shapes(ctx, chunked):
    """Cutout.runoff(smooth=..., lower_threshold_quantile=...) WITHOUT shapes or a matrix: the result is the per-cell
    (time, y, x) cube, and the reference's post-processing handles any rank - rolling(time=w, min_periods=1).mean() runs along
    time for every cell, the threshold is the quantile of ALL the cube's values (atlite/convert.py:1046-1062).  The cube
    reaches the host first; the same device routines then run on an upload, every cell a row."""
    T, Y, X = 400, 5, 7
    rng = np.random.default_rng(21)
    t = pd.date_range("2013-03-01", periods=T, freq="h")
    ro = rng.gamma(0.3, 1e-4, size=(T, Y, X))
    ro[rng.random((T, Y, X)) < 0.02] = np.nan
    height = rng.uniform(0.0, 2000.0, size=(Y, X))
    c = Cutout(Dataset({"runoff": ro, "height": height}, dict(time=t, y=np.arange(Y, dtype=float), x=np.arange(X, dtype=float)),
                       chunked=chunked))
    cube = orc.convert_runoff(ro, height[None])  # (T, Y, X)
    rows = cube.reshape(T, Y * X).T               # every cell a row, time last: what runoff_postprocess restates
    for kw in (dict(smooth=24), dict(smooth=True, lower_threshold_quantile=0.3), dict(lower_threshold_quantile=True)):
        r = c.runoff(aggregate_time=None, **kw)
        assert r.dims == ("time", "y", "x") and r.shape == (T, Y, X), (r.dims, r.shape)
        ref = orc.runoff_postprocess(rows, t, None, **kw).T.reshape(T, Y, X)
        close(np.asarray(r.values), ref)
        if "lower_threshold_quantile" in kw:
            assert (np.asarray(r.values) == 0.0).any()
    # the yearly normalisation needs rows with labels: refused for a per-cell cube, with a message that says what to pass
    with pytest.raises(ValueError, match="shapes or a matrix"):
        c.runoff(aggregate_time=None, normalize_using_yearly=pd.DataFrame({"a": [1.0]}, index=["2013"]))


def test_small_device_blocks_are_recycled_and_results_arrive_in_pinned_memory(ctx):
    """Context.empty / DeviceArray.free recycle small blocks by size (hipMalloc + hipFree cost more than a warm result
    download); DeviceArray.numpy() hands back page-locked memory for results of 64 KiB .. 1 GiB - same values either way."""
    from atlite_amd import device

    a = np.arange(20000.0).reshape(100, 200)
    d = ctx.upload(a)
    ptr = d.ptr
    h = d.numpy()
    assert isinstance(h.base, device._PinnedBlock) and np.array_equal(h, a)
    h[0, 0] = -1.0  # an ordinary writeable array
    del d
    e = ctx.empty((200, 100))  # the same number of bytes: the block that was just released
    e2 = ctx.empty((200, 100))
    import os

    if ctx._pool_state()["on"]:  # (off under ATLITE_HIP_RECYCLE=0 and the fenced allocator)
        assert e.ptr == ptr and e2.ptr != ptr
    else:
        assert e.ptr != ptr
    tiny = ctx.upload(np.ones(5))
    assert tiny.numpy().base is None
    os.environ["ATLITE_HIP_PINNED_RESULTS"] = "0"
    try:
        assert ctx.upload(a).numpy().base is None
    finally:
        del os.environ["ATLITE_HIP_PINNED_RESULTS"]
