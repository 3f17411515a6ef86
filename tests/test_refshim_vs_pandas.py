"""
The xarray stand-in that the golden generator runs the reference under (tests/golden/refshim.py) against an
INDEPENDENT implementation of the same labelled-array semantics: pandas (installed here; xarray documents its
where / fillna / clip / skipna reductions / rolling / resample as pandas-compatible and delegates the last two to
it).  Random cubes with NaN, +-inf, gaps in the time axis and partial days.  This pins the stand-in's reading of
every operation the reference's converters use on data (the golden vectors are only as good as it).
"""
import numpy as np
import pandas as pd
import pytest

from tests import helpers as H


@pytest.fixture(scope="module")
def xr():
    return H.xarray_stand_in()


def cube(rng, T, Y, X, start="2013-01-01 05:00", freq="h", holes=0.1):
    t = pd.date_range(start, periods=T, freq=freq)
    v = rng.standard_normal((T, Y, X)) * 10.0
    v[rng.random(v.shape) < holes] = np.nan
    v[rng.random(v.shape) < 0.01] = np.inf
    v[rng.random(v.shape) < 0.01] = -np.inf
    return t, v


def frame(t, v):
    return pd.DataFrame(v.reshape(len(t), -1), index=t)


def da_of(xr, t, v, Y, X):
    return xr.DataArray(v, coords={"time": t, "y": np.arange(Y, dtype=float), "x": np.arange(X, dtype=float)},
                        dims=["time", "y", "x"], name="v")


@pytest.mark.parametrize("seed", range(4))
def test_elementwise_semantics(xr, seed):
    rng = np.random.default_rng(seed)
    T, Y, X = 30, 3, 4
    t, v = cube(rng, T, Y, X)
    _, w = cube(rng, T, Y, X)
    da, db = da_of(xr, t, v, Y, X), da_of(xr, t, w, Y, X)
    fa, fb = frame(t, v), frame(t, w)
    eq = lambda got, ref: np.testing.assert_array_equal(np.asarray(got.values).reshape(T, -1), ref.to_numpy())  # noqa: E731
    with np.errstate(all="ignore"):
        eq(da.where(db > 0), fa.where(fb > 0))
        eq(da.where(db > 0, 0.0), fa.where(fb > 0, 0.0))
        eq(da.where(db > 0, db), fa.where(fb > 0, fb))
        eq(da.fillna(0.0), fa.fillna(0.0))
        eq(da.fillna(db), fa.fillna(fb))
        eq(da.clip(min=0.0), fa.clip(lower=0.0))
        eq(da.clip(max=5.0), fa.clip(upper=5.0))
        eq(da.clip(min=-2.0, max=3.0), fa.clip(lower=-2.0, upper=3.0))
        # array bounds: xarray's clip IS np.clip (apply_ufunc(np.clip, self, min, max)) - a NaN bound gives NaN, where
        # pandas would ignore the bound; with NaN-free bounds the two libraries agree
        np.testing.assert_array_equal(np.asarray(da.clip(max=db).values), np.clip(v, None, w))
        np.testing.assert_array_equal(np.asarray(da.clip(min=0.0, max=db).values), np.clip(v, 0.0, w))
        eq(da.clip(min=0.0, max=db.fillna(0.0).clip(min=0.0)), fa.clip(lower=0.0, upper=fb.fillna(0.0).clip(lower=0.0), axis=None))
        eq(da * db + 2.0 - db / da, fa * fb + 2.0 - fb / fa)
        eq(np.sin(da) ** 2, np.sin(fa) ** 2)
        eq((da > db) | (da < 0), (fa > fb) | (fa < 0))
        eq(-da, -fa)
        eq(abs(da), abs(fa))


@pytest.mark.parametrize("seed", range(4))
def test_reductions_skip_nan_like_pandas(xr, seed):
    rng = np.random.default_rng(10 + seed)
    T, Y, X = 25, 3, 5
    t, v = cube(rng, T, Y, X, holes=0.3)
    v[:, 0, 0] = np.nan  # an all-NaN series
    da, fa = da_of(xr, t, v, Y, X), frame(t, v)
    with np.errstate(all="ignore"):
        np.testing.assert_allclose(np.asarray(da.sum("time").values).ravel(), fa.sum(axis=0, skipna=True).to_numpy(), rtol=1e-13)
        np.testing.assert_allclose(np.asarray(da.mean("time").values).ravel(), fa.mean(axis=0, skipna=True).to_numpy(), rtol=1e-13,
                                   equal_nan=True)
    assert np.asarray(da.sum("time").values)[0, 0] == 0.0 and np.isnan(np.asarray(da.mean("time").values)[0, 0])


@pytest.mark.parametrize("window,minp", [(24, 1), (3, None), (7, 2), (48, 1)])
def test_rolling_mean_is_pandas_rolling(xr, window, minp):
    rng = np.random.default_rng(window)
    T, Y, X = 80, 2, 3
    t, v = cube(rng, T, Y, X, holes=0.2)
    v[~np.isfinite(v)] = np.nan  # (pandas' rolling mean is an online sum: inf would poison later windows there)
    da, fa = da_of(xr, t, v, Y, X), frame(t, v)
    got = da.rolling(time=window, min_periods=minp).mean()
    ref = fa.rolling(window, min_periods=minp).mean()
    np.testing.assert_allclose(np.asarray(got.values).reshape(T, -1), ref.to_numpy(), rtol=1e-12, atol=1e-12, equal_nan=True)


@pytest.mark.parametrize("start", ["2013-01-01 00:00", "2013-01-01 05:00", "2012-12-31 23:00"])
def test_resample_daily_mean_is_pandas_resample(xr, start):
    rng = np.random.default_rng(len(start))
    T, Y, X = 100, 2, 3
    t, v = cube(rng, T, Y, X, start=start, holes=0.2)
    v[~np.isfinite(v)] = np.nan
    keep = np.ones(T, bool)
    keep[30:57] = False  # a gap of more than a day: an empty bin in between
    t, v = t[keep], v[keep]
    da, fa = da_of(xr, t, v, Y, X), frame(t, v)
    got = da.resample(time="1D").mean()
    ref = fa.resample("1D").mean()
    assert list(pd.DatetimeIndex(got.coords["time"].values)) == list(ref.index)
    np.testing.assert_allclose(np.asarray(got.values).reshape(len(ref), -1), ref.to_numpy(), rtol=1e-13, equal_nan=True)


def test_reindex_like_and_time_fields(xr):
    rng = np.random.default_rng(3)
    T, Y, X = 48, 2, 2
    t, v = cube(rng, T, Y, X, holes=0.0)
    da = da_of(xr, t, v, Y, X)
    daily = da.resample(time="1D").mean()
    back = daily.reindex_like(da)  # values at the day labels, NaN elsewhere: pandas' reindex
    ref = frame(t, v).resample("1D").mean().reindex(t)
    np.testing.assert_allclose(np.asarray(back.values).reshape(T, -1), ref.to_numpy(), equal_nan=True, rtol=1e-13)
    np.testing.assert_array_equal(np.asarray(da.coords["time"].dt.hour.values), t.hour)
    np.testing.assert_array_equal(np.asarray(da.coords["time"].dt.minute.values), t.minute)
