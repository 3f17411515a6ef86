"""GPU: atlite.wind.extrapolate_wind_speed as an operation of its own (atlite/wind.py:23-125) - the wind converter
without a power curve (atl_wind_params.n_knots = 0) through the C ABI and through atlite_amd.wind."""
import numpy as np
import pytest

from atlite_amd import Dataset, wind
from oracle import atlite_oracle as orc
from tests import helpers as H

pytestmark = pytest.mark.gpu


def close(a, b):
    b = np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=1e-10, atol=1e-12 * np.nanmax(np.abs(b[np.isfinite(b)])), equal_nan=True)


@pytest.mark.parametrize("method,aux", [("logarithmic", "roughness"), ("power", "wnd_shear_exp"), (None, None)])
def test_speed_through_the_c_abi(ctx, method, aux):
    T, Y, X, N = 37, 7, 19, 4
    ds = H.wind_dataset(T, Y, X, seed=2)
    w, a = ds["wnd100m"].copy(), None if aux is None else ds[aux].copy()
    w[0, :6] = [0.0, -3.0, np.nan, np.inf, 1e-300, 1e6]
    if a is not None:
        a[1, :8] = [0.0, -1.0, np.nan, np.inf, 100.0, np.nextafter(100.0, 0), 80.0, 1e-320]
    dw, da = ctx.upload(w), None if a is None else ctx.upload(a)
    with np.errstate(all="ignore"):
        ref = orc.extrapolate_wind_speed(w, a, 80.0, 100.0, method)
    close(ctx.wind(dw, da, None, None, 80.0, 100.0, method, T, Y * X).numpy(), ref)
    # finite data: time reduction and the fused aggregation of the same quantity
    w2, a2 = ds["wnd100m"], None if aux is None else ds[aux]
    dw, da = ctx.upload(w2), None if a2 is None else ctx.upload(a2)
    ref = orc.extrapolate_wind_speed(w2, a2, 80.0, 100.0, method)
    close(ctx.wind(dw, da, None, None, 80.0, 100.0, method, T, Y * X, time_agg="mean").numpy(), ref.mean(0))
    M = H.blob_matrix(N, Y, X, seed=3)
    close(ctx.wind(dw, da, None, None, 80.0, 100.0, method, T, Y * X, plan=ctx.plan(M, row_len=X)).numpy(), M @ ref.T)
    if method is not None:
        with pytest.raises(ValueError, match="positive and finite"):
            ctx.wind(dw, da, None, None, -80.0, 100.0, method, T, Y * X)


def test_extrapolate_wind_speed_api():
    T, Y, X = 30, 5, 8
    raw = H.wind_dataset(T, Y, X, seed=4)
    x, y = H.grid(Y, X)
    ds = Dataset(dict(raw, wnd10m=0.7 * raw["wnd100m"]), dict(time=H.times(T), y=y, x=x))
    r = wind.extrapolate_wind_speed(ds, 80)
    assert r.name == "wnd80m" and r.dims == ("time", "y", "x") and r.attrs["units"] == "m s**-1"
    assert r.attrs["long name"] == "extrapolated 80 m wind speed using logarithmic method with roughness  and 100 m wind speed"
    close(np.asarray(r.values).reshape(T, -1), orc.extrapolate_wind_speed(raw["wnd100m"], raw["roughness"], 80, 100, "logarithmic"))
    p = wind.extrapolate_wind_speed(ds, 120.5, method="power")
    close(np.asarray(p.values).reshape(T, -1), orc.extrapolate_wind_speed(raw["wnd100m"], raw["wnd_shear_exp"], 120.5, 100, "power"))
    low = wind.extrapolate_wind_speed(ds, 30)  # the closest stored height is 10 m
    close(np.asarray(low.values).reshape(T, -1), orc.extrapolate_wind_speed(0.7 * raw["wnd100m"], raw["roughness"], 30, 10, "logarithmic"))
    forced = wind.extrapolate_wind_speed(ds, 30, from_height=100)
    close(np.asarray(forced.values).reshape(T, -1), orc.extrapolate_wind_speed(raw["wnd100m"], raw["roughness"], 30, 100, "logarithmic"))
    assert wind.extrapolate_wind_speed(ds, 100) is ds["wnd100m"]  # fast lane
    # the fast lane also holds for a spec constructed directly, whatever from_height says, and keeps the stored attrs
    from atlite_amd import convert as cv

    ds["wnd100m"].attrs = {"units": "m s**-1", "long_name": "stored 100 metre wind speed"}
    spec = cv._WindSpeedSpec(ds, 100, from_height=10)
    assert spec.method is None and spec.wnd == "wnd100m" and spec.attrs["long_name"] == "stored 100 metre wind speed"
    lane = cv._finish(cv._per_cell(spec, ds))
    np.testing.assert_array_equal(np.asarray(lane.values), raw["wnd100m"].reshape(T, Y, X))
    with pytest.raises(ValueError, match="Interpolation method must be 'logarithmic' or 'power'"):
        wind.extrapolate_wind_speed(ds, 80, method="cubic")
    bare = Dataset({"wnd100m": raw["wnd100m"]}, dict(time=H.times(T), y=y, x=x))
    with pytest.raises(RuntimeError, match="requires surface roughness"):
        wind.extrapolate_wind_speed(bare, 80)
    with pytest.raises(RuntimeError, match="requires a wind shear exponent"):
        wind.extrapolate_wind_speed(bare, 80, method="power")
    with pytest.raises(AssertionError, match="Wind speed is not in dataset"):
        wind.extrapolate_wind_speed(Dataset({"roughness": raw["roughness"]}, dict(time=H.times(T), y=y, x=x)), 80)
