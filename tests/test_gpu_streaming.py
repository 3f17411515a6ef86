"""GPU: the slab pipeline for host-resident cutouts gives the same results as the whole-dataset
launch (and as the oracle), for every converter family, ragged last slabs, day-aligned heat-demand
slabs, the in-kernel solar position tables and per-cell orientation."""
import numpy as np
import pandas as pd
import pytest
import scipy.sparse as sp

from atlite_amd import Cutout, Dataset
from oracle import atlite_oracle as orc
from tests import helpers as H

pytestmark = pytest.mark.gpu


def close(a, b, s=1e-12):
    np.testing.assert_allclose(a, b, rtol=1e-10, atol=s * np.nanmax(np.abs(b)), equal_nan=True)


@pytest.fixture
def forced(monkeypatch):
    monkeypatch.setenv("ATLITE_HIP_STREAM", "1")
    monkeypatch.setenv("ATLITE_HIP_SLAB_STEPS", "16")  # 61 steps -> 4 slabs, the last one ragged


def both(monkeypatch, fn):
    monkeypatch.setenv("ATLITE_HIP_STREAM", "0")
    whole = fn()
    monkeypatch.setenv("ATLITE_HIP_STREAM", "1")
    return whole, fn()


def test_pv_streamed(monkeypatch, forced):
    T, Y, X, N = 61, 6, 10, 4
    ds = H.pv_dataset(T, Y, X, seed=3)
    x, y = H.grid(Y, X)
    M = H.blob_matrix(N, Y, X, seed=4)
    c = Cutout(Dataset(ds, dict(time=H.times(T), y=y, x=x)))
    ori = dict(slope=np.radians(30.0), azimuth=np.radians(180.0))
    ref = orc.aggregate_matrix(orc.convert_pv(ds, H.CSI, ori), M)
    kw = dict(panel="CSi", orientation={"slope": 30.0, "azimuth": 180.0})
    for agg in (None, "sum", "mean"):
        w, s = both(monkeypatch, lambda: c.pv(matrix=M, aggregate_time=agg, **kw).values)
        np.testing.assert_array_equal(w, s) if agg is None else close(s, w)
        close(s, orc.aggregate_time(ref, agg, 1))
    w, s = both(monkeypatch, lambda: c.pv(aggregate_time=None, **kw).values)
    np.testing.assert_array_equal(w, s)
    w, s = both(monkeypatch, lambda: c.pv(aggregate_time="sum", **kw).values)
    close(s, w)
    # per-cell orientation + general kernel (tracking)
    w, s = both(monkeypatch, lambda: c.pv(panel="CSi", orientation="latitude_optimal", matrix=M, aggregate_time=None).values)
    np.testing.assert_array_equal(w, s)
    w, s = both(monkeypatch, lambda: c.pv(tracking="horizontal", matrix=M, aggregate_time=None, **kw).values)
    np.testing.assert_array_equal(w, s)
    # in-kernel solar position: tables sliced per slab
    ds5 = {k: v for k, v in ds.items() if not k.startswith("solar_")}
    c5 = Cutout(Dataset(ds5, dict(time=H.times(T), y=y, x=x)))
    with pytest.warns(DeprecationWarning):
        w, s = both(monkeypatch, lambda: c5.pv(matrix=M, aggregate_time=None, **kw).values)
    np.testing.assert_array_equal(w, s)


def test_wind_heat_runoff_streamed(monkeypatch, forced):
    T, Y, X, N = 24 * 4 + 5, 5, 8, 3
    x, y = H.grid(Y, X)
    t = H.times(T)
    rng = np.random.default_rng(1)
    wd = H.wind_dataset(T, Y, X, seed=2)
    ds = dict(wnd100m=wd["wnd100m"], roughness=wd["roughness"], temperature=283 + 8 * rng.standard_normal((T, Y * X)),
              runoff=rng.random((T, Y * X)), height=1000 * rng.random(Y * X))
    c = Cutout(Dataset(ds, dict(time=t, y=y, x=x)))
    M = H.blob_matrix(N, Y, X, seed=5)
    w, s = both(monkeypatch, lambda: c.wind(turbine="Vestas_V112_3MW", matrix=M, aggregate_time=None).values)
    np.testing.assert_array_equal(w, s)
    w, s = both(monkeypatch, lambda: c.runoff(matrix=M, aggregate_time=None).values)
    np.testing.assert_array_equal(w, s)
    w, s = both(monkeypatch, lambda: c.temperature(matrix=M, aggregate_time="mean").values)
    close(s, w)
    for shift in (0.0, 5.0):
        monkeypatch.setenv("ATLITE_HIP_SLAB_STEPS", "30")  # slabs of one calendar day each
        w, s = both(monkeypatch, lambda: c.heat_demand(hour_shift=shift, matrix=M, aggregate_time=None).values)
        np.testing.assert_array_equal(w, s)
        w, s = both(monkeypatch, lambda: c.heat_demand(hour_shift=shift, aggregate_time=None).values)
        np.testing.assert_array_equal(w, s)
        ptr, _ = orc.day_groups(t, shift)
        close(s.reshape(len(ptr) - 1, -1), orc.convert_heat_demand(ds["temperature"], ptr), 1e-9)
    # arbitrary convert_func on host data: cube streamed through atl_spmm_csr
    w, s = both(monkeypatch, lambda: c.convert_and_aggregate(lambda d: d["runoff"], matrix=M, aggregate_time=None).values)
    np.testing.assert_array_equal(w, s)


def test_time_dependent_orientation_streamed_and_sharded(monkeypatch, forced):
    """An orientation callback that follows the sun produces (time, cell) slope / azimuth cubes: they are cut along
    time with the inputs, by the slab pipeline and by the multi-device executor alike - same bits as one launch."""
    T, Y, X, N = 61, 6, 10, 4
    ds = H.pv_dataset(T, Y, X, seed=13)
    x, y = H.grid(Y, X)
    M = H.blob_matrix(N, Y, X, seed=14)
    kw = dict(panel="CSi", orientation=H.orientation_follow_sun, matrix=M, aggregate_time=None)
    c = Cutout(Dataset(ds, dict(time=H.times(T), y=y, x=x)))
    whole, streamed = both(monkeypatch, lambda: c.pv(**kw).values)
    np.testing.assert_array_equal(whole, streamed)
    sp_ = dict(altitude=ds["solar_altitude"].reshape(T, Y, X), azimuth=ds["solar_azimuth"].reshape(T, Y, X))

    class A:
        def __init__(self, v, dims=None, coords=None):
            self.values, self.dims, self.coords = np.asarray(v), dims, coords

    o = H.orientation_follow_sun(None, None, {k: A(v) for k, v in sp_.items()})
    ref = orc.convert_pv_general(ds, H.CSI, dict(slope=o["slope"].values.reshape(T, -1), azimuth=o["azimuth"].values.reshape(T, -1)))
    close(whole, orc.aggregate_matrix(ref, M))
    monkeypatch.setenv("ATLITE_HIP_STREAM", "0")
    many = Cutout(Dataset(ds, dict(time=H.times(T), y=y, x=x)), devices=[0, 0, 0])
    np.testing.assert_array_equal(many.pv(**kw).values, whole)

