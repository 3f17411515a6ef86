"""
GPU: grids whose cell count or row length is odd - every real-world ERA5 cutout with integer-degree bounds (x and y
both hold 4 (b - a) + 1 points) - run the VECTORISED kernels since round 3: the lane's two cells as one 16-byte access
that is only 8-byte aligned in every other slot, the lane that owns the last cell storing a single value, cell pairs
that straddle two grid rows owned by flat index.  Checked against the oracle, against the unvectorised instantiations
($ATLITE_HIP_NO_VEC; same arithmetic, so the same bits), and on a cube that ends on a 4 KiB page (where the launch must
not read past it: the unvectorised kernel takes it).  Reference: the conversions of atlite/convert.py on any grid.
"""
import numpy as np
import pandas as pd
import pytest
import scipy.sparse as sp

from oracle import atlite_oracle as orc
from tests import helpers as H

pytestmark = pytest.mark.gpu
PV = dict(H.CSI, slope=np.radians(30.0), azimuth=np.radians(180.0))
ORI = dict(slope=np.radians(30.0), azimuth=np.radians(180.0))


def close(a, b, atol_scale=1e-12):
    a, b = np.asarray(a), np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=1e-10, atol=atol_scale * max(float(np.nanmax(np.abs(b))), 1e-300), equal_nan=True)


def both(monkeypatch, fn):
    """fn() through the vectorised kernels and through the unvectorised ones: identical bits."""
    monkeypatch.delenv("ATLITE_HIP_NO_VEC", raising=False)
    a = fn()
    monkeypatch.setenv("ATLITE_HIP_NO_VEC", "1")
    b = fn()
    monkeypatch.delenv("ATLITE_HIP_NO_VEC")
    np.testing.assert_array_equal(a, b)
    return a


# (odd S, odd X) / (even S, odd X) / (odd S, X a multiple of the line) / one column / rows shorter than a line
GRIDS = [(37, 9, 27), (40, 12, 27), (33, 7, 33), (29, 11, 1), (50, 21, 5), (26, 3, 129)]


@pytest.mark.parametrize("T,Y,X", GRIDS)
def test_pv_on_odd_grids(ctx, monkeypatch, T, Y, X):
    S = Y * X
    ds = H.pv_dataset(T, Y, X, seed=T)
    dev = {k: ctx.upload(v) for k, v in ds.items()}
    M = H.blob_matrix(4, Y, X, seed=3)
    cells = orc.convert_pv(ds, H.CSI, ORI)
    for skip in (False, True):
        out = both(monkeypatch, lambda: ctx.pv(dev, PV, T, S, options=dict(night_skip=skip, row_len=X)).numpy())
        close(out, cells)
        agg = both(monkeypatch, lambda: ctx.pv(dev, PV, T, S, plan=ctx.plan(M, row_len=X), options=dict(night_skip=skip)).numpy())
        close(agg, orc.aggregate_matrix(cells, M))
        mean = both(monkeypatch, lambda: ctx.pv(dev, PV, T, S, time_agg="mean", options=dict(night_skip=skip, row_len=X)).numpy())
        close(mean, cells.mean(axis=0))
    # the members of the family that exist vectorised only (here: a tracker): the general kernel takes NO_VEC launches
    trk = ctx.pv(dev, PV, T, S, plan=ctx.plan(M, row_len=X), options=dict(tracking="horizontal")).numpy()
    monkeypatch.setenv("ATLITE_HIP_NO_VEC", "1")
    gen = ctx.pv(dev, PV, T, S, plan=ctx.plan(M, row_len=X), options=dict(tracking="horizontal")).numpy()
    monkeypatch.delenv("ATLITE_HIP_NO_VEC")
    close(trk, gen)


@pytest.mark.parametrize("T,Y,X", GRIDS[:4])
def test_wind_heat_runoff_on_odd_grids(ctx, monkeypatch, T, Y, X):
    S = Y * X
    w = H.wind_dataset(T, Y, X, seed=2)
    V = np.array([0, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 25, 25], dtype=float)
    POW = np.array([0.0, 0.0, 0.005, 0.15, 0.3, 0.525, 0.905, 1.375, 1.95, 2.58, 2.96, 3.05, 3.06, 3.06, 0.0])
    M = H.blob_matrix(3, Y, X, seed=5)
    dw, dz = ctx.upload(w["wnd100m"]), ctx.upload(w["roughness"])
    ref = orc.convert_wind(w["wnd100m"], w["roughness"], V, POW, 3.06, 80.0, 100.0)
    args = (dw, dz, V, POW / 3.06, 80.0, 100.0, "logarithmic", T, S)
    close(both(monkeypatch, lambda: ctx.wind(*args).numpy()), ref)
    close(both(monkeypatch, lambda: ctx.wind(*args, plan=ctx.plan(M, row_len=X)).numpy()), orc.aggregate_matrix(ref, M))
    close(both(monkeypatch, lambda: ctx.wind(*args, time_agg="mean").numpy()), ref.mean(axis=0))
    rng = np.random.default_rng(T)
    ro, h = rng.random((T, S)), rng.random(S) * 900.0
    dro, dh = ctx.upload(ro), ctx.upload(h)
    close(both(monkeypatch, lambda: ctx.runoff(dro, dh, T, S).numpy()), ro * h[None, :])
    close(both(monkeypatch, lambda: ctx.runoff(dro, dh, T, S, plan=ctx.plan(M, row_len=X)).numpy()), np.asarray(M @ (ro * h[None, :]).T))
    temp = 270.0 + 25.0 * rng.random((T, S))
    day_ptr = np.append(np.arange(0, T, 24), T)
    refh = orc.convert_heat_demand(temp, day_ptr, threshold=15.0, a=1.2, constant=0.1)
    close(both(monkeypatch, lambda: ctx.heat_demand(ctx.upload(temp), day_ptr, 288.15, 1.2, 0.1, T, S).numpy()), refh, atol_scale=1e-9)
    D = rng.normal(size=(T, S))
    close(both(monkeypatch, lambda: ctx.spmm(ctx.plan(M, row_len=X), ctx.upload(D)).numpy()), np.asarray(M @ D.T))


def test_a_cube_that_ends_on_a_page_boundary_is_not_read_past(ctx):
    """Odd cell count and T x S x 8 a multiple of 4096 (T = 512): the vectorised launch would read the 8 bytes after the
    cube in the last slot - the rule in vec_ok() hands such a launch to the unvectorised kernel.  (A fault cannot be
    provoked portably; the result must be right either way.)"""
    T, Y, X = 512, 3, 3
    S = Y * X
    assert (T * S * 8) % 4096 == 0
    rng = np.random.default_rng(0)
    ro, h = rng.random((T, S)), rng.random(S)
    out = ctx.runoff(ctx.upload(ro), ctx.upload(h), T, S).numpy()
    np.testing.assert_allclose(out, ro * h[None, :], rtol=1e-15)
    M = sp.csr_matrix(np.ones((1, S)))
    agg = ctx.runoff(ctx.upload(ro), ctx.upload(h), T, S, plan=ctx.plan(M, row_len=X)).numpy()
    np.testing.assert_allclose(agg[0], (ro * h[None, :]).sum(axis=1), rtol=1e-13)


# ---- padded slots: the library's own device copies of a cutout ---------------------------------------------------
def test_padded_device_copies_round_trip_and_layout(ctx):
    from atlite_amd.device import pitch_for

    assert pitch_for(40000) is None and pitch_for(29673) == 29680 and pitch_for(17) == 32
    rng = np.random.default_rng(0)
    a = rng.normal(size=(13, 37))
    d = ctx.upload(a, ld=pitch_for(37))
    assert d.ld == 48 and d.shape == (13, 37)
    np.testing.assert_array_equal(d.numpy(), a)
    np.testing.assert_array_equal(d.slab(3, 9).numpy(), a[3:9])
    assert d.slab(3, 9).ld == 48 and d.reshape(13, 37) is d
    with pytest.raises(ValueError, match="pitched"):
        d.reshape(37, 13)
    assert ctx.upload(rng.normal(size=(4, 32)), ld=32).ld is None  # nothing to pad


@pytest.mark.parametrize("T,Y,X", [(40, 9, 27), (50, 13, 31)])
def test_api_results_with_padded_slots_equal_contiguous_ones(monkeypatch, T, Y, X):
    """Cutout.pv / wind / heat_demand / runoff on host arrays: Dataset.device() pads the slots of its device copies when
    Y * X is not a multiple of 16; per-cell results carry the same bits as with ATLITE_HIP_PITCH=0, aggregated ones
    agree to rounding (the plan picks another tile shape, so shapes sum their cells in another order)."""
    import pandas as pd

    from atlite_amd import Cutout, Dataset

    x, y = H.grid(Y, X)
    t = pd.date_range("2013-03-01", periods=T, freq="h")
    ds = H.pv_dataset(T, Y, X, seed=3)
    w = H.wind_dataset(T, Y, X, seed=4)
    rng = np.random.default_rng(5)
    data = {k: v.reshape(T, Y, X) for k, v in {**ds, **w}.items()}
    data["runoff"] = rng.random((T, Y, X))
    data["height"] = rng.random((Y, X)) * 500.0
    M = H.blob_matrix(4, Y, X, seed=6)

    def run():
        c = Cutout(Dataset(dict(data), dict(time=t, y=y, x=x)))
        kw = dict(panel="CSi", orientation={"slope": 30.0, "azimuth": 180.0})
        out = dict(
            pv_cells=c.pv(aggregate_time=None, **kw).values, pv_agg=c.pv(matrix=M, aggregate_time=None, **kw).values,
            pv_map=c.pv(aggregate_time="mean", **kw).values,
            wind_cells=c.wind(turbine="Vestas_V112_3MW", aggregate_time=None).values,
            wind_agg=c.wind(turbine="Vestas_V112_3MW", matrix=M, aggregate_time=None).values,
            heat=c.heat_demand(matrix=M, aggregate_time=None).values, runoff=c.runoff(matrix=M, aggregate_time=None).values,
            runoff_cells=c.runoff(aggregate_time="sum").values)
        lds = {v.ld for v in c.data._device_cache.values() if v.shape[0] == T and v.size == T * Y * X}  # the time-dependent cubes
        return out, lds

    monkeypatch.delenv("ATLITE_HIP_PITCH", raising=False)
    monkeypatch.delenv("ATLITE_HIP_INTERLEAVE", raising=False)
    padded, lds = run()
    Sp = (Y * X + 15) // 16 * 16
    # (the cubes one conversion reads share a slot-interleaved allocation: 7 padded slots per time step for pv, 2 for wind)
    assert {ld % Sp for ld in lds} == {0} and 7 * Sp in lds and Sp in lds
    monkeypatch.setenv("ATLITE_HIP_PITCH", "0")
    monkeypatch.setenv("ATLITE_HIP_INTERLEAVE", "0")
    plain, lds0 = run()
    assert lds0 == {None}
    for k in padded:
        # (aggregations sum tile by tile, time reductions chunk by chunk - on contiguous cubes off the line grid the chunks
        #  follow the slots' alignment classes: the same terms in another order)
        if k.endswith("_agg") or k in ("heat", "runoff", "pv_map", "runoff_cells"):
            np.testing.assert_allclose(padded[k], plain[k], rtol=1e-12, atol=1e-13 * np.abs(plain[k]).max(), err_msg=k)
        else:
            np.testing.assert_array_equal(padded[k], plain[k], err_msg=k)
    cells = orc.convert_pv(ds, H.CSI, ORI)
    close(padded["pv_cells"].reshape(T, -1), cells)
    close(padded["pv_agg"], orc.aggregate_matrix(cells, M))


def test_cubes_of_one_call_must_share_their_layout(ctx):
    T, Y, X = 12, 5, 7
    S = Y * X
    w = H.wind_dataset(T, Y, X, seed=1)
    V = np.array([0.0, 3.0, 12.0, 25.0, 25.0])
    P = np.array([0.0, 0.0, 1.0, 1.0, 0.0])
    with pytest.raises(ValueError, match="mix slot strides"):
        ctx.wind(ctx.upload(w["wnd100m"], ld=48), ctx.upload(w["roughness"]), V, P, 80.0, 100.0, "logarithmic", T, S)
    ok = ctx.wind(ctx.upload(w["wnd100m"], ld=48), ctx.upload(w["roughness"], ld=48), V, P, 80.0, 100.0, "logarithmic", T, S).numpy()
    ref = ctx.wind(ctx.upload(w["wnd100m"]), ctx.upload(w["roughness"]), V, P, 80.0, 100.0, "logarithmic", T, S).numpy()
    # (bit for bit, the last cell included: the per-cell kernels tell the converter that the pad cell beside it does not
    # exist, so its zeros cannot send the pair through the wind converter's literal routine)
    np.testing.assert_array_equal(ok, ref)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_slab_pipeline_pads_its_buffers_too(monkeypatch, dtype):
    """ATLITE_HIP_STREAM=1 on an odd grid: the double buffers are padded (2-d DMA for fp64, the 2-d widening pass for
    float32 - what xarray hands over for a real cutout) and the result equals the device-resident run."""
    import pandas as pd

    from atlite_amd import Cutout, Dataset

    T, Y, X = 53, 9, 21
    x, y = H.grid(Y, X)
    t = pd.date_range("2013-01-01", periods=T, freq="h")
    w = H.wind_dataset(T, Y, X, seed=8)
    data = {k: v.reshape(T, Y, X).astype(dtype) for k, v in w.items()}
    M = H.blob_matrix(3, Y, X, seed=9)
    monkeypatch.setenv("ATLITE_HIP_SLAB_STEPS", "16")
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("ATLITE_HIP_STREAM", mode)
        c = Cutout(Dataset(dict(data), dict(time=t, y=y, x=x)))
        out[mode] = (c.wind(turbine="Vestas_V112_3MW", matrix=M, aggregate_time=None).values,
                     c.wind(turbine="Vestas_V112_3MW", aggregate_time=None).values)
    np.testing.assert_array_equal(out["0"][0], out["1"][0])
    np.testing.assert_array_equal(out["0"][1], out["1"][1])
    from atlite_amd.resource import get_windturbineconfig

    tb = get_windturbineconfig("Vestas_V112_3MW")
    ref = orc.convert_wind(data["wnd100m"].astype(np.float64).reshape(T, -1), data["roughness"].astype(np.float64).reshape(T, -1),
                           np.asarray(tb["V"], float), np.asarray(tb["POW"], float), tb["P"], tb["hub_height"], 100.0)
    close(out["1"][1].reshape(T, -1), ref)


def test_repack_of_caller_owned_cubes_off_the_line_grid(ctx):
    """Dataset(repack=True): (time, y, x) cubes the CALLER holds on the device, contiguous, with a cell count that is not a
    multiple of 16, are copied once into the library's padded slot-interleaved pool - later conversions read aligned
    slots.  Same values as the contiguous cubes give; without repack the caller's cubes are used where they lie."""
    from atlite_amd import Cutout, Dataset
    from atlite_amd.device import DeviceArray

    T, Y, X, N = 50, 9, 21, 4  # S = 189
    ds = H.pv_dataset(T, Y, X, seed=5)
    M = H.blob_matrix(N, Y, X, seed=6)
    t = pd.date_range("2013-03-01", periods=T, freq="h")
    x, y = H.grid(Y, X)
    dev = {k: ctx.upload(v.reshape(T, Y * X)) for k, v in ds.items()}  # contiguous device cubes: the caller's
    assert all(d.ld is None for d in dev.values())
    kw = dict(panel="CSi", orientation={"slope": 30.0, "azimuth": 180.0}, matrix=M, aggregate_time=None)
    plain = Cutout(Dataset(dict(dev), dict(time=t, y=y, x=x)))
    a = plain.pv(**kw).values
    assert all(c.ld is None for c in plain.data._device_cache.values() if isinstance(c, DeviceArray))
    packed = Cutout(Dataset(dict(dev), dict(time=t, y=y, x=x), repack=True))
    b = packed.pv(**kw).values
    cached = [c for c in packed.data._device_cache.values() if isinstance(c, DeviceArray) and c.ndim == 2]
    assert len(cached) == 7 and all(c.ld is not None and c.ld % 16 == 0 for c in cached)  # padded pool views
    assert len({c._pool for c in cached}) == 1
    # (the caller's contiguous cubes go through the line-aligned plan since round 4: the same products, summed tile by tile
    #  of another tiling - equal to rounding, no longer bit for bit)
    close(a, b, atol_scale=1e-14)
    np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-12 * np.abs(b).max())
    np.testing.assert_array_equal(packed.pv(**kw).values, b)  # second call: the resident copies
    w = packed.pv(panel="CSi", orientation={"slope": 30.0, "azimuth": 180.0}, aggregate_time="mean").values
    # (the time reduction over the contiguous cubes walks the slots' alignment classes: the same terms, another order)
    np.testing.assert_allclose(w, plain.pv(panel="CSi", orientation={"slope": 30.0, "azimuth": 180.0}, aggregate_time="mean").values,
                               rtol=1e-13, atol=1e-15)
