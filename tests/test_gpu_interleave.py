"""
GPU: the slot-interleaved residency of the library's own device copies (device.SlotPool, Dataset.device_group) - the cubes
one conversion reads lie in ONE allocation, the variables of a time step side by side; the kernels see nothing but a slot
stride (the ld_cells argument) and stream 5-9 % faster from it.  Same bits as with an allocation per cube
(ATLITE_HIP_INTERLEAVE=0); a replaced variable returns to its slot; a call that needs another set of cubes regroups on the
device; file-backed variables are inflated straight into their slots.
Reference: the variables a conversion reads, atlite/convert.py:529-562 (pv), :597-610 (wind).
"""
import numpy as np
import pandas as pd
import pytest

from oracle import atlite_oracle as orc
from tests import helpers as H

pytestmark = pytest.mark.gpu
ORI = dict(slope=np.radians(30.0), azimuth=np.radians(180.0))
KW = dict(panel="CSi", orientation={"slope": 30.0, "azimuth": 180.0})


def dataset(T, Y, X, seed=3):
    from atlite_amd import Dataset

    x, y = H.grid(Y, X)
    t = pd.date_range("2013-03-01", periods=T, freq="h")
    ds = H.pv_dataset(T, Y, X, seed=seed)
    w = H.wind_dataset(T, Y, X, seed=seed + 1)
    data = {k: v.reshape(T, Y, X) for k, v in {**ds, **w}.items()}
    rng = np.random.default_rng(seed + 2)
    data["runoff"] = rng.random((T, Y, X))
    data["height"] = rng.random((Y, X)) * 500.0
    return Dataset(data, dict(time=t, y=y, x=x)), ds, w


def pools(ds):
    return {id(p): p for p in (getattr(d, "_pool", None) for d in ds._device_cache.values()) if p is not None}


@pytest.mark.parametrize("T,Y,X", [(48, 8, 16), (40, 9, 27)])
def test_api_results_from_the_interleaved_layout_equal_separate_allocations(monkeypatch, T, Y, X):
    from atlite_amd import Cutout

    M = H.blob_matrix(5, Y, X, seed=6)

    def run():
        d, ds, _ = dataset(T, Y, X)
        c = Cutout(d)
        out = dict(pv_cells=c.pv(aggregate_time=None, **KW).values, pv_agg=c.pv(matrix=M, aggregate_time=None, **KW).values,
                   pv_map=c.pv(aggregate_time="mean", **KW).values,
                   wind_agg=c.wind(turbine="Vestas_V112_3MW", matrix=M, aggregate_time=None).values,
                   wind_cells=c.wind(turbine="Vestas_V112_3MW", aggregate_time=None).values,
                   heat=c.heat_demand(matrix=M, aggregate_time=None).values,
                   runoff=c.runoff(aggregate_time="sum").values)
        return out, c.data, ds

    monkeypatch.delenv("ATLITE_HIP_INTERLEAVE", raising=False)
    inter, d, ds = run()
    S, Sp = Y * X, (Y * X + 15) // 16 * 16
    ps = sorted(pools(d).values(), key=lambda p: -len(p.names))
    assert [len(p.names) for p in ps] == [7, 2] and ps[0].ld == 7 * Sp and ps[1].ld == 2 * Sp and ps[0].Sp == Sp
    v = d._device_cache["temperature"]
    k = ps[0].names.index("temperature")
    assert v.ptr == ps[0].base.ptr + k * Sp * 8 and v.ld == 7 * Sp and v.shape == (T, S)
    np.testing.assert_array_equal(v.numpy(), ds["temperature"])  # strided download of one cube of the pool
    np.testing.assert_array_equal(v.slab(5, 9).numpy(), ds["temperature"][5:9])
    monkeypatch.setenv("ATLITE_HIP_INTERLEAVE", "0")
    plain, d0, _ = run()
    assert not pools(d0)
    for key in inter:  # same kernels, same plan (the stride's alignment is the same), same arithmetic
        np.testing.assert_array_equal(inter[key], plain[key], err_msg=key)
    cells = orc.convert_pv(ds, H.CSI, ORI)
    np.testing.assert_allclose(inter["pv_agg"], orc.aggregate_matrix(cells, M), rtol=1e-10, atol=1e-12 * np.abs(cells).max())


def test_a_replaced_variable_returns_to_its_slot_and_a_new_set_regroups(monkeypatch):
    from atlite_amd import Cutout

    monkeypatch.delenv("ATLITE_HIP_INTERLEAVE", raising=False)
    T, Y, X = 36, 7, 19
    M = H.blob_matrix(3, Y, X, seed=2)
    d, ds, _ = dataset(T, Y, X, seed=11)
    c = Cutout(d)
    heat0 = c.heat_demand(matrix=M, aggregate_time=None).values  # temperature alone: an allocation of its own
    assert not pools(c.data) and "temperature" in c.data._device_cache
    alone = c.data._device_cache["temperature"]
    a = c.pv(matrix=M, aggregate_time=None, **KW).values  # regroups: six uploads + one device copy into a pool of seven
    (pool,) = pools(c.data).values()
    assert len(pool.names) == 7 and c.data._device_cache["temperature"] is not alone
    assert c.data._device_cache["temperature"]._pool is pool
    np.testing.assert_array_equal(c.heat_demand(matrix=M, aggregate_time=None).values, heat0)  # now from the pool's slot
    base = pool.base.ptr
    warmer = ds["temperature"].reshape(T, Y, X) + 7.5
    c.data["temperature"] = warmer
    assert "temperature" not in c.data._device_cache
    b = c.pv(matrix=M, aggregate_time=None, **KW).values
    (pool2,) = pools(c.data).values()
    assert pool2 is pool and pool.base.ptr == base  # refilled in place: no second copy of the other six
    ref = orc.aggregate_matrix(orc.convert_pv(dict(ds, temperature=warmer.reshape(T, -1)), H.CSI, ORI), M)
    np.testing.assert_allclose(b, ref, rtol=1e-10, atol=1e-12 * np.abs(ref).max())
    assert np.abs(a - b).max() > 0
    fresh, _, _ = dataset(T, Y, X, seed=11)
    fresh["temperature"] = warmer
    np.testing.assert_array_equal(Cutout(fresh).pv(matrix=M, aggregate_time=None, **KW).values, b)


def test_caller_device_arrays_keep_their_layout(monkeypatch):
    """A dataset whose cubes the caller already holds on the device is used where it lies (no pool, no copies)."""
    from atlite_amd import Cutout, Dataset
    from atlite_amd.device import default_context

    ctx = default_context()
    monkeypatch.delenv("ATLITE_HIP_INTERLEAVE", raising=False)
    T, Y, X = 24, 6, 16
    x, y = H.grid(Y, X)
    ds = H.pv_dataset(T, Y, X, seed=5)
    dev = {k: ctx.upload(v) for k, v in ds.items()}
    c = Cutout(Dataset(dict(dev), dict(time=pd.date_range("2013-06-01", periods=T, freq="h"), y=y, x=x)))
    M = H.blob_matrix(3, Y, X, seed=1)
    got = c.pv(matrix=M, aggregate_time=None, **KW).values
    assert not pools(c.data) and all(c.data._device_cache[k] is dev[k] or c.data._device_cache[k].ptr == dev[k].ptr for k in ds)
    ref = orc.aggregate_matrix(orc.convert_pv(ds, H.CSI, ORI), M)
    np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-12 * np.abs(ref).max())


def test_synthetic_generator_writes_the_interleaved_layout(ctx):
    from atlite_amd import synthetic

    T, Y, X = 30, 9, 21
    sep, _ = synthetic.pv_inputs(ctx, T, Y, X)
    il, _ = synthetic.pv_inputs(ctx, T, Y, X, interleaved=True)
    Sp = (Y * X + 15) // 16 * 16
    for k in sep:
        assert il[k].ld == 7 * Sp
        np.testing.assert_array_equal(il[k].numpy(), sep[k].numpy(), err_msg=k)
    M = H.blob_matrix(4, Y, X, seed=9)
    PV = dict(H.CSI, **ORI)
    for skip in (False, True):
        a = ctx.pv(sep, PV, T, Y * X, plan=ctx.plan(M, row_len=X), options=dict(night_skip=skip)).numpy()
        b = ctx.pv(il, PV, T, Y * X, plan=ctx.plan(M, row_len=X, ld=7 * Sp), options=dict(night_skip=skip)).numpy()
        np.testing.assert_allclose(b, a, rtol=1e-12, atol=1e-13 * np.abs(a).max())  # (another tile shape on the padded slots)


@pytest.mark.parametrize("fname", ["cutout_small_f32", "cutout_small_f64"])
def test_file_backed_variables_are_inflated_straight_into_their_slots(monkeypatch, fname):
    """A cutout file held resident (ATLITE_HIP_STREAM=0): the chunks of every variable a conversion reads are inflated,
    decoded and scattered into the slot-interleaved block (FileArray.to_device(out=view)); same bits as the in-memory twin."""
    import os

    from atlite_amd import Cutout, Dataset, io

    monkeypatch.delenv("ATLITE_HIP_INTERLEAVE", raising=False)
    monkeypatch.setenv("ATLITE_HIP_STREAM", "0")
    path = os.path.join(os.path.dirname(__file__), "golden", "nc", fname + ".nc")
    f = io.NcFile(path)
    ds = io.open_cutout(path)
    data = {n: f.read(n) for n in ds.keys()}
    cf = Cutout(ds)
    cm = Cutout(Dataset(data, {k: ds.coords[k] for k in ("time", "y", "x")}, chunked=True))
    Y, X = cf.shape
    M = H.blob_matrix(4, Y, X, seed=2)
    a = cf.pv(matrix=M, aggregate_time=None, **KW)
    b = cm.pv(matrix=M, aggregate_time=None, **KW)
    np.testing.assert_array_equal(a.values, b.values)
    (pf,), (pm,) = pools(cf.data).values(), pools(cm.data).values()
    assert len(pf.names) == len(pm.names) == 7 and pf.ld == pm.ld
    for n in pf.names:  # the file's bytes, widened to fp64, where the kernels read them
        np.testing.assert_array_equal(cf.data._device_cache[n].numpy().reshape(data[n].shape), data[n].astype(np.float64), err_msg=n)
    w = cf.wind(turbine="Vestas_V112_3MW", matrix=M, aggregate_time=None)
    w2 = cm.wind(turbine="Vestas_V112_3MW", matrix=M, aggregate_time=None)
    np.testing.assert_array_equal(w.values, w2.values)
    assert sorted(len(p.names) for p in pools(cf.data).values()) == [2, 7]
