"""
GPU: grids whose cell count or row length is odd - every real-world ERA5 cutout with integer-degree bounds (x and y
both hold 4 (b - a) + 1 points) - run the VECTORISED kernels since round 3: the lane's two cells as one 16-byte access
that is only 8-byte aligned in every other slot, the lane that owns the last cell storing a single value, cell pairs
that straddle two grid rows owned by flat index.  Checked against the oracle, against the unvectorised instantiations
($ATLITE_HIP_NO_VEC; same arithmetic, so the same bits), and on a cube that ends on a 4 KiB page (where the launch must
not read past it: the unvectorised kernel takes it).  Reference: the conversions of atlite/convert.py on any grid.
"""
import numpy as np
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
