"""
GPU: line-aligned plans (atl_agg_create_aligned) - contiguous (T, S) cubes whose slots do not start on 128-byte lines
(S % 16 != 0: what a caller's own C-ordered device arrays of a real ERA5 cutout look like, and what the reference's
stack(spatial=...) produces, convert.py:244).  The plan stacks 16 / gcd(S, 16) shifted copies of the matrix so that every tile
reads whole lines in every slot; the results must equal the ordinary plan's (same values, another order of the partial
sums: rtol 1e-12) and the oracle's (rtol 1e-10): pv with and without the early-out, wind, runoff, the temperature family, the
plain product, time reductions, NaN weights, fewer slots than alignment classes, several windows of partial rows, dense
(MFMA) tiles; conversions that cannot be re-addressed (heat demand: day groups; the in-kernel solar position: per-time
tables) and padded cubes are refused with an error the gateway answers with the ordinary plan.
"""
import numpy as np
import pandas as pd
import pytest
import scipy.sparse as sp

from atlite_amd import Cutout, Dataset
from oracle import atlite_oracle as orc
from tests import helpers as H

pytestmark = pytest.mark.gpu
PV = dict(H.CSI, slope=np.radians(30.0), azimuth=np.radians(180.0))
ORI = dict(slope=np.radians(30.0), azimuth=np.radians(180.0))
V = np.array([0, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 25, 25], dtype=float)
POW = np.array([0.0, 0.0, 0.005, 0.15, 0.3, 0.525, 0.905, 1.375, 1.95, 2.58, 2.96, 3.05, 3.06, 3.06, 0.0])


def close(a, b, rtol=1e-10, atol_scale=1e-12):
    a, b = np.asarray(a), np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol_scale * max(float(np.nanmax(np.abs(b))), 1e-300), equal_nan=True)


# (T, Y, X): S odd (16 classes) / S % 16 = 8 (2) / 4 (4) / 2 (8) / odd with fewer slots than classes / one row longer than a tile
GRIDS = [(37, 9, 27), (40, 12, 26), (21, 7, 36), (50, 21, 6), (5, 11, 13), (33, 1, 391)]


@pytest.mark.parametrize("T,Y,X", GRIDS)
def test_pv_on_aligned_plans(ctx, T, Y, X):
    S = Y * X
    ds = H.pv_dataset(T, Y, X, seed=T)
    dev = {k: ctx.upload(v) for k, v in ds.items()}  # contiguous
    M = H.blob_matrix(5, Y, X, seed=3)
    plan, aplan = ctx.plan(M, row_len=X), ctx.plan(M, row_len=X, aligned=True)
    assert aplan.aligned and aplan.info()["n_rows"] == 5 and aplan.info()["n_cells"] == S
    ref = orc.aggregate_matrix(orc.convert_pv(ds, H.CSI, ORI), M)
    for skip in (False, True):
        a = ctx.pv(dev, PV, T, S, plan=aplan, options=dict(night_skip=skip)).numpy()
        b = ctx.pv(dev, PV, T, S, plan=plan, options=dict(night_skip=skip)).numpy()
        close(a, ref)
        close(a, b, rtol=1e-12)
        for agg in ("sum", "mean"):
            close(ctx.pv(dev, PV, T, S, plan=aplan, time_agg=agg, options=dict(night_skip=skip)).numpy(),
                  getattr(ref, agg)(axis=1))
    # another member of the fast family (Hay-Davies + bofinger panel) and a tracker
    for opt in (dict(trigon_model="other"), dict(tracking="horizontal")):
        close(ctx.pv(dev, PV, T, S, plan=aplan, options=opt).numpy(), ctx.pv(dev, PV, T, S, plan=plan, options=opt).numpy(), rtol=1e-12)


@pytest.mark.parametrize("T,Y,X", GRIDS[:4])
def test_wind_runoff_temperature_spmm_on_aligned_plans(ctx, T, Y, X):
    S = Y * X
    w = H.wind_dataset(T, Y, X, seed=2)
    M = H.blob_matrix(3, Y, X, seed=5)
    aplan = ctx.plan(M, row_len=X if T % 2 else None, aligned=True)  # 2-d tiles / flat strips
    dw, dz = ctx.upload(w["wnd100m"]), ctx.upload(w["roughness"])
    ref = orc.convert_wind(w["wnd100m"], w["roughness"], V, POW, 3.06, 80.0, 100.0)
    close(ctx.wind(dw, dz, V, POW / 3.06, 80.0, 100.0, "logarithmic", T, S, plan=aplan).numpy(), orc.aggregate_matrix(ref, M))
    # a static roughness field: the per-cell setup reads it at the REAL cell index
    z0 = ctx.upload(np.ascontiguousarray(w["roughness"][0]))
    ref0 = orc.convert_wind(w["wnd100m"], np.broadcast_to(w["roughness"][0], w["roughness"].shape), V, POW, 3.06, 80.0, 100.0)
    close(ctx.wind(dw, z0, V, POW / 3.06, 80.0, 100.0, "logarithmic", T, S, plan=aplan).numpy(), orc.aggregate_matrix(ref0, M))
    rng = np.random.default_rng(T)
    ro, h = rng.random((T, S)), rng.random(S) * 900.0
    ro[rng.random((T, S)) < 0.02] = np.nan
    dro, dh = ctx.upload(ro), ctx.upload(h)
    close(ctx.runoff(dro, dh, T, S, plan=aplan).numpy(), orc.aggregate_matrix(ro * h[None, :], M))
    close(ctx.spmm(aplan, dro).numpy(), orc.aggregate_matrix(ro, M))
    close(ctx.spmm(aplan, dro, time_agg="mean").numpy(), np.nanmean(orc.aggregate_matrix(ro, M), axis=1))
    t = 250.0 + 50.0 * rng.random((T, S))
    close(ctx.thermo(ctx.upload(t), T, S, plan=aplan).numpy(), orc.aggregate_matrix(t - 273.15, M))


def test_nan_weights_windows_and_dense_tiles(ctx, monkeypatch):
    T, Y, X = 70, 13, 21  # S = 273: 16 classes
    S = Y * X
    rng = np.random.default_rng(4)
    ro = rng.random((T, S))
    dro = ctx.upload(ro)
    # 20 rows that cover every cell (dense tiles: the MFMA instantiation), one of them with a NaN weight
    D = 0.5 + rng.random((20, S))
    D[7, 100] = np.nan
    M = sp.csr_matrix(D)
    ref = orc.aggregate_matrix(ro, M)
    assert np.isnan(ref[7]).all()
    aplan = ctx.plan(M, row_len=X, aligned=True)
    close(ctx.spmm(aplan, dro).numpy(), ref)
    close(ctx.spmm(ctx.plan(M, aligned=True), dro).numpy(), ref)
    # partial rows in windows of 64 virtual slots... here the whole range is 5 virtual slots; force tiny windows anyway
    monkeypatch.setenv("ATLITE_HIP_PARTIAL_BUDGET", "1")
    T2 = 16 * 70 + 3
    ro2 = rng.random((T2, S))
    close(ctx.spmm(aplan, ctx.upload(ro2)).numpy(), orc.aggregate_matrix(ro2, M))
    monkeypatch.delenv("ATLITE_HIP_PARTIAL_BUDGET")
    close(ctx.spmm(aplan, ctx.upload(ro2), time_agg="sum").numpy(), np.nansum(orc.aggregate_matrix(ro2, M), axis=1))  # NaN-skipping, like the gateway


def test_refusals(ctx):
    T, Y, X = 48, 9, 27
    S = Y * X
    M = H.blob_matrix(3, Y, X, seed=1)
    aplan = ctx.plan(M, aligned=True)
    t = ctx.upload(280.0 + np.zeros((T, S)))
    with pytest.raises(NotImplementedError, match="line-aligned plan"):  # day groups index the cube by the hour (ATL_E_UNSUPPORTED: the gateway falls back)
        ctx.heat_demand(t, np.arange(0, T + 1, 24), 288.15, 1.0, 0.0, T, S, plan=aplan)
    padded = ctx.upload(np.zeros((T, S)), ld=S + 13)
    # (padded slots: ATL_E_UNSUPPORTED as well since round 6 - the gateway takes the ordinary plan instead of raising to the user)
    with pytest.raises(NotImplementedError, match="line-aligned plan"):
        ctx.spmm(aplan, padded)
    with pytest.raises(ValueError, match="start on 128-byte lines already"):
        ctx.plan(sp.csr_matrix(np.ones((2, 64))), aligned=True)
    with pytest.raises(ValueError, match="has 243 columns"):
        ctx.spmm(aplan, ctx.upload(np.zeros((T, S + 1))))


def test_gateway_uses_the_aligned_plan_for_caller_owned_cubes(ctx, monkeypatch):
    """Cutout.pv / wind / runoff / heat_demand on a dataset whose variables are the caller's own contiguous device arrays
    (odd grid): pv, wind and runoff go through the line-aligned plan, heat demand takes the ordinary one; a conversion the
    library refuses on such a plan (the in-kernel solar position) falls back by itself."""
    from atlite_amd import convert as cv
    from atlite_amd.device import default_context

    T, Y, X = 48, 9, 27
    dctx = default_context()
    host = H.pv_dataset(T, Y, X, seed=9)
    w = H.wind_dataset(T, Y, X, seed=9)
    rng = np.random.default_rng(9)
    host.update(wnd100m=w["wnd100m"], roughness=w["roughness"], runoff=rng.random((T, Y * X)), height=rng.random(Y * X) * 500.0)
    data = {k: (dctx.upload(np.ascontiguousarray(v)).reshape(T, Y, X) if v.ndim == 2 and v.shape[0] == T else v.reshape(Y, X))
            for k, v in host.items()}
    t = pd.date_range("2013-01-01", periods=T, freq="h")
    ds = Dataset(data, dict(time=t, y=30.0 + np.arange(Y), x=-5.0 + np.arange(X)))
    assert ds._caller_layout()
    c = Cutout(ds)
    M = H.blob_matrix(4, Y, X, seed=2)
    used = []
    real = cv._execute
    monkeypatch.setattr(cv, "_execute", lambda ctx_, spec, ds_, plan, ta: (used.append((type(spec).__name__, bool(getattr(plan, "aligned", False)))),
                                                                              real(ctx_, spec, ds_, plan, ta))[1])
    pv = c.pv(panel="CSi", orientation=dict(slope=30.0, azimuth=180.0), matrix=M, aggregate_time=None)
    close(np.asarray(pv.values), orc.aggregate_matrix(orc.convert_pv(host, H.CSI, ORI), M))
    wd = c.wind(turbine="Vestas_V112_3MW", matrix=M, aggregate_time=None)
    close(np.asarray(wd.values), orc.aggregate_matrix(orc.convert_wind(w["wnd100m"], w["roughness"], V, POW, 3.06, 80.0, 100.0), M))
    ro = c.runoff(matrix=M, aggregate_time=None)
    close(np.asarray(ro.values), orc.aggregate_matrix(host["runoff"] * host["height"][None, :], M))
    hd = c.heat_demand(matrix=M, aggregate_time=None)
    assert used[0] == ("_PvSpec", True) and used[1] == ("_WindSpec", True) and used[2] == ("_RunoffSpec", True)
    assert used[3] == ("_HeatSpec", False) and len(used) == 4 and hd.values.shape[0] == 4  # (day groups: never tried)
    # no stored solar angles: the in-kernel solar position reads per-time tables - no line-aligned plan for it, the gateway
    # runs the ordinary plan; same values as the same call with the plans switched off
    import warnings

    ds2 = Dataset({k: v for k, v in data.items() if not k.startswith("solar_")}, dict(time=t, y=30.0 + np.arange(Y), x=-5.0 + np.arange(X)))
    used.clear()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sp1 = Cutout(ds2).pv(panel="CSi", orientation=dict(slope=30.0, azimuth=180.0), matrix=M, aggregate_time=None)
        assert used == [("_PvSpec", False)]  # (round 5: the spec says so itself - aligned_ok - and the stacked plan is never built)
        monkeypatch.setenv("ATLITE_HIP_ALIGNED_PLANS", "0")
        sp0 = Cutout(ds2).pv(panel="CSi", orientation=dict(slope=30.0, azimuth=180.0), matrix=M, aggregate_time=None)
    np.testing.assert_array_equal(np.asarray(sp1.values), np.asarray(sp0.values))
    assert np.isfinite(np.asarray(sp1.values)).all() and np.asarray(sp1.values).max() > 0
    used.clear()
    pv0 = c.pv(panel="CSi", orientation=dict(slope=30.0, azimuth=180.0), matrix=M, aggregate_time=None)
    assert used == [("_PvSpec", False)]
    close(np.asarray(pv0.values), np.asarray(pv.values), rtol=1e-12)


def test_full_year_on_a_real_world_grid(ctx):
    """8760 x 201 x 201 (S % 16 = 1, the bench's odd_caller leg), 100 blob shapes, the seven cubes contiguous: the line-aligned
    plan against the ordinary plan over the whole year, both pv kernels, and a size-independent property - the time sum of
    the aggregated series equals the weights' product with the per-cell time sum (another kernel, another order)."""
    from atlite_amd import synthetic

    T, Y, X = 8760, 201, 201
    S = Y * X
    cubes, _ = synthetic.pv_inputs(ctx, T, Y, X)
    M = H.blob_matrix(100, Y, X, seed=7)
    plain, aligned = ctx.plan(M, row_len=X), ctx.plan(M, row_len=X, aligned=True)
    assert aligned.info()["n_segments"] == 16 * (aligned.info()["n_segments"] // 16)
    for skip in (False, True):
        a = ctx.pv(cubes, PV, T, S, plan=aligned, options=dict(night_skip=skip)).numpy()
        b = ctx.pv(cubes, PV, T, S, plan=plain, options=dict(night_skip=skip)).numpy()
        close(a, b, rtol=1e-12)
        assert a.max() > 0 and (a == 0).any()
    per_cell_sum = ctx.pv(cubes, PV, T, S, time_agg="sum", options=dict(night_skip=True, row_len=X)).numpy()
    close(a.sum(axis=1), np.asarray(M @ per_cell_sum).ravel(), rtol=1e-10)
    close(ctx.pv(cubes, PV, T, S, plan=aligned, time_agg="sum", options=dict(night_skip=True)).numpy(), a.sum(axis=1), rtol=1e-11)
