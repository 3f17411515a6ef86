"""
GPU parity: the HIP path, called through the C ABI (ctypes), against the CPU oracle on the
same seeded inputs.  Tolerance (BASELINE.json north_star): rtol 1e-10 (+ atol = 1e-12*max to
absorb denormal-scale differences at sunrise/sunset; heat demand: atol 1e-9*a).
"""
import numpy as np
import pytest

from oracle import atlite_oracle as orc
from tests import helpers as H

pytestmark = pytest.mark.gpu

RTOL = 1e-10


def close(a, b, atol_scale=1e-12):
    a, b = np.asarray(a), np.asarray(b)
    atol = atol_scale * max(float(np.nanmax(np.abs(b))) if b.size else 0.0, 1e-300)
    np.testing.assert_allclose(a, b, rtol=RTOL, atol=atol, equal_nan=True)


def up(ctx, ds):
    return {k: ctx.upload(v) for k, v in ds.items()}


PV_PARAMS = dict(H.CSI, slope=np.radians(30.0), azimuth=np.radians(180.0))


@pytest.mark.parametrize("T,Y,X", [(48, 7, 9), (30, 8, 16), (24, 1, 1), (9, 3, 129)])
def test_pv_cells_series(ctx, T, Y, X):
    ds = H.pv_dataset(T, Y, X, seed=1)
    ref = orc.convert_pv(ds, H.CSI, dict(slope=np.radians(30.0), azimuth=np.radians(180.0)))
    out = ctx.pv(up(ctx, ds), PV_PARAMS, T, Y * X).numpy()
    assert ref.max() > 0.1 and (ref == 0).any()
    close(out, ref)


@pytest.mark.parametrize("T,Y,X,N", [(48, 12, 20, 5), (100, 16, 16, 7), (17, 5, 27, 3), (8, 2, 64, 1)])
def test_pv_fused_aggregate(ctx, T, Y, X, N):
    ds = H.pv_dataset(T, Y, X, seed=2)
    M = H.blob_matrix(N, Y, X, seed=3)
    ref_cells = orc.convert_pv(ds, H.CSI, dict(slope=np.radians(30.0), azimuth=np.radians(180.0)))
    ref = orc.aggregate_matrix(ref_cells, M)
    plan = ctx.plan(M)
    out = ctx.pv(up(ctx, ds), PV_PARAMS, T, Y * X, plan=plan).numpy()
    assert out.shape == (N, T)
    close(out, ref)
    for agg in ("sum", "mean"):
        o = ctx.pv(up(ctx, ds), PV_PARAMS, T, Y * X, plan=plan, time_agg=agg).numpy()
        close(o, orc.aggregate_time(ref, agg, axis=1))


def test_pv_per_cell_orientation(ctx):
    T, Y, X = 36, 6, 10
    ds = H.pv_dataset(T, Y, X, seed=4)
    _, y = H.grid(Y, X)
    ori = orc.orientation_latitude_optimal(np.radians(y))
    slope = np.repeat(ori["slope"], X)
    az = np.repeat(ori["azimuth"], X)
    ref = orc.convert_pv(ds, H.CSI, dict(slope=slope[None, :], azimuth=az[None, :]))
    out = ctx.pv(up(ctx, ds), dict(H.CSI, slope=slope, azimuth=az), T, Y * X).numpy()
    close(out, ref)


def test_pv_time_reduced_cells(ctx):
    T, Y, X = 50, 6, 11
    ds = H.pv_dataset(T, Y, X, seed=5)
    ref = orc.convert_pv(ds, H.CSI, dict(slope=np.radians(30.0), azimuth=np.radians(180.0)))
    for agg in ("sum", "mean"):
        out = ctx.pv(up(ctx, ds), PV_PARAMS, T, Y * X, time_agg=agg).numpy()
        close(out, orc.aggregate_time(ref, agg, axis=0))


@pytest.mark.parametrize("method,aux", [("logarithmic", "roughness"), ("power", "wnd_shear_exp"), (None, None)])
def test_wind_cells_and_fused(ctx, method, aux):
    T, Y, X, N = 40, 9, 14, 4
    ds = H.wind_dataset(T, Y, X, seed=6)
    # exercise knots, cut-out, NaN, inf, below/above range
    ds["wnd100m"][0, :8] = [0.0, 2.0, 25.0, 24.999999, 30.0, np.nan, np.inf, 13.0]
    tb = H.V112
    a = ds[aux] if aux else None
    ref = orc.convert_wind(ds["wnd100m"], a, tb["V"], tb["POW"], tb["P"], 80.0, 100.0, method)
    d_w = ctx.upload(ds["wnd100m"])
    d_a = ctx.upload(a) if aux else None
    out = ctx.wind(d_w, d_a, tb["V"], tb["POW"] / tb["P"], 80.0, 100.0, method, T, Y * X).numpy()
    close(out, ref)
    M = H.blob_matrix(N, Y, X, seed=7)
    plan = ctx.plan(M)
    ds["wnd100m"][0, :8] = 5.0  # NaN/inf would poison whole rows; covered separately
    ref = orc.convert_wind(ds["wnd100m"], a, tb["V"], tb["POW"], tb["P"], 80.0, 100.0, method)
    d_w = ctx.upload(ds["wnd100m"])
    out = ctx.wind(d_w, d_a, tb["V"], tb["POW"] / tb["P"], 80.0, 100.0, method, T, Y * X, plan=plan).numpy()
    close(out, orc.aggregate_matrix(ref, M))


def test_wind_knot_exact(ctx):
    tb = H.V112
    x = np.array([[-1, 0, 1.9, 2, 2.5, 12.99, 13, 24.999999, 25, 25.0000001, 30, np.nan, np.inf, 7.0]])
    ref = np.interp(x, tb["V"], tb["POW"] / tb["P"])
    out = ctx.wind(ctx.upload(x), None, tb["V"], tb["POW"] / tb["P"], 80.0, 80.0, None, 1, x.shape[1]).numpy()
    np.testing.assert_array_equal(out, ref)


def test_heat_demand(ctx):
    T, Y, X, N = 24 * 5 + 7, 6, 9, 3
    rng = np.random.default_rng(8)
    temp = 283.15 + 8 * rng.standard_normal((T, Y * X))
    temp[5, 3] = np.nan
    t = H.times(T)
    for shift in (0.0, 4.0, -5.0):
        ptr, _ = orc.day_groups(t, shift)
        ref = orc.convert_heat_demand(temp, ptr, threshold=15.0, a=1.3, constant=0.2)
        d_t = ctx.upload(temp)
        out = ctx.heat_demand(d_t, ptr, 15.0 + 273.15, 1.3, 0.2, T, Y * X).numpy()
        close(out, ref, atol_scale=1e-9)
        tfin = np.where(np.isnan(temp), 280.0, temp)
        ref = orc.convert_heat_demand(tfin, ptr, threshold=15.0, a=1.3, constant=0.2)
        M = H.blob_matrix(N, Y, X, seed=9)
        out = ctx.heat_demand(ctx.upload(tfin), ptr, 15.0 + 273.15, 1.3, 0.2, T, Y * X, plan=ctx.plan(M)).numpy()
        close(out, orc.aggregate_matrix(ref, M), atol_scale=1e-9)


def test_runoff(ctx):
    T, Y, X, N = 33, 7, 13, 4
    rng = np.random.default_rng(10)
    ro = -1e-4 * np.log1p(-rng.random((T, Y * X)))
    h = 2000 * rng.random(Y * X)
    M = H.blob_matrix(N, Y, X, seed=11)
    plan = ctx.plan(M)
    for height in (h, None):
        ref = orc.convert_runoff(ro, height[None, :] if height is not None else None)
        d_h = ctx.upload(height) if height is not None else None
        close(ctx.runoff(ctx.upload(ro), d_h, T, Y * X).numpy(), ref)
        close(ctx.runoff(ctx.upload(ro), d_h, T, Y * X, plan=plan).numpy(), orc.aggregate_matrix(ref, M))


@pytest.mark.parametrize("S", [1, 2, 127, 128, 129, 255, 1000])
def test_spmm_shapes(ctx, S):
    """generic CSR product incl. odd S (scalar path), duplicates, empty rows, explicit zeros."""
    import scipy.sparse as sp

    T, N = 19, 6
    rng = np.random.default_rng(S)
    D = rng.standard_normal((T, S))
    M = sp.random(N, S, density=min(1.0, 8.0 / S + 0.05), random_state=S, format="csr")
    M = sp.csr_matrix(M)
    M[2, :] = 0  # becomes structurally empty after eliminate_zeros
    M.eliminate_zeros()
    ref = M @ D.T
    out = ctx.spmm(ctx.plan(M), ctx.upload(D)).numpy()
    close(out, ref, atol_scale=1e-13)


def test_spmm_nan_semantics(ctx):
    """CSR skips structural zeros: NaN cells outside a shape must not leak into it
    (atlite/convert.py:313-316 relies on this); explicit zero weights do propagate NaN."""
    import scipy.sparse as sp

    T, S = 8, 300
    rng = np.random.default_rng(0)
    D = rng.random((T, S))
    D[:, 10] = np.nan
    D[3, 200] = np.nan
    rows = [0, 0, 1, 1, 2]
    cols = [5, 6, 10, 11, 200]
    vals = [1.0, 0.5, 0.0, 1.0, 2.0]  # row 1 holds an EXPLICIT zero on the NaN cell
    M = sp.csr_matrix((vals, (rows, cols)), shape=(3, S))
    ref = M @ D.T
    out = ctx.spmm(ctx.plan(M), ctx.upload(D)).numpy()
    np.testing.assert_allclose(out, ref, rtol=1e-13, equal_nan=True)
    assert np.isfinite(out[0]).all() and np.isnan(out[1]).all() and np.isnan(out[2, 3])


@pytest.mark.parametrize("tile", ["flat", "128x1", "64x2", "32x4", "16x8", None])
@pytest.mark.parametrize("Y,X", [(7, 9), (8, 16), (13, 50), (3, 130), (40, 33)])
def test_tile_layouts(ctx, monkeypatch, tile, Y, X):
    """Every cell-tile shape (and the automatic choice) gives the same aggregation; odd X takes
    the scalar-load path, edge tiles are partially filled."""
    if tile is None:
        monkeypatch.delenv("ATLITE_HIP_TILE", raising=False)
    else:
        monkeypatch.setenv("ATLITE_HIP_TILE", tile)
    T, N = 21, 6
    rng = np.random.default_rng(Y * 1000 + X)
    D = rng.standard_normal((T, Y * X))
    M = H.blob_matrix(N, Y, X, seed=X)
    plan = ctx.plan(M, row_len=X)
    info = plan.info()
    if tile not in (None, "flat"):
        assert f"{info['tile_w']}x{info['tile_h']}" == tile
    close(ctx.spmm(plan, ctx.upload(D)).numpy(), M @ D.T, atol_scale=1e-13)
    ds = H.pv_dataset(T, Y, X, seed=3)
    ref = orc.aggregate_matrix(orc.convert_pv(ds, H.CSI, dict(slope=np.radians(30.0), azimuth=np.radians(180.0))), M)
    close(ctx.pv(up(ctx, ds), PV_PARAMS, T, Y * X, plan=plan).numpy(), ref)


# ---- edge cases: empty and ragged inputs, degenerate matrices ------------------------------------
def test_empty_and_degenerate_shapes(ctx):
    import scipy.sparse as sp

    rng = np.random.default_rng(0)
    # no shapes at all / a shape with no cells / an all-zero matrix
    D = rng.random((5, 40))
    assert ctx.spmm(ctx.plan(sp.csr_matrix((0, 40))), ctx.upload(D)).numpy().shape == (0, 5)
    M = sp.csr_matrix(([1.0], ([1], [3])), shape=(3, 40))
    out = ctx.spmm(ctx.plan(M, row_len=8), ctx.upload(D)).numpy()
    np.testing.assert_array_equal(out[0], 0.0)
    np.testing.assert_array_equal(out[2], 0.0)
    np.testing.assert_allclose(out[1], D[:, 3])
    # empty time axis
    E = ctx.spmm(ctx.plan(M), ctx.upload(np.zeros((0, 40)))).numpy()
    assert E.shape == (3, 0)
    assert ctx.spmm(ctx.plan(M), ctx.upload(np.zeros((0, 40))), time_agg="sum").numpy().tolist() == [0.0, 0.0, 0.0]
    assert np.isnan(ctx.spmm(ctx.plan(M), ctx.upload(np.zeros((0, 40))), time_agg="mean").numpy()).all()
    # ... also for the per-cell pv kernels, with and without the night early-out
    ds0 = {k: ctx.upload(np.zeros((0, 40))) for k in H.pv_dataset(1, 5, 8)}
    for skip in (False, True):
        opt = dict(night_skip=skip, row_len=8)
        assert ctx.pv(ds0, PV_PARAMS, 0, 40, options=opt).numpy().shape == (0, 40)
        assert (ctx.pv(ds0, PV_PARAMS, 0, 40, time_agg="sum", options=opt).numpy() == 0.0).all()
        assert np.isnan(ctx.pv(ds0, PV_PARAMS, 0, 40, time_agg="mean", options=opt).numpy()).all()
        pl = ctx.plan(M, row_len=8)
        assert ctx.pv(ds0, PV_PARAMS, 0, 40, plan=pl, options=opt).numpy().shape == (3, 0)
        assert ctx.pv(ds0, PV_PARAMS, 0, 40, plan=pl, time_agg="sum", options=opt).numpy().tolist() == [0.0, 0.0, 0.0]
    # single cell, single step; chunk tails (T not a multiple of 8 / 64)
    for T, S in ((1, 1), (1, 2), (7, 3), (65, 130), (129, 64)):
        D = rng.standard_normal((T, S))
        M = sp.random(4, S, density=0.6, random_state=T, format="csr")
        close(ctx.spmm(ctx.plan(M), ctx.upload(D)).numpy(), M @ D.T, atol_scale=1e-13)
        close(ctx.runoff(ctx.upload(D), None, T, S, time_agg="sum").numpy(), D.sum(0), atol_scale=1e-13)
        close(ctx.runoff(ctx.upload(D), None, T, S).numpy(), D)
    # duplicate entries are summed like scipy does on conversion
    M = sp.csr_matrix((np.array([1.0, 2.0, 0.5]), np.array([2, 2, 5]), np.array([0, 3])), shape=(1, 9))
    D = rng.random((3, 9))
    close(ctx.spmm(ctx.plan(M), ctx.upload(D)).numpy(), (3.0 * D[:, 2] + 0.5 * D[:, 5])[None, :], atol_scale=1e-14)
    # a NaN weight poisons its whole row (what the scipy product gives), other rows stay clean
    M = sp.csr_matrix((np.array([np.nan, 1.0]), (np.array([0, 1]), np.array([1, 2]))), shape=(2, 9))
    out = ctx.spmm(ctx.plan(M), ctx.upload(D)).numpy()
    assert np.isnan(out[0]).all() and np.isfinite(out[1]).all()


def test_bad_arguments_raise(ctx):
    import scipy.sparse as sp

    M = sp.csr_matrix(np.ones((2, 10)))
    with pytest.raises(ValueError, match="columns"):
        ctx.spmm(ctx.plan(M), ctx.upload(np.zeros((3, 12))))
    with pytest.raises(ValueError, match="increasing"):
        ctx.wind(ctx.upload(np.ones((2, 4))), None, [0, 5, 3], [0, 1, 1], 80, 80, None, 2, 4)
    with pytest.raises(ValueError):
        ctx.plan(sp.csr_matrix((np.ones(1), np.array([11]), np.array([0, 1])), shape=(1, 12)).__class__(
            (np.ones(1), np.array([5]), np.array([0, 1])), shape=(1, 4)))


def test_many_shapes_and_large_tile_rows(ctx):
    """More shapes per tile than the register-cached rows (generic row loop) and a dense matrix."""
    import scipy.sparse as sp

    T, Y, X, N = 19, 8, 32, 40
    rng = np.random.default_rng(5)
    D = rng.standard_normal((T, Y * X))
    M = sp.csr_matrix(rng.random((N, Y * X)) * (rng.random((N, Y * X)) < 0.7))
    close(ctx.spmm(ctx.plan(M, row_len=X), ctx.upload(D)).numpy(), M @ D.T, atol_scale=1e-13)
    D[3, 17] = np.nan
    D[5, 100] = np.inf
    ref = M @ D.T
    got = ctx.spmm(ctx.plan(M, row_len=X), ctx.upload(D)).numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-12, equal_nan=True)


def test_partial_row_windowing(ctx, monkeypatch):
    """Dense matrices: the slot axis is processed in windows so that scratch stays bounded."""
    import scipy.sparse as sp

    T, Y, X, N = 300, 6, 40, 9
    rng = np.random.default_rng(11)
    D = rng.standard_normal((T, Y * X))
    M = sp.csr_matrix(rng.random((N, Y * X)))
    ref = M @ D.T
    monkeypatch.setenv("ATLITE_HIP_PARTIAL_BUDGET", "2000")  # -> windows of 64 slots
    plan = ctx.plan(M, row_len=X)
    close(ctx.spmm(plan, ctx.upload(D)).numpy(), ref, atol_scale=1e-13)
    close(ctx.spmm(plan, ctx.upload(D), time_agg="mean").numpy(), ref.mean(1), atol_scale=1e-13)
    ds = H.pv_dataset(T, Y, X, seed=1)
    refpv = orc.aggregate_matrix(orc.convert_pv(ds, H.CSI, dict(slope=np.radians(30.0), azimuth=np.radians(180.0))), M)
    close(ctx.pv(up(ctx, ds), PV_PARAMS, T, Y * X, plan=plan).numpy(), refpv)


def test_rccl_comm_single_rank(ctx):
    """C-ABI RCCL communicator: with one rank the all-gather is a placement copy and the all-reduce
    the identity (the multi-rank flow is covered by tests/test_distributed_gloo.py and by
    `bench.py --debug-gloo-one-gpu`)."""
    from atlite_amd.distributed import RcclComm

    comm = RcclComm(ctx, 1, 0, RcclComm.unique_id())
    a = np.random.default_rng(0).random((5, 37))
    d = ctx.upload(a)
    np.testing.assert_array_equal(comm.gather_time(d).numpy(), a)
    np.testing.assert_array_equal(comm.allreduce_sum(d).numpy(), a)
    comm.close()


@pytest.mark.parametrize("Y,X", [(12, 20), (5, 27)])
def test_pv_night_skip_is_bit_identical(ctx, Y, X):
    """night_skip only avoids reading streams whose values cannot matter: identical bits, for the
    fused, per-cell and time-reduced kernels, scalar and per-cell orientation, NaN altitudes."""
    T, N = 72, 5
    ds = H.pv_dataset(T, Y, X, seed=9)
    ds["solar_altitude"][40, 3] = np.nan          # a NaN altitude is not "night"
    ds["temperature"][2, :] = np.nan              # night rows: NaN inputs must not leak either way
    ds["influx_direct"][3, :] = np.inf
    M = H.blob_matrix(N, Y, X, seed=10)
    plan = ctx.plan(M, row_len=X)
    dev = up(ctx, ds)
    _, y = H.grid(Y, X)
    lo = orc.orientation_latitude_optimal(np.radians(y))
    for params in (PV_PARAMS, dict(H.CSI, slope=np.repeat(lo["slope"], X), azimuth=np.repeat(lo["azimuth"], X))):
        for kw in (dict(plan=plan), dict(), dict(time_agg="sum"), dict(plan=plan, time_agg="mean")):
            a = ctx.pv(dev, params, T, Y * X, options=dict(night_skip=False), **kw).numpy()
            b = ctx.pv(dev, params, T, Y * X, options=dict(night_skip=True), **kw).numpy()
            np.testing.assert_array_equal(a, b)
    ref = orc.aggregate_matrix(orc.convert_pv(ds, H.CSI, dict(slope=np.radians(30.0), azimuth=np.radians(180.0))), M)
    got = ctx.pv(dev, PV_PARAMS, T, Y * X, plan=plan, options=dict(night_skip=True)).numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-12 * np.nanmax(np.abs(ref[np.isfinite(ref)])), equal_nan=True)


@pytest.mark.parametrize("Y,X", [(3, 5), (9, 16), (17, 33), (8, 200), (40, 7), (1, 130), (25, 2)])
def test_pv_night_skip_per_cell_kernels_on_grid_tiles(ctx, Y, X):
    """With the grid's row length (atl_pv_inputs.X / options row_len, what Cutout.pv() passes) the per-cell early-out
    kernels walk the fused kernels' 16 x 8 tiles instead of 128-cell strips: every cell still owned exactly once -
    the same bits as the kernels without the early-out, for rows shorter than a tile, odd row lengths (unvectorised),
    single rows and columns, stored angles and the in-kernel solar position."""
    from atlite_amd import solar

    T = 50
    ds = H.pv_dataset(T, Y, X, seed=5)
    ds["temperature"][2, :] = np.nan
    dev = up(ctx, ds)
    x, y = H.grid(Y, X)
    h, dec = solar.hour_angle(H.times(T), x, "-30min")
    lat = np.radians(y)
    tables = dict(sin_dec=np.sin(dec), cos_dec=np.cos(dec), h=h, cos_h=np.cos(h), sin_lat=np.sin(lat), cos_lat=np.cos(lat))
    five = {k: v for k, v in dev.items() if not k.startswith("solar_")}
    lo = orc.orientation_latitude_optimal(lat)
    for params in (PV_PARAMS, dict(H.CSI, slope=np.repeat(lo["slope"], X), azimuth=np.repeat(lo["azimuth"], X))):
        for kw in (dict(), dict(time_agg="sum"), dict(time_agg="mean")):
            a = ctx.pv(dev, params, T, Y * X, options=dict(night_skip=False), **kw).numpy()
            b = ctx.pv(dev, params, T, Y * X, options=dict(night_skip=True, row_len=X), **kw).numpy()
            np.testing.assert_array_equal(a, b)
            a = ctx.pv(five, params, T, Y * X, solar_tables=tables, options=dict(night_skip=False), **kw).numpy()
            b = ctx.pv(five, params, T, Y * X, solar_tables=tables, options=dict(night_skip=True), **kw).numpy()
            np.testing.assert_array_equal(a, b)
            if not kw:
                assert (a == 0).any() and a.max() > 0


@pytest.mark.parametrize("opts", [
    dict(tracking="horizontal"), dict(tracking="tilted_horizontal", trigon_model="other"), dict(tracking="vertical"),
    dict(tracking="dual", trigon_model="other"), dict(trigon_model="other"), dict(panel_model="none"),
    dict(panel_model="none", trigon_model="other", irradiation="diffuse"),
    dict(panel_model="solar_thermal", c0=0.8, c1=3.0, t_store_K=353.15),
    dict(panel_model="solar_thermal", c0=0.8, c1=3.0, t_store_K=353.15, trigon_model="other"),
    dict(panel="KANENA"), dict(panel="KANENA", trigon_model="other"),
    # trackers x tails that are fused fast-family kernels since round 6 (atl_kernels_pvkt.hip; the general kernel before)
    dict(tracking="horizontal", trigon_model="other"), dict(tracking="vertical", trigon_model="other"),
    dict(panel="KANENA", tracking="horizontal"), dict(panel="KANENA", tracking="dual"),
    dict(panel_model="none", tracking="tilted_horizontal"), dict(panel_model="none", tracking="vertical"),
    dict(panel_model="none", tracking="dual", irradiation="ground"),
    # ... and the rest behind a run-time tracker switch (atl_kernels_pvka.hip: per-cell orientations, Hay-Davies before the
    # irradiation / bofinger tails)
    dict(panel="KANENA", tracking="tilted_horizontal", trigon_model="other"),
    dict(panel_model="none", tracking="horizontal", trigon_model="other", irradiation="direct"),
    dict(panel_model="none", tracking="vertical", trigon_model="other")])
def test_pv_night_skip_other_tails_and_trackers(ctx, opts):
    """The night early-out for the family's other members - trackers with the Huld panel, and the bofinger /
    solar thermal / irradiation tails after either trigon model with a fixed panel: the same bits as without it (fused,
    per-cell series, time reduction; one orientation or one per cell; NaN / inf in night rows), oracle values."""
    from atlite_amd.resource import get_solarpanelconfig

    T, Y, X, N = 61, 9, 20, 6
    opts = dict(opts)
    panel = get_solarpanelconfig(opts.pop("panel")) if "panel" in opts else H.CSI
    ds = H.pv_dataset(T, Y, X, seed=41)
    ds["solar_altitude"][30, 5] = np.nan
    ds["temperature"][2, :] = np.nan
    ds["influx_direct"][3, :] = np.inf
    M = H.blob_matrix(N, Y, X, seed=42)
    plan = ctx.plan(M, row_len=X)
    dev = up(ctx, ds)
    _, y = H.grid(Y, X)
    lo = orc.orientation_latitude_optimal(np.radians(y))
    scal = dict(panel, slope=np.radians(30.0), azimuth=np.radians(180.0))
    for params in (scal, dict(panel, slope=np.repeat(lo["slope"], X), azimuth=np.repeat(lo["azimuth"], X))):
        for kw in (dict(plan=plan), dict(), dict(time_agg="sum")):
            a = ctx.pv(dev, params, T, Y * X, options=dict(opts, night_skip=False), **kw).numpy()
            b = ctx.pv(dev, params, T, Y * X, options=dict(opts, night_skip=True), **kw).numpy()
            np.testing.assert_array_equal(a, b)
    ori = dict(slope=np.radians(30.0), azimuth=np.radians(180.0))
    trk, tm = opts.get("tracking"), opts.get("trigon_model", "simple")
    with np.errstate(all="ignore"):
        if opts.get("panel_model") == "none":
            ref = orc.convert_irradiation(ds, ori, trk, opts.get("irradiation", "total"), tm, "simple")
        elif opts.get("panel_model") == "solar_thermal":
            ref = orc.convert_solar_thermal(ds, ori, tm, "simple", 0.8, 3.0, 80.0)
        else:
            ref = orc.convert_pv_general(ds, panel, ori, trk, tm, "simple")
    got = ctx.pv(dev, scal, T, Y * X, options=dict(opts, night_skip=True)).numpy()
    ref = np.asarray(ref).reshape(T, -1)
    np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-12 * np.nanmax(np.abs(ref[np.isfinite(ref)])), equal_nan=True)
    # ... and the FUSED kernels (convert + aggregate: for trackers x tails another instantiation than the per-cell one) against
    # the oracle's aggregate of the same cube
    with np.errstate(invalid="ignore"):
        refa = orc.aggregate_matrix(ref, M)
    for ns in (False, True):
        agg = ctx.pv(dev, scal, T, Y * X, plan=plan, options=dict(opts, night_skip=ns)).numpy()
        np.testing.assert_allclose(agg, refa, rtol=1e-10, atol=1e-12 * np.nanmax(np.abs(refa[np.isfinite(refa)])), equal_nan=True)
    # ... with one orientation per cell (trackers x Hay-Davies / bofinger / irradiation: fused kernels of their own)
    ori_pc = dict(slope=np.repeat(lo["slope"], X)[None, :], azimuth=np.repeat(lo["azimuth"], X)[None, :])
    with np.errstate(all="ignore"):
        if opts.get("panel_model") == "none":
            refpc = orc.convert_irradiation(ds, ori_pc, trk, opts.get("irradiation", "total"), tm, "simple")
        elif opts.get("panel_model") == "solar_thermal":
            refpc = orc.convert_solar_thermal(ds, ori_pc, tm, "simple", 0.8, 3.0, 80.0)
        else:
            refpc = orc.convert_pv_general(ds, panel, ori_pc, trk, tm, "simple")
        refpa = orc.aggregate_matrix(np.asarray(refpc).reshape(T, -1), M)
    percell = dict(panel, slope=np.repeat(lo["slope"], X), azimuth=np.repeat(lo["azimuth"], X))
    agg = ctx.pv(dev, percell, T, Y * X, plan=plan, options=dict(opts, night_skip=True)).numpy()
    np.testing.assert_allclose(agg, refpa, rtol=1e-10, atol=1e-12 * np.nanmax(np.abs(refpa[np.isfinite(refpa)])), equal_nan=True)


def test_pv_influx_outflux_dataset_fast_family(ctx):
    """SARAH-shaped datasets (total influx + outflux, no direct / diffuse split, no albedo): pv() with its defaults
    runs in the fast kernel family (Reindl split and albedo = outflux / influx in the converter's head, 48 B/cell)
    instead of the general kernel - per cell, aggregated, with the night early-out (same bits), against the oracle;
    hostile values in the two new streams included."""
    T, Y, X, N = 61, 10, 22, 7
    S = Y * X
    ds = H.pv_dataset(T, Y, X, seed=31)
    infl = ds["influx_direct"] + ds["influx_diffuse"]
    outf = ds["albedo"] * infl
    rng = np.random.default_rng(5)
    day = np.argwhere(infl > 50.0)
    for n, (t, c) in enumerate(day[:: max(1, len(day) // 8)][:8]):
        if n == 0: infl[t, c] = -3.0            # clipped to 0
        if n == 1: infl[t, c] = 1e6             # clipped to toa
        if n == 2: outf[t, c] = np.nan          # albedo NaN -> 0
        if n == 3: outf[t, c] = 5e3             # albedo capped at 1
        if n == 4: infl[t, c] = np.nan
        if n == 5: outf[t, c] = -1.0
    sar = {k: ds[k] for k in ("influx_toa", "temperature", "solar_altitude", "solar_azimuth")}
    sar["influx"], sar["outflux"] = infl, outf
    dev = {k: ctx.upload(v) for k, v in sar.items()}
    ori = dict(slope=np.radians(30.0), azimuth=np.radians(180.0))
    ref = orc.convert_pv_general(sar, H.CSI, ori, clearsky_model="simple")
    got = ctx.pv(dev, PV_PARAMS, T, S, options=dict(clearsky_model="simple")).numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-12 * np.nanmax(np.abs(ref)), equal_nan=True)
    M = H.blob_matrix(N, Y, X, seed=32)
    plan = ctx.plan(M, row_len=X)
    with np.errstate(invalid="ignore"):
        refa = orc.aggregate_matrix(ref, M)
    a = ctx.pv(dev, PV_PARAMS, T, S, plan=plan, options=dict(clearsky_model="simple", night_skip=False)).numpy()
    b = ctx.pv(dev, PV_PARAMS, T, S, plan=plan, options=dict(clearsky_model="simple", night_skip=True)).numpy()
    np.testing.assert_array_equal(a, b)
    np.testing.assert_allclose(a, refa, rtol=1e-10, atol=1e-12 * np.nanmax(np.abs(refa[np.isfinite(refa)])), equal_nan=True)
    for kw in (dict(), dict(time_agg="mean")):  # per-cell kernels with the early-out
        a = ctx.pv(dev, PV_PARAMS, T, S, options=dict(clearsky_model="simple", night_skip=False), **kw).numpy()
        b = ctx.pv(dev, PV_PARAMS, T, S, options=dict(clearsky_model="simple", night_skip=True), **kw).numpy()
        np.testing.assert_array_equal(a, b)
    _, ygrid = H.grid(Y, X)
    lo = orc.orientation_latitude_optimal(np.radians(ygrid))
    pc = dict(H.CSI, slope=np.repeat(lo["slope"], X), azimuth=np.repeat(lo["azimuth"], X))
    refpc = orc.convert_pv_general(sar, H.CSI, dict(slope=np.repeat(lo["slope"], X)[None, :], azimuth=np.repeat(lo["azimuth"], X)[None, :]),
                                   clearsky_model="simple")
    got = ctx.pv(dev, pc, T, S, options=dict(clearsky_model="simple")).numpy()
    np.testing.assert_allclose(got, refpc, rtol=1e-10, atol=1e-12 * np.nanmax(np.abs(refpc)), equal_nan=True)


@pytest.mark.parametrize("both", [False, True])
def test_pv_influx_dataset_with_an_albedo_variable(ctx, both):
    """Total influx and an `albedo` variable (irradiation.py:128-131: "albedo" is used as it is and wins over "outflux"):
    the fast family's influx head with the albedo riding in the outflux's stream (round 6; the general kernel before) -
    either trigon model, either clearsky model, scalar and per-cell orientation, per cell and aggregated, early-out on /
    off with the same bits, against the oracle; hostile albedos (NaN, > 1, negative) stay what they are."""
    T, Y, X, N = 50, 9, 20, 6
    S = Y * X
    ds = H.pv_dataset(T, Y, X, seed=77)
    infl = ds["influx_direct"] + ds["influx_diffuse"]
    alb = ds["albedo"].copy()
    day = np.argwhere(infl > 50.0)
    for n, (t, c) in enumerate(day[:: max(1, len(day) // 6)][:6]):
        alb[t, c] = (np.nan, 3.0, -0.5, 0.0, 1.0, 1e300)[n]
    d = {k: ds[k] for k in ("influx_toa", "temperature", "solar_altitude", "solar_azimuth")}
    d["influx"], d["albedo"] = infl, alb
    d["humidity"] = np.random.default_rng(3).random((T, S))
    if both:
        d["outflux"] = 0.5 * infl  # ignored: albedo wins
    dev = {k: ctx.upload(v) for k, v in d.items()}
    M = H.blob_matrix(N, Y, X, seed=78)
    plan = ctx.plan(M, row_len=X)
    _, ygrid = H.grid(Y, X)
    lo = orc.orientation_latitude_optimal(np.radians(ygrid))
    pc = dict(H.CSI, slope=np.repeat(lo["slope"], X), azimuth=np.repeat(lo["azimuth"], X))
    ori_pc = dict(slope=np.repeat(lo["slope"], X)[None, :], azimuth=np.repeat(lo["azimuth"], X)[None, :])
    ori = dict(slope=np.radians(30.0), azimuth=np.radians(180.0))
    for tm in ("simple", "other"):
        for cs in ("simple", "enhanced"):
            opts = dict(clearsky_model=cs, trigon_model=tm)
            with np.errstate(all="ignore"):
                ref = orc.convert_pv_general(d, H.CSI, ori, None, tm, cs)
                refpc = orc.convert_pv_general(d, H.CSI, ori_pc, None, tm, cs)
                refa = orc.aggregate_matrix(ref, M)
            tol = dict(rtol=1e-10, equal_nan=True)
            got = ctx.pv(dev, PV_PARAMS, T, S, options=opts).numpy()
            np.testing.assert_allclose(got, ref, atol=1e-12 * np.nanmax(np.abs(ref)), **tol)
            got = ctx.pv(dev, pc, T, S, options=opts).numpy()
            np.testing.assert_allclose(got, refpc, atol=1e-12 * np.nanmax(np.abs(refpc)), **tol)
            a = ctx.pv(dev, PV_PARAMS, T, S, plan=plan, options=dict(opts, night_skip=False)).numpy()
            b = ctx.pv(dev, PV_PARAMS, T, S, plan=plan, options=dict(opts, night_skip=True)).numpy()
            np.testing.assert_array_equal(a, b)
            np.testing.assert_allclose(a, refa, atol=1e-12 * np.nanmax(np.abs(refa[np.isfinite(refa)])), **tol)


@pytest.mark.parametrize("R", [16, 17, 28, 40, 48, 70])
def test_dense_tiles_on_the_matrix_cores(ctx, monkeypatch, R):
    """Tiles with >= 16 partial rows are contracted with v_mfma_f64_16x16x4_f64 (groups of 16 rows, a last group from
    12 rows, the rest through the butterfly path): dense rows with explicit zeros and negative weights against scipy,
    for the generic cube, runoff, pv with and without the night early-out; a batch that holds NaN / inf takes the
    guarded path (structural zeros must not leak NaN), ragged time axis, grid rows that are not a multiple of a line."""
    import scipy.sparse as sp

    monkeypatch.setenv("ATLITE_HIP_FORCE_MFMA", "1")
    T, Y, X = 43, 11, 27
    S = Y * X
    rng = np.random.default_rng(R)
    W = rng.normal(size=(R, S))
    W[rng.random((R, S)) < 0.3] = 0.0          # structurally absent
    M = sp.csr_matrix(W)
    M.data[::7] = 0.0                           # explicit zeros stay structural entries
    plan = ctx.plan(M, row_len=X)
    D = rng.normal(size=(T, S)) * 10.0
    out = ctx.spmm(plan, ctx.upload(D)).numpy()
    ref = (M @ D.T)
    np.testing.assert_allclose(out, ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max())
    Dn = D.copy()
    Dn[5, 17] = np.nan
    Dn[20, 3] = np.inf
    Dn[21, 200] = -np.inf
    out = ctx.spmm(plan, ctx.upload(Dn)).numpy()
    with np.errstate(invalid="ignore"):
        ref = np.asarray(M @ Dn.T)
    assert np.isnan(ref).any() and np.isinf(ref).any()
    np.testing.assert_allclose(out, ref, rtol=1e-12, atol=1e-12 * np.abs(ref[np.isfinite(ref)]).max(), equal_nan=True)
    # converters: pv (both kernels) and runoff through the same plan
    ds = H.pv_dataset(T, Y, X, seed=3)
    dev = up(ctx, ds)
    refpv = orc.aggregate_matrix(orc.convert_pv(ds, H.CSI, dict(slope=np.radians(30.0), azimuth=np.radians(180.0))), M)
    for skip in (False, True):
        got = ctx.pv(dev, PV_PARAMS, T, S, plan=plan, options=dict(night_skip=skip)).numpy()
        np.testing.assert_allclose(got, refpv, rtol=1e-10, atol=1e-12 * np.abs(refpv).max())
    ro = rng.random((T, S))
    h = rng.random(S) * 1000.0
    got = ctx.runoff(ctx.upload(ro), ctx.upload(h), T, S, plan=plan).numpy()
    refr = np.asarray(M @ (ro * h[None, :]).T)
    np.testing.assert_allclose(got, refr, rtol=1e-12, atol=1e-12 * np.abs(refr).max())
    for T2 in (1, 9, 16, 17, 64):  # every kind of last sweep: one batch only, a ragged second batch, full
        np.testing.assert_allclose(ctx.spmm(plan, ctx.upload(D[:T2] if T2 <= T else np.tile(D, (2, 1))[:T2])).numpy(),
                                   (M @ (D[:T2] if T2 <= T else np.tile(D, (2, 1))[:T2]).T), rtol=1e-12, atol=1e-12 * np.abs(ref[np.isfinite(ref)]).max())


@pytest.mark.parametrize("T,Y,X,N", [(72, 12, 20, 5), (61, 9, 27, 40)])
def test_pv_night_skip_in_kernel_solar_position(ctx, T, Y, X, N):
    """The night early-out with the in-kernel solar position: night follows from the (T, X) hour-angle table,
    the declination and the latitude before any cube byte is read.  Same bits as without the skip (ragged last
    batch, tiles with more partial rows than the LDS cache holds, NaN / inf inputs in night rows), and the
    oracle's values."""
    from atlite_amd import solar

    ds = H.pv_dataset(T, Y, X, seed=21)
    ds["temperature"][2, :] = np.nan
    ds["influx_direct"][3, :] = np.inf
    x, y = H.grid(Y, X)
    time = H.times(T)
    h, dec = solar.hour_angle(time, x, "0h")
    lat = np.radians(y)
    tables = dict(sin_dec=np.sin(dec), cos_dec=np.cos(dec), h=h, cos_h=np.cos(h), sin_lat=np.sin(lat), cos_lat=np.cos(lat))
    five = {k: ctx.upload(ds[k].reshape(T, -1)) for k in ("influx_direct", "influx_diffuse", "influx_toa", "albedo", "temperature")}
    M = H.blob_matrix(N, Y, X, seed=22)
    plan = ctx.plan(M, row_len=X)
    _, ygrid = H.grid(Y, X)
    lo = orc.orientation_latitude_optimal(np.radians(ygrid))
    for params in (PV_PARAMS, dict(H.CSI, slope=np.repeat(lo["slope"], X), azimuth=np.repeat(lo["azimuth"], X))):
        for kw in (dict(), dict(time_agg="mean")):
            a = ctx.pv(five, params, T, Y * X, plan=plan, solar_tables=tables, options=dict(night_skip=False), **kw).numpy()
            b = ctx.pv(five, params, T, Y * X, plan=plan, solar_tables=tables, options=dict(night_skip=True), **kw).numpy()
            np.testing.assert_array_equal(a, b)
        for kw in (dict(), dict(time_agg="sum")):  # per-cell kernels (series, capacity-factor map)
            a = ctx.pv(five, params, T, Y * X, solar_tables=tables, options=dict(night_skip=False), **kw).numpy()
            b = ctx.pv(five, params, T, Y * X, solar_tables=tables, options=dict(night_skip=True), **kw).numpy()
            np.testing.assert_array_equal(a, b)
    alt, az = orc.solar_position(time, x, y, "0h")
    full = dict(ds, solar_altitude=alt.reshape(T, -1), solar_azimuth=az.reshape(T, -1))
    ref = orc.aggregate_matrix(orc.convert_pv(full, H.CSI, dict(slope=np.radians(30.0), azimuth=np.radians(180.0))), M)
    got = ctx.pv(five, PV_PARAMS, T, Y * X, plan=plan, solar_tables=tables, options=dict(night_skip=True)).numpy()
    assert (got == 0).any() and got.max() > 0
    np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-12 * np.nanmax(np.abs(ref[np.isfinite(ref)])), equal_nan=True)


@pytest.mark.parametrize("tile", ["16x8", "32x4", "64x2"])
def test_every_cell_owned_exactly_once(ctx, monkeypatch, tile):
    """A 128-byte line that straddles two grid rows belongs to ONE tile (the lower row's first tile
    column reaches back into the upper row's tail): with a dense matrix every cell must be counted
    exactly once, for every residue of X modulo 16 and for grids shorter than a tile."""
    import scipy.sparse as sp

    monkeypatch.setenv("ATLITE_HIP_TILE", tile)
    rng = np.random.default_rng(5)
    for Y, X in [(5, x) for x in range(17, 34)] + [(1, 37), (2, 7), (9, 3), (17, 15), (11, 200)]:
        S, T = Y * X, 11
        D = rng.standard_normal((T, S))
        W = rng.random((3, S)) + 0.5  # every cell in every shape
        W[1, :] = 1.0  # plain sum of all cells: a cell counted twice or never shows up at once
        M = sp.csr_matrix(W)
        out = ctx.spmm(ctx.plan(M, row_len=X, cache=False), ctx.upload(D)).numpy()
        close(out, W @ D.T, atol_scale=1e-13)


def test_wind_repeated_first_knot(ctx):
    """np.interp answers F[0] left of the table but the upper duplicate's value AT a repeated first knot;
    the table builder adds a guard knot one ulp below for that (found by the host-side interp test)."""
    T, Y, X = 6, 2, 8
    rng = np.random.default_rng(0)
    V = np.array([3.0, 3.0, 5.0, 9.0, 12.0, 25.0, 25.0])
    POW = np.array([0.1, 0.4, 0.9, 2.0, 3.0, 3.0, 0.0])
    wnd = rng.uniform(0.0, 30.0, (T, Y * X))
    wnd[0, :6] = [3.0, np.nextafter(3.0, 0), 2.0, 0.0, 25.0, 26.0]
    out = ctx.wind(ctx.upload(wnd), None, V, POW / 3.0, 100.0, 100.0, None, T, Y * X).numpy()  # method None: no extrapolation
    ref = np.interp(wnd, V, POW / 3.0)
    np.testing.assert_allclose(out, ref, rtol=1e-14, atol=1e-16)
    assert out[0, 0] == ref[0, 0] and out[0, 1] == ref[0, 1] == POW[0] / 3.0 and out[0, 2] == POW[0] / 3.0


def test_wind_grid_lookup_equals_knot_search(ctx, monkeypatch):
    """Power curves whose knots are all multiples of 1, 1/2, 1/4 or 1/8 m/s take a bucket lookup instead of
    the binary knot search (interp_grid, atl_math.h): both must select the same interval, i.e. give the
    same bits - for every shipped turbine, hostile wind speeds, per cell and aggregated."""
    from atlite_amd.resource import get_windturbineconfig, windturbines

    T, Y, X = 16, 6, 20
    rng = np.random.default_rng(3)
    wnd = 30.0 * rng.random((T, Y * X)) ** 1.5
    z0 = np.exp(np.log(1e-3) + rng.random((T, Y * X)) * np.log(1.5e3))
    wnd[0, :10] = [0.0, 2.0, 25.0, np.nextafter(25.0, 0), np.nextafter(25.0, 30), 30.0, np.nan, np.inf, -1.0, 12.5]
    d_w, d_z = ctx.upload(wnd), ctx.upload(z0)
    M = H.blob_matrix(3, Y, X, seed=1)
    n_grid = 0
    for name in windturbines():
        tb = get_windturbineconfig(name)
        V, F = np.asarray(tb["V"], float), np.asarray(tb["POW"], float) / tb["P"]
        aligned = bool(np.all(V * 8 == np.floor(V * 8)))
        n_grid += aligned
        args = (d_w, d_z, V, F, float(tb["hub_height"]), 100.0, "logarithmic", T, Y * X)
        monkeypatch.delenv("ATLITE_HIP_WIND_NO_GRID", raising=False)
        a, fa = ctx.wind(*args).numpy(), ctx.wind(*args[:8], Y * X, time_agg="mean").numpy()
        lane = ctx.wind(d_w, None, V, F, 80.0, 80.0, None, T, Y * X).numpy()
        monkeypatch.setenv("ATLITE_HIP_WIND_NO_GRID", "1")
        b, fb = ctx.wind(*args).numpy(), ctx.wind(*args[:8], Y * X, time_agg="mean").numpy()
        np.testing.assert_array_equal(a, b, err_msg=name)
        np.testing.assert_array_equal(fa, fb, err_msg=name)
        np.testing.assert_allclose(lane, np.interp(wnd, V, F), rtol=1e-14, atol=1e-16, err_msg=name)  # fast lane vs np.interp
    assert n_grid >= 20  # the lookup is what the shipped turbines actually run


def test_indicator_matrix_on_the_device(ctx):
    """SURVEY 8 f-2, device half: area(shape n cell) / area(cell) as line integrals on the GPU against the host
    polygon clipper (same contract, atlite/gis.py:104-145) - rectangles with exact answers, ring orientation,
    holes, multi-part shapes, shapes reaching past the grid, finely digitised borders (many edges per cell column),
    area conservation of star polygons, a tessellation covering every cell exactly once."""
    import time

    from atlite_amd import gis

    x, y = np.arange(10.0), 10.0 + 2.0 * np.arange(5.0)
    cell = np.array([[1.5, 11], [2.5, 11], [2.5, 13], [1.5, 13]])
    M = gis.compute_indicatormatrix(x, y, [cell], ctx=ctx)
    assert M.nnz == 1 and abs(M[0, 1 * 10 + 2] - 1.0) < 1e-13
    rect = np.array([[0.25, 9.5], [3.5, 9.5], [3.5, 12.0], [0.25, 12.0]])
    hole = np.array([[1.0, 10.0], [2.0, 10.0], [2.0, 11.0], [1.0, 11.0]])
    far = rect + np.array([5.0, 0.0])
    outside = rect + np.array([8.0, 6.0])          # partly past the grid's upper right corner
    nowhere = rect + np.array([100.0, 100.0])      # no overlap at all: an empty row
    shapes = [rect, rect[::-1], dict(exterior=rect, holes=[hole]), [rect, far], outside, nowhere]
    D = gis.compute_indicatormatrix(x, y, shapes, ctx=ctx)
    Hm = gis.compute_indicatormatrix(x, y, shapes)
    assert D.shape == Hm.shape and D[5].nnz == 0
    np.testing.assert_allclose(D.toarray(), Hm.toarray(), rtol=0, atol=1e-13)
    assert abs(D[2].sum() * 2.0 - (3.25 * 2.5 - 1.0)) < 1e-12

    X = Y = 120
    xx, yy = -25 + (70 / X) * np.arange(X), 30 + (42 / Y) * np.arange(Y)
    ca = (70 / X) * (42 / Y)
    polys = gis.random_star_polygons(40, (-15, 35, 35, 66), seed=3)
    # a finely digitised border: 4000 vertices on a wobbly circle, ~100 edges per cell column
    th = np.linspace(0, 2 * np.pi, 4000, endpoint=False)
    rad = 9.0 + 0.8 * np.sin(17 * th) + 0.3 * np.cos(61 * th)
    polys.append(np.stack([5.0 + 1.6 * rad * np.cos(th), 50.0 + rad * np.sin(th)], axis=1))
    Dm = gis.compute_indicatormatrix(xx, yy, polys, ctx=ctx)
    Hm = gis.compute_indicatormatrix(xx, yy, polys)
    np.testing.assert_allclose(Dm.toarray(), Hm.toarray(), rtol=0, atol=1e-12)
    for i, p in enumerate(polys[:-1]):
        a = 0.5 * abs(np.sum(p[:, 0] * np.roll(p[:, 1], -1) - np.roll(p[:, 0], -1) * p[:, 1]))
        assert abs(Dm[i].sum() * ca - a) < 1e-10 * a
    tess = gis.random_tessellation(50, (xx[0] - 35 / X, yy[0] - 21 / Y, xx[-1] + 35 / X, yy[-1] + 21 / Y), seed=1)
    t0 = time.perf_counter()
    Dt = gis.compute_indicatormatrix(xx, yy, tess, ctx=ctx)
    t_dev = time.perf_counter() - t0
    t0 = time.perf_counter()
    Ht = gis.compute_indicatormatrix(xx, yy, tess)
    t_host = time.perf_counter() - t0
    np.testing.assert_allclose(np.asarray(Dt.sum(0)).ravel(), 1.0, rtol=0, atol=1e-11)
    np.testing.assert_allclose(Dt.toarray(), Ht.toarray(), rtol=0, atol=1e-12)
    print(f"indicator matrix 50 shapes x 120x120: device {t_dev * 1e3:.1f} ms, host {t_host * 1e3:.1f} ms")

