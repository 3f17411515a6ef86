#!/usr/bin/env python3
"""
Freeze golden input/output vectors by executing the REFERENCE's own source files
(/root/reference/atlite/convert.py, aggregate.py, wind.py, resource.py, pv/*.py) under the
xarray/dask stand-in of ``refshim.py``.  Run in the build container only:

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz

The .npz files are small (T<=77, 6x8 grid) and committed; tests never need /root/reference.
"""

from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pandas as pd
import scipy.sparse as sp

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import refshim  # noqa: E402

refshim.install()
import xarray as xr  # noqa: E402  (the stand-in)

conv = refshim.reference("atlite.convert")
res = refshim.reference("atlite.resource")
solpos = refshim.reference("atlite.pv.solar_position")
orient = refshim.reference("atlite.pv.orientation")

Y, X = 6, 8
x = -25.0 + (70.0 / X) * np.arange(X)
y = 30.0 + (42.0 / Y) * np.arange(Y)


def dataset(variables, time):
    coords = {"time": time, "y": y, "x": x}
    ds = xr.Dataset(
        {k: xr.DataArray(v, dims=["time", "y", "x"][-v.ndim:], coords={d: coords[d] for d in ["time", "y", "x"][-v.ndim:]})
         for k, v in variables.items()},
        coords={"time": time, "y": y, "x": x, "lon": ("x", x), "lat": ("y", y)},
    )
    return ds


class MockCutout:  # as in the reference's test/test_aggregate_time.py:13-17
    def __init__(self, data):
        self.data = data
        grid_coords = np.array([(xx, yy) for yy in data["y"].values for xx in data["x"].values])
        self.grid = pd.DataFrame(grid_coords, columns=["x", "y"])


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        v = v.values if hasattr(v, "values") and not isinstance(v, np.ndarray) else v
        out[k] = np.asarray(v)
    np.savez_compressed(HERE / f"{name}.npz", **out)
    print(f"  wrote {name}.npz: " + ", ".join(f"{k}{tuple(v.shape)}" for k, v in out.items()))


def main():
    rng = np.random.default_rng(20260925)
    # ---------------------------------------------------------------- solar position ------
    t = pd.date_range("2013-03-19 18:00", periods=77, freq="h")
    ds0 = dataset({"dummy": np.zeros((len(t), Y, X))}, t)
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore", DeprecationWarning)
        sp_ = solpos.SolarPosition(ds0, time_shift="-30min")  # era5.py:185-186
        sp0 = solpos.SolarPosition(ds0)
    alt, az = sp_["altitude"].values, sp_["azimuth"].values
    save("solar_position", time=t.values.astype("datetime64[ns]").astype(np.int64), x=x, y=y,
         altitude_shift30=alt, azimuth_shift30=az, altitude_noshift=sp0["altitude"].values,
         azimuth_noshift=sp0["azimuth"].values)

    # ---------------------------------------------------------------- pv --------------------
    T = len(t)
    toa = 1361.0 * np.maximum(np.sin(alt), 0.0)
    kt = 0.2 + 0.55 * rng.random((T, Y, X))
    fd = 0.3 + 0.5 * rng.random((T, Y, X))
    v = dict(
        influx_direct=toa * kt * fd,
        influx_diffuse=toa * kt * (1 - fd),
        influx_toa=toa,
        albedo=0.05 + 0.3 * rng.random((T, Y, X)),
        temperature=283.15 + 10 * rng.standard_normal((T, Y, X)),
        solar_altitude=alt,
        solar_azimuth=az,
    )
    # edge cases the reference's clip / fillna / mask logic must handle
    day = np.argwhere(toa > 300.0)
    e = [tuple(i) for i in day[:: max(1, len(day) // 12)][:12]]
    v["influx_direct"][e[0]] = -5.0  # clipped to 0
    v["influx_direct"][e[1]] = 5000.0  # clipped to toa -> diffuse clipped to 0
    v["influx_diffuse"][e[2]] = 5000.0  # clipped to toa - direct
    v["temperature"][e[3]] = np.nan  # eff NaN -> 0
    v["albedo"][e[4]] = np.nan  # ground_t NaN -> fillna 0
    v["influx_diffuse"][e[5]] = -1.0
    v["solar_altitude"][e[6]] = np.radians(1.0)  # exactly at the threshold: not capped
    v["solar_altitude"][e[7]] = np.nextafter(np.radians(1.0), 0)  # just below: capped
    v["influx_direct"][e[8]] = 0.004
    v["influx_diffuse"][e[8]] = 0.006  # direct + diffuse <= 0.01 (exactly 0.01): capped
    v["solar_altitude"][e[9]] = np.nan  # NaN altitude: not capped, k NaN -> direct_t 0
    v["temperature"][e[10]] = 400.0  # eff negative -> clipped at 0
    v["influx_toa"][e[11]] = np.nan
    ds = dataset(v, t)
    pv_out = {}
    for panel_name in ("CSi", "CdTe"):
        panel = res.get_solarpanelconfig(panel_name)
        for oname, ospec in (("const30_180", {"slope": 30.0, "azimuth": 180.0}),
                             ("const0_0", {"slope": 0.0, "azimuth": 0.0}),
                             ("latopt", "latitude_optimal"),
                             ("latitude", {"name": "latitude", "azimuth": 170.0})):
            o = orient.get_orientation(dict(ospec) if isinstance(ospec, dict) else ospec)
            da = conv.convert_pv(ds, panel, o, tracking=None)
            pv_out[f"out_{panel_name}_{oname}"] = da.transpose("time", "y", "x").values
    save("pv", time=t.values.astype("datetime64[ns]").astype(np.int64), x=x, y=y, **v, **pv_out)

    # ---------------------------------------------------------------- pv options (8 f-1) -----
    opt = {}
    csi = res.get_solarpanelconfig("CSi")
    kan = res.get_solarpanelconfig("KANENA")
    o30 = {"slope": 30.0, "azimuth": 180.0}
    for trk in ("horizontal", "tilted_horizontal", "vertical", "dual"):
        for tm in ("simple", "other"):
            da = conv.convert_pv(ds, csi, orient.get_orientation(dict(o30)), tracking=trk, trigon_model=tm)
            opt[f"pv_{trk}_{tm}"] = da.transpose("time", "y", "x").values
    opt["pv_none_other"] = conv.convert_pv(ds, csi, orient.get_orientation(dict(o30)), tracking=None,
                                           trigon_model="other").transpose("time", "y", "x").values
    opt["pv_kanena_simple"] = conv.convert_pv(ds, kan, orient.get_orientation(dict(o30)), tracking=None
                                              ).transpose("time", "y", "x").values
    opt["pv_kanena_latopt_other"] = conv.convert_pv(ds, kan, orient.get_orientation("latitude_optimal"), tracking=None,
                                                    trigon_model="other").transpose("time", "y", "x").values
    for q in ("total", "direct", "diffuse", "ground"):
        for tm in ("simple", "other"):
            da = conv.convert_irradiation(ds, orient.get_orientation(dict(o30)), irradiation=q, trigon_model=tm)
            opt[f"irr_{q}_{tm}"] = da.transpose("time", "y", "x").values
    opt["irr_total_dual"] = conv.convert_irradiation(ds, orient.get_orientation(dict(o30)), tracking="dual"
                                                     ).transpose("time", "y", "x").values
    opt["thermal_default"] = conv.convert_solar_thermal(ds, orient.get_orientation({"slope": 45.0, "azimuth": 180.0}),
                                                        "simple", "simple", 0.8, 3.0, 80.0).transpose("time", "y", "x").values
    # SARAH-like dataset: total influx + outflux, no direct/diffuse split, no albedo
    v2 = dict(influx=v["influx_direct"] + v["influx_diffuse"], influx_toa=v["influx_toa"],
              outflux=(v["influx_direct"] + v["influx_diffuse"]) * np.where(np.isnan(v["albedo"]), 0.2, v["albedo"]),
              temperature=v["temperature"], humidity=0.3 + 0.5 * rng.random((T, Y, X)),
              solar_altitude=v["solar_altitude"], solar_azimuth=v["solar_azimuth"])
    v2["influx"][e[0]] = 0.0          # influx == 0 -> albedo 0/NaN -> fillna 0
    v2["outflux"][e[1]] = 1e9         # albedo clipped to 1
    ds2 = dataset(v2, t)
    for cs in ("simple", "enhanced"):
        opt[f"pv_influx_{cs}"] = conv.convert_pv(ds2, csi, orient.get_orientation(dict(o30)), tracking=None,
                                                 clearsky_model=cs).transpose("time", "y", "x").values
    opt["pv_influx_enhanced_other"] = conv.convert_pv(ds2, csi, orient.get_orientation(dict(o30)), tracking=None,
                                                      trigon_model="other", clearsky_model="enhanced"
                                                      ).transpose("time", "y", "x").values
    save("pv_options", influx=v2["influx"], outflux=v2["outflux"], humidity=v2["humidity"], **opt)

    # ---------------------------------------------------------------- wind ------------------
    Tw = 40
    tw = pd.date_range("2013-01-01", periods=Tw, freq="h")
    u = rng.random((Tw, Y, X))
    w = dict(
        wnd100m=8.0 * np.sqrt(-np.log1p(-u)) * (2 / np.sqrt(np.pi)),
        roughness=np.exp(np.log(1e-3) + rng.random((Tw, Y, X)) * np.log(1.5e3)),
        wnd_shear_exp=0.05 + 0.3 * rng.random((Tw, Y, X)),
    )
    w["wnd100m"][0, 0, :8] = [0.0, 2.0, 25.0, 24.999999, 30.0, np.nan, np.inf, 13.0]
    w["wnd100m"][1, 0, :4] = [1e-300, 3.0, 12.0, 25.0000001]
    w["roughness"][2, 0, :3] = [2e-4, 0.0, 100.0]  # sanitized floor / degenerate / == from_height
    dsw = dataset(w, tw)
    wout = {}
    for tname in ("Vestas_V112_3MW", "Enercon_E101_3000kW", "NREL_ReferenceTurbine_5MW_offshore"):
        turb = res.get_windturbineconfig(tname, add_cutout_windspeed=False)
        wout[f"{tname}_V"] = np.asarray(turb["V"], dtype=float)
        wout[f"{tname}_POW"] = np.asarray(turb["POW"], dtype=float)
        wout[f"{tname}_P_hub"] = np.array([turb["P"], turb["hub_height"]], dtype=float)
        for m in ("logarithmic", "power"):
            wout[f"out_{tname}_{m}"] = conv.convert_wind(dsw, turb, m).values
    turb = res.get_windturbineconfig("Vestas_V112_3MW", add_cutout_windspeed=False)
    sm = res.windturbine_smooth(turb, params=True)
    wout["smooth_V"], wout["smooth_POW"], wout["smooth_P"] = sm["V"], sm["POW"], np.array([sm["P"]])
    wout["out_smooth_logarithmic"] = conv.convert_wind(dsw, sm, "logarithmic").values
    # fast lane: hub-height wind speed present in the dataset (wind.py:76-78)
    ds80 = dataset({"wnd80m": w["wnd100m"], "wnd100m": 2 * w["wnd100m"], "roughness": w["roughness"]}, tw)
    wout["out_fastlane"] = conv.convert_wind(ds80, turb, "logarithmic").values
    # padding rule of add_cutout_windspeed (test/test_resource.py:28-34)
    padded = res.get_windturbineconfig(dict(V=[0, 10, 20], POW=[0, 1.0, 1.0], P=1.0, hub_height=90.0),
                                       add_cutout_windspeed=True)
    wout["padded_V"], wout["padded_POW"] = np.asarray(padded["V"], float), np.asarray(padded["POW"], float)
    save("wind", time=tw.values.astype("datetime64[ns]").astype(np.int64), x=x, y=y, **w, **wout)

    # ---------------------------------------------------------------- heat demand / runoff ---
    Th = 24 * 3 + 5
    th = pd.date_range("2013-01-01", periods=Th, freq="h")
    temp = 283.15 + 8 * rng.standard_normal((Th, Y, X))
    temp[5, 3, 3] = np.nan
    temp[30:54, 1, 1] = np.nan  # a whole day missing for one cell
    dsh = dataset({"temperature": temp}, th)
    hout = {}
    for shift in (0.0, 4.0, -5.0):
        da = conv.convert_heat_demand(dsh, threshold=15.0, a=1.3, constant=0.2, hour_shift=shift)
        hout[f"out_shift{shift:+.0f}"] = da.values
        hout[f"days_shift{shift:+.0f}"] = da.coords["time"].values.astype("datetime64[ns]").astype(np.int64)
    for shift in (0.0, 3.0):
        da = conv.convert_cooling_demand(dsh, threshold=3.0, a=0.7, constant=0.1, hour_shift=shift)
        hout[f"cool_shift{shift:+.0f}"] = da.values
    soil = 278.15 + 5 * rng.standard_normal((Th, Y, X))
    soil[:, 0, :3] = np.nan  # sea
    dew = 275.15 + 6 * rng.standard_normal((Th, Y, X))
    dst = dataset({"temperature": temp, "soil temperature": soil, "dewpoint temperature": dew}, th)
    hout["soil"], hout["dew"] = soil, dew
    hout["out_temperature"] = conv.convert_temperature(dst).values
    hout["out_soil_temperature"] = conv.convert_soil_temperature(dst).values
    hout["out_dewpoint_temperature"] = conv.convert_dewpoint_temperature(dst).values
    hout["out_cop_air"] = conv.convert_coefficient_of_performance(dst, "air", 55.0, None, None, None).values
    hout["out_cop_soil"] = conv.convert_coefficient_of_performance(dst, "soil", 45.0, None, -0.14, None).values
    save("heat_demand", time=th.values.astype("datetime64[ns]").astype(np.int64), x=x, y=y, temperature=temp, **hout)

    ro = -1e-4 * np.log1p(-rng.random((Th, Y, X)))
    height = 2000 * rng.random((Y, X))
    dsr = dataset({"runoff": ro, "height": height}, th)
    save("runoff", time=th.values.astype("datetime64[ns]").astype(np.int64), x=x, y=y, runoff=ro, height=height,
         out_weighted=conv.convert_runoff(dsr).values, out_plain=conv.convert_runoff(dsr, weight_with_height=False).values)

    # ---------------------------------------------------------------- gateway ---------------
    cut = MockCutout(ds)
    N = 5
    M = sp.random(N, Y * X, density=0.3, random_state=7, format="csr")
    M.data[:] = rng.random(M.nnz)
    lay = 3.0 * rng.random((Y, X))
    lay[0, 0] = 0.0
    layout = xr.DataArray(lay, dims=["y", "x"], coords={"y": y, "x": x})
    panel = res.get_solarpanelconfig("CSi")
    kw = dict(panel=panel, orientation={"slope": 30.0, "azimuth": 180.0})
    g = dict(matrix_indptr=M.indptr, matrix_indices=M.indices, matrix_data=M.data, layout=lay)
    import warnings

    def run(**k):
        r = conv.pv(cut_bound, **kw, **k)
        return r

    # convert.pv calls cutout.convert_and_aggregate: bind the reference gateway on the mock
    MockCutout.convert_and_aggregate = conv.convert_and_aggregate
    cut_bound = cut
    g["series_matrix"] = run(matrix=M, aggregate_time=None).values  # (N, T)
    g["mean_matrix"] = run(matrix=M, aggregate_time="mean").values
    g["sum_matrix"] = run(matrix=M, aggregate_time="sum").values
    g["series_layout"] = run(layout=layout, aggregate_time=None).values  # (1, T)
    g["series_matrix_layout"] = run(matrix=M, layout=layout, aggregate_time=None).values
    r, cap = run(matrix=M, layout=layout, per_unit=True, return_capacity=True, aggregate_time=None)
    g["pu_matrix_layout"], g["capacity_matrix_layout"] = r.values, cap.values
    g["pu_mean_matrix"] = run(matrix=M, per_unit=True, aggregate_time="mean").values
    g["cells_mean"] = run(aggregate_time="mean").values  # (y, x)
    g["cells_sum"] = run(aggregate_time="sum").values
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", FutureWarning)
        g["legacy_nomatrix"] = run().values  # legacy = time sum
        g["legacy_matrix"] = run(matrix=M).values  # legacy = series
        g["capfactor"] = run(capacity_factor=True).values
    save("gateway_pv", **g)

    # ---------------------------------------------------------------- runoff() post-processing ---
    # convert.py:1037-1084 (smooth / lower_threshold_quantile / normalize_using_yearly) needs whole years: two
    # years + a stub on a 3 x 4 grid.  The input cube is NOT stored: the test regenerates it from the seed
    # (tests/helpers.py: runoff_post_inputs); outputs are kept at the sampled time steps `sel`.
    sys.path.insert(0, str(HERE.parents[1]))
    from tests.helpers import runoff_post_inputs, runoff_post_sample  # seeded inputs shared with the tests

    ro2, height2, M2, names, t2, y2, x2 = runoff_post_inputs()
    ds2 = xr.Dataset(
        {"runoff": xr.DataArray(ro2, dims=["time", "y", "x"], coords={"time": t2, "y": y2, "x": x2}),
         "height": xr.DataArray(height2, dims=["y", "x"], coords={"y": y2, "x": x2})},
        coords={"time": t2, "y": y2, "x": x2, "lon": ("x", x2), "lat": ("y", y2)})
    cut2 = MockCutout(ds2)
    idx = pd.Index(names, name="countries")
    sel = runoff_post_sample(len(t2))
    yearly_dt = pd.DataFrame([[3.0, 5.0, 1.5], [2.0, 4.0, 1.0], [7.0, 7.0, 7.0]],
                             index=pd.to_datetime(["2012-01-01", "2013-01-01", "2015-01-01"]), columns=names)
    yearly_str = pd.DataFrame([[1.0, 2.0, 3.0], [3.0, 5.0, 1.5], [2.0, 4.0, 1.0]], index=["2011", "2012", "2013"],
                              columns=names)
    rp = dict(M=M2.toarray(), height=height2, sel=sel)
    cases = {
        "plain": dict(),
        "smooth_true": dict(smooth=True),
        "smooth24_q": dict(smooth=24, lower_threshold_quantile=True),
        "q30": dict(lower_threshold_quantile=0.3),
        "norm_dt_smooth48": dict(smooth=48, normalize_using_yearly=yearly_dt),
        "norm_str": dict(normalize_using_yearly=yearly_str),
        "noheight_norm": dict(normalize_using_yearly=yearly_str, weight_with_height=False),
    }
    for name, kw in cases.items():
        r = conv.runoff(cut2, matrix=M2, index=idx, **kw)
        assert r.dims == ("countries", "time"), r.dims
        rp[name] = r.values[:, sel]
    save("runoff_post", **rp)

    # ---------------------------------------------------------------- orientation callbacks that read the sun ---
    # pv/orientation.py:104-107: orientation(lon, lat, solar_position) may return angles that depend on time
    from tests.helpers import orientation_follow_sun

    cb = {}
    for tm in ("simple", "other"):
        da = conv.convert_pv(ds, csi, orientation_follow_sun, tracking=None, trigon_model=tm)
        cb[f"follow_{tm}"] = da.transpose("time", "y", "x").values
    ds5 = dataset({k: v[k] for k in ("influx_direct", "influx_diffuse", "influx_toa", "albedo", "temperature")}, t)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", DeprecationWarning)
        cb["follow_computed_position"] = conv.convert_pv(ds5, csi, orientation_follow_sun, tracking=None).transpose("time", "y", "x").values
    save("pv_callback", **cb)

    # ---------------------------------------------------------------- extrapolate_wind_speed on its own (wind.py:23-125) ---
    # (the wind section's inputs: no further random draws, so every file above stays as it was)
    windmod = refshim.reference("atlite.wind")
    ws = {}
    ds2h = dataset(dict(w, wnd10m=0.7 * w["wnd100m"]), tw)
    for name, args, kw in (("log_80", (80,), {}), ("power_120p5", (120.5,), dict(method="power")), ("log_30_closest_is_10", (30,), {}),
                           ("log_30_from_100", (30,), dict(from_height=100)), ("power_15p5_from_100", (15.5,), dict(from_height=100, method="power"))):
        da = windmod.extrapolate_wind_speed(ds2h, *args, **kw)
        ws[name] = da.transpose("time", "y", "x").values
        ws[name + "_long_name"] = np.array(da.attrs["long name"])
        ws[name + "_name"] = np.array(str(da.name))
    # (wnd{int(to_height)}m present -> returned as it is, wind.py:76-78: 10.5 m finds wnd10m)
    ws["fastlane_10p5"] = windmod.extrapolate_wind_speed(ds2h, 10.5).transpose("time", "y", "x").values
    save("wind_speed", **ws)
    print("done")


if __name__ == "__main__":
    main()
