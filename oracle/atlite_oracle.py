"""
CPU ORACLE — test infrastructure only, NOT a product path.

A NumPy restatement of the arithmetic PyPSA/atlite performs on the convert_and_aggregate hot
path, eager, fp64, written for fidelity of *operation order* with the reference, not for
speed.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module; ``atlite_amd`` never does.

Every function cites the reference lines it follows (paths relative to /root/reference).
xarray semantics that the reference relies on are encoded by hand and noted inline:
``clip`` = np.clip (NaN propagates), ``fillna``, ``where``, broadcasting by dimension name
(here: arrays are (time, y, x) or (time, cell); per-y / per-x vectors are reshaped by the
caller), ``sum/mean("time")`` = nan-skipping reductions, ``resample("1D").mean`` = nan-skipping
mean over calendar-day bins.

Pinning status: the pv / wind / heat-demand / runoff arithmetic is pinned against vectors
produced by executing the reference's own source files (atlite/pv/*.py, atlite/wind.py,
atlite/convert.py, atlite/aggregate.py) under a minimal xarray/dask stand-in - see
tests/golden/make_golden.py and tests/test_oracle_golden.py (45 cases, bit-exact for wind /
solar position / runoff / temperatures, <= 5e-15 relative elsewhere).  The stand-in is itself
validated by the reference's own test/test_aggregate_time.py running green under it
(tests/golden/run_reference_tests.py).  Beyond the fixed vectors, tests/golden/fuzz_oracle_vs_reference.py
compares this module with the reference's code on random points of the option space with hostile values
(1380 cases incl. the gateway algebra: no mismatch, results identical bit for bit; log beside the script).  The reference ships no
numeric golden vectors of its own for this path.
"""

from __future__ import annotations

import numpy as np
import pandas as pd
import scipy.sparse as sp

# --------------------------------------------------------------------------------------
# solar position
# --------------------------------------------------------------------------------------


def solar_position_tables(time, time_shift="0h"):
    """
    Per-time quantities of the almanac algorithm, atlite/pv/solar_position.py:71-97.

    Returns dict with n, hour, minute, ra, dec and ``lmst0`` (the longitude-independent part
    of the local mean sidereal time in degrees), all shape (T,).
    """
    t = pd.DatetimeIndex(time) + pd.to_timedelta(time_shift)
    n = np.asarray(t.to_julian_date(), dtype=np.float64) - 2451545.0  # :73-74
    hour = np.asarray(t.hour, dtype=np.int64)  # :75
    minute = np.asarray(t.minute, dtype=np.int64)  # :76
    L = 280.460 + 0.9856474 * n  # :86
    g = np.radians(357.528 + 0.9856003 * n)  # :87
    l = np.radians(L + 1.915 * np.sin(g) + 0.020 * np.sin(2 * g))  # :88  # noqa: E741
    ep = np.radians(23.439 - 4e-7 * n)  # :89
    ra = np.arctan2(np.cos(ep) * np.sin(l), np.cos(l))  # :91
    lmst0 = (6.697375 + (hour + minute / 60.0) + 0.0657098242 * n) * 15.0  # :92
    dec = np.arcsin(np.sin(ep) * np.sin(l))  # :97
    return dict(n=n, hour=hour, minute=minute, ra=ra, dec=dec, lmst0=lmst0)


def solar_position(time, lon_deg, lat_deg, time_shift="0h"):
    """
    atlite/pv/solar_position.py:71-114 (compute branch).  lon_deg (X,), lat_deg (Y,) in
    degrees.  Returns (altitude, azimuth), each (T, Y, X) radians.
    """
    tb = solar_position_tables(time, time_shift)
    lon = np.asarray(lon_deg, dtype=np.float64)
    lat = np.radians(np.asarray(lat_deg, dtype=np.float64))  # :100
    lmst = tb["lmst0"][:, None] + lon[None, :]  # :92-94  (time, x)
    h = (np.radians(lmst) - tb["ra"][:, None] + np.pi) % (2 * np.pi) - np.pi  # :95
    dec = tb["dec"][:, None, None]
    latb = lat[None, :, None]
    hb = h[:, None, :]
    # :103-105
    alt = np.arcsin(
        np.clip(np.sin(dec) * np.sin(latb) + np.cos(dec) * np.cos(latb) * np.cos(hb), -1.0, 1.0)
    )
    # :109-113
    az = np.arccos(
        np.clip((np.sin(dec) * np.cos(latb) - np.cos(dec) * np.sin(latb) * np.cos(hb)) / np.cos(alt), -1.0, 1.0)
    )
    az = np.where(hb <= 0, az, 2 * np.pi - az)  # :114
    return alt, az


# --------------------------------------------------------------------------------------
# orientation
# --------------------------------------------------------------------------------------


def orientation_constant(slope_deg, azimuth_deg):
    """make_constant, atlite/pv/orientation.py:72-79: degrees -> radians scalars."""
    return dict(slope=np.radians(slope_deg), azimuth=np.radians(azimuth_deg))


def orientation_latitude_optimal(lat_rad):
    """make_latitude_optimal, atlite/pv/orientation.py:50-67; lat_rad (Y,) radians."""
    lat = np.asarray(lat_rad, dtype=np.float64)
    slope = np.empty_like(lat)
    below_25 = np.abs(lat) <= np.radians(25)
    below_50 = np.abs(lat) <= np.radians(50)
    slope[below_25] = 0.87 * np.abs(lat[below_25])
    slope[~below_25 & below_50] = 0.76 * np.abs(lat[~below_25 & below_50]) + np.radians(0.31)
    slope[~below_50] = np.radians(40.0)
    azimuth = np.where(lat < 0, 0, np.pi).astype(np.float64)
    return dict(slope=slope, azimuth=azimuth)


def orientation_latitude(lat_rad, azimuth_deg=180):
    """make_latitude, atlite/pv/orientation.py:82-88."""
    return dict(slope=np.asarray(lat_rad, dtype=np.float64), azimuth=np.radians(azimuth_deg))


def surface_orientation(alt, az, slope, azimuth):
    """
    SurfaceOrientation with tracking=None, atlite/pv/orientation.py:114-117,188.
    slope/azimuth must already broadcast against alt/az.
    """
    cosincidence = np.sin(slope) * np.cos(alt) * np.cos(azimuth - az) + np.cos(slope) * np.sin(alt)
    cosincidence = np.clip(cosincidence, 0, None)  # :188 clip(min=0)
    return cosincidence


# --------------------------------------------------------------------------------------
# irradiation + panel
# --------------------------------------------------------------------------------------


def _fillna0(a):
    return np.where(np.isnan(a), 0.0, a)


def tilted_irradiation(influx_direct, influx_diffuse, influx_toa, albedo, alt, cosincidence, slope,
                       altitude_threshold=1.0):
    """
    TiltedIrradiation, ERA5 branch + trigon_model="simple" + irradiation="total":
    atlite/pv/irradiation.py:196-208 (clip), 214-226 (simple model), 247-255 (mask).
    """
    with np.errstate(divide="ignore", invalid="ignore"):
        direct = np.clip(influx_direct, 0, influx_toa)  # :207
        diffuse = np.clip(influx_diffuse, 0, influx_toa - direct)  # :208
        k = cosincidence / np.sin(alt)  # :215
        cos_surface_slope = np.cos(slope)  # :217
        influx = direct + diffuse  # :221
        direct_t = k * direct  # :222
        diffuse_t = (1.0 + cos_surface_slope) / 2.0 * diffuse  # :223
        ground_t = albedo * influx * ((1.0 - cos_surface_slope) / 2.0)  # :224
        total_t = _fillna0(direct_t) + _fillna0(diffuse_t) + _fillna0(ground_t)  # :226
        cap_alt = alt < np.radians(altitude_threshold)  # :251
        result = np.where(~(cap_alt | (direct + diffuse <= 0.01)), total_t, 0)  # :252
    return result


def power_huld(irradiance, t_amb, pc):
    """_power_huld, atlite/pv/solar_panel_model.py:22-41."""
    with np.errstate(divide="ignore", invalid="ignore"):
        T_ = (pc["c_temp_amb"] * t_amb + pc["c_temp_irrad"] * irradiance) - pc["r_tmod"]  # :23
        G_ = irradiance / pc["r_irradiance"]  # :26
        log_G_ = np.log(np.where(G_ > 0, G_, np.nan))  # :28
        eff = (
            1
            + pc["k_1"] * log_G_
            + pc["k_2"] * (log_G_) ** 2
            + T_ * (pc["k_3"] + pc["k_4"] * log_G_ + pc["k_5"] * log_G_**2)
            + pc["k_6"] * (T_**2)
        )  # :30-36
        eff = np.clip(_fillna0(eff), 0, None)  # :38
        da = G_ * eff * pc.get("inverter_efficiency", 1.0)  # :40
    return da


def convert_pv(ds, panel, orientation, altitude_threshold=1.0):
    """
    convert_pv, atlite/convert.py:840-854 with tracking=None, trigon_model="simple", for a
    dataset holding solar_altitude / solar_azimuth (getter branch, solar_position.py:54-60).
    ds: dict of arrays broadcastable to a common shape; orientation: dict(slope, azimuth) in
    radians, broadcastable.
    """
    alt, az = ds["solar_altitude"], ds["solar_azimuth"]
    slope, azimuth = orientation["slope"], orientation["azimuth"]
    cosinc = surface_orientation(alt, az, slope, azimuth)
    irr = tilted_irradiation(
        ds["influx_direct"], ds["influx_diffuse"], ds["influx_toa"], ds["albedo"], alt, cosinc, slope,
        altitude_threshold,
    )
    return power_huld(irr, ds["temperature"], panel)


# --------------------------------------------------------------------------------------
# wind
# --------------------------------------------------------------------------------------


def extrapolate_wind_speed(wnd, aux, to_height, from_height, method="logarithmic"):
    """extrapolate_wind_speed, atlite/wind.py:91-112 (method None = fast lane :76-78)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        if method is None:
            return wnd
        if method == "logarithmic":
            return wnd * (np.log(to_height / aux) / np.log(from_height / aux))  # :99-101
        if method == "power":
            return wnd * (to_height / from_height) ** aux  # :111
    raise ValueError(f"Interpolation method must be 'logarithmic' or 'power',  but is: {method}")


def convert_wind(wnd, aux, V, POW, P, to_height, from_height, method="logarithmic"):
    """convert_wind, atlite/convert.py:634-662: np.interp(wnd_hub, V, POW / P)."""
    wnd_hub = extrapolate_wind_speed(wnd, aux, to_height, from_height, method)
    return np.interp(wnd_hub, V, np.asarray(POW) / P)  # :648-649


# --------------------------------------------------------------------------------------
# heat demand / runoff
# --------------------------------------------------------------------------------------


def day_groups(time, hour_shift=0.0):
    """
    Offsets of the calendar-day bins xarray's ``resample(time="1D")`` makes on the shifted
    time axis (atlite/convert.py:408-412).  Returns (day_ptr (D+1,), day_labels (D,)).
    Bins between the first and last day that hold no sample are kept (empty -> NaN mean).
    """
    import datetime as dt

    t = pd.DatetimeIndex(time) + pd.Timedelta(np.timedelta64(dt.timedelta(hours=hour_shift)))
    if len(t) == 0:
        return np.zeros(1, dtype=np.int64), pd.DatetimeIndex([])
    days = t.floor("D")
    labels = pd.date_range(days[0], days[-1], freq="D")
    # sorted time axis: searchsorted gives the contiguous ranges
    ptr = np.searchsorted(days.values, np.append(labels.values, labels.values[-1] + np.timedelta64(1, "D")))
    return ptr.astype(np.int64), labels


def convert_heat_demand(temperature, day_ptr, threshold=15.0, a=1.0, constant=0.0):
    """
    convert_heat_demand, atlite/convert.py:405-418.  temperature (T, ...) in K; groups given by
    day_ptr.  nan-skipping mean per bin (xarray resample().mean(), skipna for floats).
    """
    D = len(day_ptr) - 1
    out_shape = (D,) + temperature.shape[1:]
    Tm = np.full(out_shape, np.nan)
    with np.errstate(invalid="ignore", divide="ignore"):
        for d in range(D):
            blk = temperature[day_ptr[d] : day_ptr[d + 1]]
            if blk.shape[0]:
                cnt = np.sum(~np.isnan(blk), axis=0)
                Tm[d] = np.nansum(blk, axis=0) / cnt
        threshold = threshold + 273.15  # :413
        heat_demand = a * (threshold - Tm)  # :414
        heat_demand = np.clip(heat_demand, 0.0, None)  # :416
        return constant + heat_demand  # :418


def convert_runoff(runoff, height=None):
    """convert_runoff, atlite/convert.py:1028-1034; height broadcastable (static y,x)."""
    return runoff * height if height is not None else runoff


# --------------------------------------------------------------------------------------
# aggregation + gateway algebra
# --------------------------------------------------------------------------------------


def aggregate_matrix(da, matrix, dask_branch=False):
    """
    aggregate_matrix, atlite/aggregate.py:16-35.  da (T, S) already stacked y-major.
    numpy branch (:33-35): ``matrix * da.T`` -> (N, T).  dask branch (:21-32):
    ``chunk * matrix.T`` per time chunk -> (T, N).
    """
    matrix = sp.csr_matrix(matrix)
    if dask_branch:
        return da * matrix.T
    return matrix * np.ascontiguousarray(da.T)


def aggregate_time(a, method, axis):
    """_aggregate_time, atlite/convert.py:51-56 (xarray sum/mean skip NaN for floats)."""
    with np.errstate(invalid="ignore", divide="ignore"):
        if method == "sum":
            return np.nansum(a, axis=axis)
        if method == "mean":
            cnt = np.sum(~np.isnan(a), axis=axis)
            return np.nansum(a, axis=axis) / cnt
    return a


def gateway(da, matrix=None, layout=None, per_unit=False, aggregate_time_method=None):
    """
    Numerical core of convert_and_aggregate, atlite/convert.py:200-271, for an already
    converted cube ``da`` (T, S): matrix / layout handling (:233,:242-249), aggregation (:257),
    capacity (:259-262), per-unit (:264-266), time aggregation (:270-271).
    Returns (result, capacity_or_None); result is (N, T) like the reference's numpy branch.
    """
    if matrix is None and layout is None:
        return aggregate_time(da, aggregate_time_method, axis=0), None
    if matrix is not None:
        matrix = sp.csr_matrix(matrix)
    if layout is not None:
        lay = np.asarray(layout, dtype=np.float64).ravel()
        if matrix is None:
            matrix = sp.csr_matrix(lay[None, :])  # :246-247
        else:
            N = len(lay)
            inds = np.arange(N + 1, dtype=np.int32)
            matrix = sp.csr_matrix(matrix) * sp.csr_matrix((lay, inds[:-1], inds), (N, N))  # :249, gis.py:78-84
    res = aggregate_matrix(da, matrix)
    capacity = np.asarray(matrix.sum(-1)).flatten()  # :260-261
    if per_unit:
        with np.errstate(invalid="ignore", divide="ignore"):
            cap = np.where(capacity != 0, capacity, np.nan)[:, None]
            res = _fillna0(res / cap)  # :265
    res = aggregate_time(res, aggregate_time_method, axis=1)
    return res, capacity


# --------------------------------------------------------------------------------------
# remaining pv options: tracking, Hay-Davies, Reindl split, albedo from
# outflux, bofinger panel, irradiation quantities, solar thermal
# --------------------------------------------------------------------------------------


def surface_orientation_tracking(alt, az, slope, azimuth, tracking=None):
    """
    SurfaceOrientation, atlite/pv/orientation.py:104-196, all tracking modes.
    Returns dict(cosincidence, slope, azimuth); slope/azimuth broadcast against alt/az.
    """
    pi = np.pi
    surface_slope, surface_azimuth = slope, azimuth
    sun_altitude, sun_azimuth = alt, az
    sin, cos = np.sin, np.cos
    with np.errstate(all="ignore"):
        if tracking is None:
            cosincidence = sin(surface_slope) * cos(sun_altitude) * cos(surface_azimuth - sun_azimuth) + cos(
                surface_slope
            ) * sin(sun_altitude)  # :114-117
        elif tracking == "horizontal":  # :119-131
            axis_azimuth = azimuth
            rotation = np.arctan((cos(sun_altitude) / sin(sun_altitude)) * sin(sun_azimuth - axis_azimuth))
            surface_slope = abs(rotation)
            surface_azimuth = axis_azimuth + np.arcsin(sin(rotation) / sin(surface_slope))
            cosincidence = cos(surface_slope) * sin(sun_altitude) + sin(surface_slope) * cos(sun_altitude) * cos(
                sun_azimuth - surface_azimuth
            )
        elif tracking == "tilted_horizontal":  # :133-168
            axis_tilt = slope
            rotation = np.arctan(
                (cos(sun_altitude) * sin(sun_azimuth - surface_azimuth))
                / (cos(sun_altitude) * cos(sun_azimuth - surface_azimuth) * sin(axis_tilt) + sin(sun_altitude) * cos(axis_tilt))
            )
            surface_slope = np.arccos(cos(rotation) * cos(axis_tilt))
            azimuth_difference = sun_azimuth - surface_azimuth
            azimuth_difference = np.where(azimuth_difference > pi, azimuth_difference - 2 * pi, azimuth_difference)
            azimuth_difference = np.where(azimuth_difference < -pi, 2 * pi + azimuth_difference, azimuth_difference)
            rotation = np.where(np.logical_and(rotation < 0, azimuth_difference > 0), rotation + pi, rotation)
            rotation = np.where(np.logical_and(rotation > 0, azimuth_difference < 0), rotation - pi, rotation)
            cosincidence = cos(rotation) * (
                sin(axis_tilt) * cos(sun_altitude) * cos(sun_azimuth - surface_azimuth) + cos(axis_tilt) * sin(sun_altitude)
            ) + sin(rotation) * cos(sun_altitude) * sin(sun_azimuth - surface_azimuth)
        elif tracking == "vertical":  # :170-173
            cosincidence = sin(surface_slope) * cos(sun_altitude) + cos(surface_slope) * sin(sun_altitude)
        elif tracking == "dual":  # :174-175
            cosincidence = np.float64(1.0)
        else:
            raise AssertionError("bad tracking")
        cosincidence = np.clip(cosincidence, 0, None)  # :188
    return dict(cosincidence=cosincidence, slope=surface_slope, azimuth=surface_azimuth)


def diffuse_horizontal_irrad(influx, influx_toa, alt, clearsky_model, temperature=None, humidity=None):
    """DiffuseHorizontalIrrad (Reindl 1990), atlite/pv/irradiation.py:13-73."""
    sinaltitude = np.sin(alt)
    with np.errstate(all="ignore"):
        k = influx / influx_toa  # :29
        if clearsky_model == "simple":  # :36-41
            fraction = (
                ((k > 0.0) & (k <= 0.3)) * np.fmin(1.0, 1.020 - 0.254 * k + 0.0123 * sinaltitude)
                + ((k > 0.3) & (k < 0.78)) * np.fmin(0.97, np.fmax(0.1, 1.400 - 1.749 * k + 0.177 * sinaltitude))
                + (k >= 0.78) * np.fmax(0.1, 0.486 * k - 0.182 * sinaltitude)
            )
        elif clearsky_model == "enhanced":  # :48-63
            T, rh = temperature, humidity
            fraction = (
                ((k > 0.0) & (k <= 0.3)) * np.fmin(1.0, 1.000 - 0.232 * k + 0.0239 * sinaltitude - 0.000682 * T + 0.0195 * rh)
                + ((k > 0.3) & (k < 0.78))
                * np.fmin(0.97, np.fmax(0.1, 1.329 - 1.716 * k + 0.267 * sinaltitude - 0.00357 * T + 0.106 * rh))
                + (k >= 0.78) * np.fmax(0.1, 0.426 * k - 0.256 * sinaltitude + 0.00349 * T + 0.0734 * rh)
            )
        else:
            raise KeyError("`clearsky model` must be chosen from 'simple' and 'enhanced'")
        return influx * fraction  # :73


def _albedo(ds, influx):
    """irradiation.py:128-139."""
    if "albedo" in ds:
        return ds["albedo"]
    with np.errstate(all="ignore"):
        a = ds["outflux"] / np.where(influx != 0, influx, np.nan)
        return np.clip(_fillna0(a), None, 1)


def tilted_irradiation_general(ds, alt, so, trigon_model="simple", clearsky_model="simple", tracking=None,
                               altitude_threshold=1.0, irradiation="total"):
    """TiltedIrradiation, atlite/pv/irradiation.py:196-255, every branch."""
    influx_toa = ds["influx_toa"]
    cosincidence, surface_slope = so["cosincidence"], so["slope"]
    with np.errstate(all="ignore"):
        if "influx" in ds:  # :202-205
            influx = np.clip(ds["influx"], 0, influx_toa)
            if clearsky_model is None:
                clearsky_model = "enhanced" if ("temperature" in ds and "humidity" in ds) else "simple"
            diffuse = diffuse_horizontal_irrad(influx, influx_toa, alt, clearsky_model, ds.get("temperature"),
                                               ds.get("humidity"))
            direct = influx - diffuse
        else:  # :206-208
            direct = np.clip(ds["influx_direct"], 0, influx_toa)
            diffuse = np.clip(ds["influx_diffuse"], 0, influx_toa - direct)
        sinalt = np.sin(alt)
        if trigon_model == "simple":  # :214-226
            k = cosincidence / sinalt
            cos_surface_slope = np.cos(surface_slope) if tracking != "dual" else sinalt
            influx = direct + diffuse
            direct_t = k * direct
            diffuse_t = (1.0 + cos_surface_slope) / 2.0 * diffuse
            ground_t = _albedo(ds, influx) * influx * ((1.0 - cos_surface_slope) / 2.0)
            total_t = _fillna0(direct_t) + _fillna0(diffuse_t) + _fillna0(ground_t)
        else:  # :227-236  Hay-Davies (:76-115), direct (:118-125), ground (:142-145)
            influx = direct + diffuse
            f = _fillna0(np.sqrt(direct / influx))
            A = direct / influx_toa
            R_b = cosincidence / sinalt
            # sin(slope / 2) ** 3 on a labelled array is numpy's ARRAY power even for one orientation for the whole grid
            # (a 0-d variable); numpy's scalar power of an np.float64 can differ from it in the last bit
            s3 = np.power(np.atleast_1d(np.sin(surface_slope / 2.0)), 3).reshape(np.shape(surface_slope))
            diffuse_t = ((1.0 - A) * ((1 + np.cos(surface_slope)) / 2.0) * (1.0 + f * s3) + A * R_b) * diffuse
            diffuse_t = _fillna0(np.clip(diffuse_t, 0, None))
            direct_t = R_b * direct
            ground_t = influx * _albedo(ds, influx) * (1.0 - np.cos(surface_slope)) / 2.0
            total_t = direct_t + diffuse_t + ground_t
        result = dict(total=total_t, direct=direct_t, diffuse=diffuse_t, ground=ground_t)[irradiation]  # :238-245
        cap_alt = alt < np.radians(altitude_threshold)
        shape = np.broadcast_shapes(np.shape(result), np.shape(cap_alt), np.shape(direct))
        result = np.where(~(cap_alt | (direct + diffuse <= 0.01)), np.broadcast_to(result, shape), 0)  # :251-252
    return result


def power_bofinger(irradiance, t_amb, pc):
    """_power_bofinger, atlite/pv/solar_panel_model.py:47-74."""
    with np.errstate(all="ignore"):
        fraction = (pc["NOCT"] - pc["Tamb"]) / pc["Intc"]
        eta_ref = pc["A"] + pc["B"] * irradiance + pc["C"] * np.log(np.where(irradiance != 0, irradiance, np.nan))
        eta = _fillna0(
            eta_ref * (1.0 + pc["D"] * (fraction * irradiance + (t_amb - pc["Tstd"])))
            / (1.0 + pc["D"] * fraction / pc["ta"] * eta_ref * irradiance)
        )
        capacity = (pc["A"] + pc["B"] * 1000.0 + pc["C"] * np.log(1000.0)) * 1e3
        power = irradiance * eta * (pc.get("inverter_efficiency", 1.0) / capacity)
        return np.where(irradiance >= pc["threshold"], power, 0)


def convert_pv_general(ds, panel, orientation, tracking=None, trigon_model="simple", clearsky_model="simple",
                       altitude_threshold=1.0):
    """convert_pv, atlite/convert.py:840-854, any option (stored solar angles)."""
    alt, az = ds["solar_altitude"], ds["solar_azimuth"]
    so = surface_orientation_tracking(alt, az, orientation["slope"], orientation["azimuth"], tracking)
    irr = tilted_irradiation_general(ds, alt, so, trigon_model, clearsky_model, tracking, altitude_threshold)
    if panel.get("model", "huld") == "huld":
        return power_huld(irr, ds["temperature"], panel)
    return power_bofinger(irr, ds["temperature"], panel)


def convert_irradiation(ds, orientation, tracking=None, irradiation="total", trigon_model="simple",
                        clearsky_model="simple"):
    """convert_irradiation, atlite/convert.py:748-767."""
    alt, az = ds["solar_altitude"], ds["solar_azimuth"]
    so = surface_orientation_tracking(alt, az, orientation["slope"], orientation["azimuth"], tracking)
    return tilted_irradiation_general(ds, alt, so, trigon_model, clearsky_model, tracking, irradiation=irradiation)


def convert_solar_thermal(ds, orientation, trigon_model="simple", clearsky_model="simple", c0=0.8, c1=3.0,
                          t_store=80.0):
    """convert_solar_thermal, atlite/convert.py:550-574."""
    t_store = t_store + 273.15
    alt, az = ds["solar_altitude"], ds["solar_azimuth"]
    so = surface_orientation_tracking(alt, az, orientation["slope"], orientation["azimuth"], None)
    irr = tilted_irradiation_general(ds, alt, so, trigon_model, clearsky_model, tracking=0)
    with np.errstate(all="ignore"):
        eta = c0 - c1 * _fillna0((t_store - ds["temperature"]) / np.where(irr != 0, irr, np.nan))
        output = irr * eta
        return np.where(output > 0.0, output, 0.0)


# --------------------------------------------------------------------------------------
# temperatures, COP, cooling demand
# --------------------------------------------------------------------------------------


def convert_temperature(T):
    """convert_temperature / convert_dewpoint_temperature, atlite/convert.py:292-299, 326-330."""
    return T - 273.15


def convert_soil_temperature(T):
    """convert_soil_temperature, atlite/convert.py:307-318."""
    return _fillna0(T - 273.15)


def convert_coefficient_of_performance(T, source="air", sink_T=55.0, c0=None, c1=None, c2=None):
    """convert_coefficient_of_performance, atlite/convert.py:338-364 (T = air or soil temperature, K)."""
    if source == "air":
        source_T = convert_temperature(T)
        d = (6.81, -0.121, 0.000630)
    else:
        source_T = convert_soil_temperature(T)
        d = (8.77, -0.150, 0.000734)
    c0, c1, c2 = (d[i] if v is None else v for i, v in enumerate((c0, c1, c2)))
    delta_T = sink_T - source_T
    return c0 + c1 * delta_T + c2 * delta_T**2


def convert_cooling_demand(temperature, day_ptr, threshold=23.0, a=1.0, constant=0.0):
    """convert_cooling_demand, atlite/convert.py:475-490."""
    D = len(day_ptr) - 1
    Tm = np.full((D,) + temperature.shape[1:], np.nan)
    with np.errstate(invalid="ignore", divide="ignore"):
        for d in range(D):
            blk = temperature[day_ptr[d] : day_ptr[d + 1]]
            if blk.shape[0]:
                Tm[d] = np.nansum(blk, axis=0) / np.sum(~np.isnan(blk), axis=0)
        return constant + np.clip(a * (Tm - (threshold + 273.15)), 0.0, None)


def runoff_postprocess(series, time, rows, smooth=None, lower_threshold_quantile=None, normalize_using_yearly=None):
    """convert.py:1046-1082 on the aggregated (rows x time) runoff series (NumPy array), pandas doing what xarray does there:
    rolling mean over `smooth` steps with min_periods=1 (:1046-1052), values below the quantile of ALL values -> 0
    (:1054-1062), scaling to the reported totals of the full years (> 8700 steps) the series shares with
    `normalize_using_yearly` (a DataFrame: years x rows) (:1064-1082).  `rows`: labels of the series' rows."""
    import pandas as pd

    vals = np.asarray(series, dtype=np.float64)
    if smooth is not None:
        if smooth is True:
            smooth = 24 * 7  # :1047-1048
        vals = pd.DataFrame(vals.T).rolling(smooth, min_periods=1).mean().values.T  # :1052
    if lower_threshold_quantile is not None:
        if lower_threshold_quantile is True:
            lower_threshold_quantile = 5e-3  # :1055-1056
        thr = pd.Series(vals.ravel()).quantile(lower_threshold_quantile)  # :1057-1059
        vals = np.where(vals >= thr, vals, 0.0)  # :1060
    if normalize_using_yearly is not None:
        nidx = normalize_using_yearly.index
        nidx = nidx.year if isinstance(nidx, pd.DatetimeIndex) else nidx.astype(int)  # :1063-1067
        tyear = pd.Series(pd.to_datetime(time).year)
        years = tyear.value_counts().loc[lambda x: x > 8700].index.intersection(nidx)  # :1069-1074
        assert len(years), "Need at least a full year of data (more is better)"
        lo, hi = min(years), max(years)
        tmask = ((tyear >= lo) & (tyear <= hi)).values  # sel(time=slice(str(min), str(max)))
        ref = normalize_using_yearly.copy()
        ref.index = nidx
        ref = ref.loc[lo:hi].sum().reindex(pd.Index(rows)).values  # :1079-1081 (+ reindex to the result's rows)
        with np.errstate(invalid="ignore", divide="ignore"):
            vals = vals * (ref / np.nansum(vals[:, tmask], axis=1))[:, None]
    return vals

