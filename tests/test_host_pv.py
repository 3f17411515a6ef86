"""
The pv kernels' per-cell routines on the HOST (atl_pv_probe_host: the same source as the kernels, host
build) against the oracle, on random points of the option space with hostile values - the CPU-side twin
of tests/fuzz_pv_options.py.  Both families are exercised for every option combination they implement:
the fast family (Huld / Hay-Davies / bofinger / solar thermal / irradiation tails, closed-form trackers with
either trigon model) and the
general kernel's routine (every tracker x trigon model x panel x dataset flavour).
"""
import ctypes as C

import numpy as np
import pytest

from atlite_amd import _lib
from atlite_amd._lib import check
from atlite_amd.resource import get_solarpanelconfig
from oracle import atlite_oracle as orc

TRACK = [None, "horizontal", "tilted_horizontal", "vertical", "dual"]
NAMES = ["influx_direct", "influx_diffuse", "influx", "influx_toa", "albedo", "outflux", "temperature", "humidity",
         "solar_altitude", "solar_azimuth"]


def params(panel, trk, tm, cs, what, irr="total"):
    pp = _lib.PvParams()
    model = "none" if what == "irradiation" else "solar_thermal" if what == "thermal" else panel.get("model", "huld")
    pp.panel_model = _lib.PANEL[model]
    if model == "huld":
        for k in ("c_temp_amb", "c_temp_irrad", "r_tmod", "r_irradiance", "k_1", "k_2", "k_3", "k_4", "k_5", "k_6"):
            setattr(pp, k, float(panel[k]))
    else:
        pp.r_irradiance = 1.0
    if model == "bofinger":
        for k in ("A", "B", "C", "D", "NOCT", "Tstd", "Tamb", "Intc", "ta", "threshold"):
            setattr(pp, "bof_" + k, float(panel[k]))
    if model == "solar_thermal":
        pp.st_c0, pp.st_c1, pp.st_t_store_K = 0.8, 3.0, 80.0 + 273.15
    pp.inverter_efficiency = float(panel.get("inverter_efficiency", 1.0)) if what == "pv" else 1.0
    pp.altitude_threshold = float(np.radians(1.0))
    pp.tracking = _lib.TRACKING[trk]
    pp.trigon_model = _lib.TRIGON[tm]
    pp.clearsky_model = _lib.CLEARSKY[cs]
    pp.irradiation = _lib.IRRADIATION[irr]
    return pp


def probe(pp, family, ds, slope, azim):
    n = ds["influx_toa"].size
    arrs = [np.ascontiguousarray(ds[k], dtype=np.float64).ravel() if k in ds else None for k in NAMES]
    arrs += [np.ascontiguousarray(np.broadcast_to(slope, ds["influx_toa"].shape), dtype=np.float64).ravel(),
             np.ascontiguousarray(np.broadcast_to(azim, ds["influx_toa"].shape), dtype=np.float64).ravel()]
    ptrs = (C.c_void_p * 12)(*[a.ctypes.data if a is not None else None for a in arrs])
    out = np.empty(n)
    check(_lib.load().atl_pv_probe_host(C.byref(pp), family, n, ptrs, out.ctypes.data))
    return out.reshape(ds["influx_toa"].shape)


def hostile_dataset(rng, shape):
    n = int(np.prod(shape))
    alt = (rng.random(shape) - 0.35) * 1.6
    toa = 1361.0 * np.maximum(np.sin(alt), 0.0)
    kt, fd = 0.2 + 0.55 * rng.random(shape), 0.3 + 0.5 * rng.random(shape)
    ds = dict(influx_toa=toa, influx_direct=toa * kt * fd, influx_diffuse=toa * kt * (1 - fd), albedo=0.05 + 0.3 * rng.random(shape),
              temperature=268.0 + 30.0 * rng.random(shape), solar_altitude=alt, solar_azimuth=2 * np.pi * rng.random(shape))
    for k in ("influx_direct", "influx_diffuse", "temperature", "albedo"):
        ds[k][rng.random(shape) < 0.03] = rng.choice([np.nan, 0.0, -5.0, 1e4])
    ds["solar_altitude"][rng.random(shape) < 0.03] = rng.choice([0.0, np.radians(1.0), np.pi / 2, -0.3, np.nan])
    ds["solar_azimuth"][rng.random(shape) < 0.03] = np.pi
    ds["influx_toa"][rng.random(shape) < 0.02] = 0.0
    assert n
    return ds


def allowance_error(got, ref):
    scale = np.nanmax(np.abs(ref[np.isfinite(ref)])) if np.isfinite(ref).any() else 1.0
    with np.errstate(all="ignore"):
        err = np.abs(got - ref) / (1e-10 * np.abs(ref) + 1e-12 * max(scale, 1e-300))
    same = (got == ref) | (np.isnan(got) & np.isnan(ref))
    err = np.where(same, 0.0, err)
    return float(np.where(np.isnan(err), np.inf, err).max())


@pytest.mark.parametrize("seed", range(3))
def test_host_pv_math_against_oracle(seed):
    rng = np.random.default_rng(700 + seed)
    worst = 0.0
    for case in range(120):
        shape = (int(rng.integers(3, 40)), int(rng.integers(1, 6)), int(rng.integers(1, 9)))
        ds = hostile_dataset(rng, shape)
        flavour = str(rng.choice(["split", "split", "influx", "outflux", "sarah"]))
        if flavour in ("outflux", "sarah"):  # "sarah": total influx and outflux (the fast family's influx head)
            ds["outflux"] = (ds["influx_direct"] + ds["influx_diffuse"]) * ds["albedo"]
            ds["outflux"][rng.random(shape) < 0.03] = rng.choice([np.nan, 0.0, -5.0, 1e4])
            del ds["albedo"]
        if flavour in ("influx", "sarah"):
            ds["influx"] = ds["influx_direct"] + ds["influx_diffuse"]
            ds["humidity"] = rng.random(shape)
            del ds["influx_direct"], ds["influx_diffuse"]
        trk = TRACK[int(rng.integers(5))]
        tm, cs = str(rng.choice(["simple", "other"])), str(rng.choice(["simple", "enhanced"]))
        what = str(rng.choice(["pv", "pv", "irradiation", "thermal"]))
        if what == "thermal":
            trk = None  # convert_solar_thermal has no tracking argument
        pname = str(rng.choice(["CSi", "CdTe", "KANENA"]))
        panel = get_solarpanelconfig(pname)
        irr = str(rng.choice(["total", "direct", "diffuse", "ground"])) if what == "irradiation" else "total"
        if rng.random() < 0.5:
            sl, az = float(rng.choice([0.0, 30.0, 90.0, rng.random() * 90])), float(rng.choice([180.0, 0.0, rng.random() * 360]))
            ori = orc.orientation_constant(sl, az)
        else:
            lat = np.radians(30 + 40 * rng.random(shape[1]))
            o = orc.orientation_latitude_optimal(lat)
            ori = dict(slope=o["slope"][:, None] * np.ones(shape[1:]), azimuth=o["azimuth"][:, None] * np.ones(shape[1:]))
        with np.errstate(all="ignore"):
            if what == "pv":
                ref = orc.convert_pv_general(ds, panel, ori, trk, tm, cs)
            elif what == "irradiation":
                ref = orc.convert_irradiation(ds, ori, trk, irr, tm, cs)
            else:
                ref = orc.convert_solar_thermal(ds, ori, tm, cs, 0.8, 3.0, 80.0)
        ref = np.broadcast_to(ref, shape)
        pp = params(panel, trk, tm, cs, what, irr)
        got = probe(pp, 1, ds, ori["slope"], ori["azimuth"])  # the general routine covers everything
        e = allowance_error(got, ref)
        assert e <= 1.0, ("general", what, trk, tm, cs, pname, flavour, e)
        worst = max(worst, e)
        # the fast family wherever the dispatcher would use it
        model = pp.panel_model
        fast_ok = flavour == "split"  # (solar_thermal has no tracker in the reference's API either)
        if flavour in ("sarah", "influx"):  # total influx with outflux or an albedo variable: the Huld panel on a fixed mount, either trigon / clearsky model
            fast_ok = what == "pv" and trk is None and model == _lib.PANEL["huld"]
        if fast_ok:
            got = probe(pp, 0, ds, ori["slope"], ori["azimuth"])
            e = allowance_error(got, ref)
            assert e <= 1.0, ("fast", what, trk, tm, cs, pname, flavour, e)
            worst = max(worst, e)
    assert worst < 0.05  # far inside the allowance, like on the device
