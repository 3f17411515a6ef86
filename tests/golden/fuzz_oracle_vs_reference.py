#!/usr/bin/env python3
"""
Randomised differential test ORACLE vs the REFERENCE'S OWN CODE (executed under the xarray/dask
stand-in of refshim.py) - the link that pins the oracle beyond the fixed golden vectors.  Runs in the
build container only (it needs /root/reference); the GPU-vs-oracle fuzzers (tests/fuzz_*.py) run on
the GPU box.  Together: GPU == oracle == reference on random points of the option space with hostile
values (NaN, zero, negative, clipped radiation; sun on the horizon / at the zenith / in the panel
azimuth; NaN, zero and negative roughness; NaN temperatures).

    python tests/golden/fuzz_oracle_vs_reference.py [n_cases] [seed]
"""
from __future__ import annotations

import sys
import warnings
from pathlib import Path

import numpy as np
import pandas as pd

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent.parent))
import refshim  # noqa: E402

refshim.install()
import xarray as xr  # noqa: E402  (the stand-in)

from oracle import atlite_oracle as orc  # noqa: E402
from tests import helpers as H  # noqa: E402

conv = refshim.reference("atlite.convert")
res = refshim.reference("atlite.resource")
orient = refshim.reference("atlite.pv.orientation")

TRACK = [None, "horizontal", "tilted_horizontal", "vertical", "dual"]


class MockCutout:  # as in the reference's test/test_aggregate_time.py:13-17
    def __init__(self, data):
        self.data = data
        grid_coords = np.array([(xx, yy) for yy in data["y"].values for xx in data["x"].values])
        self.grid = pd.DataFrame(grid_coords, columns=["x", "y"])


MockCutout.convert_and_aggregate = conv.convert_and_aggregate


def dataset(variables, time, x, y):
    coords = {"time": time, "y": y, "x": x}
    return xr.Dataset(
        {k: xr.DataArray(v, dims=["time", "y", "x"][-v.ndim:], coords={d: coords[d] for d in ["time", "y", "x"][-v.ndim:]})
         for k, v in variables.items()},
        coords={"time": time, "y": y, "x": x, "lon": ("x", x), "lat": ("y", y)},
    )


def compare(got, ref, what, stats):
    got, ref = np.asarray(got, float), np.asarray(ref, float)
    ref = np.broadcast_to(ref, got.shape)
    scale = np.nanmax(np.abs(ref[np.isfinite(ref)])) if np.isfinite(ref).any() else 1.0
    with np.errstate(all="ignore"):
        err = np.abs(got - ref) / (1e-10 * np.abs(ref) + 1e-12 * max(scale, 1e-300))
    same = (got == ref) | (np.isnan(got) & np.isnan(ref))
    err = np.where(same, 0.0, err)
    err = np.where(np.isnan(err), np.inf, err)
    e = float(err.max()) if err.size else 0.0
    if np.isfinite(e) and e > stats["worst"]:
        stats["worst"], stats["worst_what"] = e, what
    if e > 1.0:
        i = np.unravel_index(np.argmax(err), err.shape)
        print(f"MISMATCH {what}: {e:.3e} of the allowance at {i}: oracle {got[i]!r} reference {ref[i]!r}")
        stats["fails"] += 1


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    stats = dict(worst=0.0, fails=0)
    turbines = ["Vestas_V112_3MW", "Enercon_E101_3000kW", "NREL_ReferenceTurbine_5MW_offshore", "Vestas_V90_3MW",
                "Siemens_SWT_2300kW", "Bonus_B1000_1000kW"]
    for case in range(n):
        T, Y, X = int(rng.integers(3, 30)), int(rng.integers(1, 7)), int(rng.integers(2, 9))
        x, y = H.grid(Y, X) if Y > 1 else (H.grid(2, X)[0], np.array([47.0]))
        time = H.times(T, str(rng.choice(["2013-01-01", "2013-06-20 03:00", "2012-12-31 22:00"])))
        fam = str(rng.choice(["pv", "pv", "irradiation", "thermal", "wind", "heat", "runoff"]))
        with warnings.catch_warnings(), np.errstate(all="ignore"):
            warnings.simplefilter("ignore")
            if fam in ("pv", "irradiation", "thermal"):
                ds = H.pv_dataset(T, Y, X, seed=int(rng.integers(1 << 30)))
                ds = {k: v.reshape(T, Y, X).copy() for k, v in ds.items()}
                for k in ("influx_direct", "influx_diffuse", "temperature", "albedo"):
                    ds[k][rng.random((T, Y, X)) < 0.03] = rng.choice([np.nan, 0.0, -5.0, 1e4])
                ds["solar_altitude"][rng.random((T, Y, X)) < 0.03] = rng.choice([0.0, np.radians(1.0), np.pi / 2, -0.3, np.nan])
                ds["solar_azimuth"][rng.random((T, Y, X)) < 0.03] = np.pi
                ds["influx_toa"][rng.random((T, Y, X)) < 0.02] = 0.0
                flavour = str(rng.choice(["split", "influx", "outflux"]))
                if flavour == "influx":
                    ds["influx"] = ds["influx_direct"] + ds["influx_diffuse"]
                    ds["humidity"] = rng.random((T, Y, X))
                    del ds["influx_direct"], ds["influx_diffuse"]
                if flavour == "outflux":
                    ds["outflux"] = (ds["influx_direct"] + ds["influx_diffuse"]) * ds["albedo"]
                    del ds["albedo"]
                trk = TRACK[int(rng.integers(5))]
                tm, cs = str(rng.choice(["simple", "other"])), str(rng.choice(["simple", "enhanced"]))
                okind = str(rng.choice(["const", "latitude_optimal", "latitude"]))
                if trk == "tilted_horizontal":
                    # the reference passes DataArrays through np.where there (orientation.py:149-163), which
                    # broadcasts POSITIONALLY: with a per-latitude orientation the operands' dim orders differ
                    # and real xarray fails (or mis-broadcasts when Y == T) just like the stand-in
                    okind = "const"
                if okind == "const":
                    sl = float(rng.choice([0.0, 30.0, 90.0, rng.random() * 90]))
                    az = float(rng.choice([180.0, 0.0, rng.random() * 360]))
                    ospec, ori = {"slope": sl, "azimuth": az}, orc.orientation_constant(sl, az)
                else:
                    ospec = okind
                    lat = np.radians(y)
                    o = orc.orientation_latitude_optimal(lat) if okind == "latitude_optimal" else orc.orientation_latitude(lat)
                    ori = dict(slope=np.asarray(o["slope"], float).reshape(-1, 1) * np.ones((1, X)),
                               azimuth=(np.asarray(o["azimuth"], float).reshape(-1, 1) if np.ndim(o["azimuth"]) else
                                        np.asarray(o["azimuth"], float)) * np.ones((Y, X)))
                xds = dataset(ds, time, x, y)
                oref = orient.get_orientation(dict(ospec) if isinstance(ospec, dict) else ospec)
                if fam == "pv":
                    panel = str(rng.choice(["CSi", "CdTe", "KANENA"]))
                    pc = res.get_solarpanelconfig(panel)
                    ref = conv.convert_pv(xds, pc, oref, tracking=trk, trigon_model=tm, clearsky_model=cs).transpose("time", "y", "x").values
                    got = orc.convert_pv_general(ds, pc, ori, trk, tm, cs)
                    what = f"pv {panel} trk={trk} {tm} {cs} ori={okind} {flavour}"
                elif fam == "irradiation":
                    q = str(rng.choice(["total", "direct", "diffuse", "ground"]))
                    ref = conv.convert_irradiation(xds, oref, tracking=trk, irradiation=q, trigon_model=tm,
                                                   clearsky_model=cs).transpose("time", "y", "x").values
                    got = orc.convert_irradiation(ds, ori, trk, q, tm, cs)
                    what = f"irradiation {q} trk={trk} {tm} {cs} ori={okind} {flavour}"
                else:
                    ref = conv.convert_solar_thermal(xds, oref, tm, cs, 0.8, 3.0, 80.0).transpose("time", "y", "x").values
                    got = orc.convert_solar_thermal(ds, ori, tm, cs, 0.8, 3.0, 80.0)
                    what = f"thermal {tm} {cs} ori={okind} {flavour}"
            elif fam == "wind":
                v = 12 * rng.random((T, Y, X)) ** 1.5
                z0 = np.exp(np.log(1e-3) + rng.random((T, Y, X)) * np.log(2e3))
                v[rng.random((T, Y, X)) < 0.03] = rng.choice([np.nan, 0.0, 25.0, 13.0, 1e3, np.inf, -1.0])
                z0[rng.random((T, Y, X)) < 0.03] = rng.choice([0.0, -1.0, np.nan, 100.0, np.inf])
                name = str(rng.choice(turbines))
                tb = res.get_windturbineconfig(name)
                if rng.random() < 0.25:
                    tb = res.windturbine_smooth(tb, params=True)
                method = "logarithmic"
                xds = dataset(dict(wnd100m=v, roughness=z0), time, x, y)
                ref = conv.convert_wind(xds, tb, method).values
                got = orc.convert_wind(v, z0, np.asarray(tb["V"], float), np.asarray(tb["POW"], float), float(tb["P"]),
                                       float(tb["hub_height"]), 100.0, method)
                what = f"wind {name}"
            elif fam == "heat":
                tk = 283 + 12 * rng.standard_normal((T, Y, X))
                tk[rng.random((T, Y, X)) < 0.05] = np.nan
                shift = float(rng.choice([0.0, 1.0, -5.0, 3.5]))
                thr, a, c0 = float(rng.choice([15.0, 10.5])), float(rng.choice([1.0, 2.5])), float(rng.choice([0.0, 1.25]))
                xds = dataset(dict(temperature=tk), time, x, y)
                ref = conv.convert_heat_demand(xds, thr, a, c0, shift).values
                ptr, _ = orc.day_groups(time, shift)
                got = orc.convert_heat_demand(tk, ptr, thr, a, c0)
                what = f"heat_demand shift={shift}"
            else:
                r = rng.random((T, Y, X)) * 1e-3
                r[rng.random((T, Y, X)) < 0.03] = np.nan
                h = 2000 * rng.random((Y, X))
                wh = bool(rng.random() < 0.7)
                xds = dataset(dict(runoff=r, height=h), time, x, y)
                ref = conv.convert_runoff(xds, wh).values
                got = orc.convert_runoff(r, h if wh else None)
                what = "runoff"
        compare(got, ref, f"case {case} ({T},{Y},{X}) {what}", stats)
        # gateway algebra of convert_and_aggregate on the same case (runoff only: any cube does)
        if fam == "runoff":
            import scipy.sparse as sp

            with warnings.catch_warnings(), np.errstate(all="ignore"):
                warnings.simplefilter("ignore")
                S = Y * X
                agg = str(rng.choice(["matrix", "layout", "both"]))
                tagg = [None, "sum", "mean"][int(rng.integers(3))]
                pu = bool(rng.random() < 0.4)
                M = None
                if agg in ("matrix", "both"):
                    M = sp.random(int(rng.integers(1, 6)), S, density=float(rng.choice([0.1, 0.5, 1.0])),
                                  random_state=int(rng.integers(1 << 30)), format="csr")
                    if M.nnz:
                        M.data[rng.integers(0, M.nnz, size=max(1, M.nnz // 10))] = rng.choice([0.0, -1.5])
                lay = None
                if agg in ("layout", "both"):
                    lay = rng.random((Y, X)) * 3
                    lay[rng.random((Y, X)) < 0.2] = 0.0
                cut = MockCutout(xds)
                kw = dict(weight_with_height=wh, aggregate_time=tagg, per_unit=pu)
                if M is not None:
                    kw["matrix"] = M
                if lay is not None:
                    kw["layout"] = xr.DataArray(lay, dims=["y", "x"], coords={"y": y, "x": x})
                gref = conv.runoff(cut, **kw)
                gref = gref.values
                gor, _ = orc.gateway(orc.convert_runoff(r, h if wh else None).reshape(T, S), M, lay, pu, tagg)
                if gref.shape != np.shape(gor) and gref.T.shape == np.shape(gor):
                    gref = gref.T
            compare(gor, gref, f"case {case} gateway agg={agg} tagg={tagg} per_unit={pu}", stats)
    print(f"{n} cases, {stats['fails']} mismatches, worst error {stats['worst']:.3e} of the rtol 1e-10 / atol 1e-12 max allowance"
          + (f" ({stats['worst_what']})" if stats.get("worst_what") else ""))
    return 1 if stats["fails"] else 0


if __name__ == "__main__":
    sys.exit(main())
