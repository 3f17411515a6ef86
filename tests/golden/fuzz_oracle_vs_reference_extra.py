#!/usr/bin/env python3
"""
Oracle vs the reference's own code (run under the stand-in), the families fuzz_oracle_vs_reference.py does not draw -
kept in a script of its own so that the recorded runs of that one stay reproducible:

  solarpos   the computed solar position (pv/solar_position.py:62-121): random start times, hourly / half-hourly /
             3-hourly axes, time shifts, grids on both hemispheres and across the date line
  pvsp       convert_pv on a dataset WITHOUT stored angles (the reference computes them, with its DeprecationWarning)
  windx      the power law, the fast lane and the closest-height rule of extrapolate_wind_speed, directly and through
             convert_wind
  thermo     temperature, soil temperature, dewpoint temperature, coefficient of performance (air / soil, default and
             custom coefficients), cooling demand

Build container only (needs /root/reference):  python tests/golden/fuzz_oracle_vs_reference_extra.py [n_cases] [seed]
"""
import sys
import warnings
from pathlib import Path

import numpy as np
import pandas as pd

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent.parent))
import fuzz_oracle_vs_reference as F  # noqa: E402  (installs the stand-in, loads the reference modules)
from oracle import atlite_oracle as orc  # noqa: E402
from tests import helpers as H  # noqa: E402

conv, res, orient = F.conv, F.res, F.orient
solpos = F.refshim.reference("atlite.pv.solar_position")
windmod = F.refshim.reference("atlite.wind")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    stats = dict(worst=0.0, fails=0)
    seen = {}
    for case in range(n):
        T, Y, X = int(rng.integers(3, 40)), int(rng.integers(1, 7)), int(rng.integers(2, 9))
        x = np.sort(rng.uniform(-180, 180, X)) if rng.random() < 0.3 else H.grid(max(Y, 2), X)[0]
        y = np.sort(rng.uniform(-80, 80, Y))
        start = pd.Timestamp("2010-01-01") + pd.Timedelta(minutes=int(rng.integers(0, 12 * 365 * 24 * 2)) * 30)
        freq = str(rng.choice(["h", "30min", "3h"]))
        time = pd.date_range(start, periods=T, freq=freq)
        fam = str(rng.choice(["solarpos", "pvsp", "windx", "thermo"]))
        seen[fam] = seen.get(fam, 0) + 1
        with warnings.catch_warnings(), np.errstate(all="ignore"):
            warnings.simplefilter("ignore")
            if fam == "solarpos":
                shift = str(rng.choice(["0h", "-30min", "+30min", "-1h", "15min"]))
                xds = F.dataset(dict(temperature=np.zeros((T, Y, X))), time, x, y)
                sp = solpos.SolarPosition(xds, time_shift=shift)
                alt, az = orc.solar_position(time, x, y, shift)
                F.compare(alt, sp["altitude"].transpose("time", "y", "x").values, f"case {case} solarpos altitude {shift} {freq}", stats)
                F.compare(az, sp["azimuth"].transpose("time", "y", "x").values, f"case {case} solarpos azimuth {shift} {freq}", stats)
                continue
            if fam == "pvsp":
                ds = H.pv_dataset(T, Y, X, seed=int(rng.integers(1 << 30)))
                ds = {k: v.reshape(T, Y, X).copy() for k, v in ds.items() if not k.startswith("solar_")}
                sl, az = float(rng.random() * 90), float(rng.random() * 360)
                xds = F.dataset(ds, time, x, y)
                pc = res.get_solarpanelconfig(str(rng.choice(["CSi", "CdTe"])))
                ref = conv.convert_pv(xds, pc, orient.get_orientation({"slope": sl, "azimuth": az}), tracking=None).transpose("time", "y", "x").values
                alt, azi = orc.solar_position(time, x, y, "0h")
                got = orc.convert_pv(dict(ds, solar_altitude=alt, solar_azimuth=azi), pc, orc.orientation_constant(sl, az))
                what = "pv with the computed solar position"
            elif fam == "windx":
                v = 12 * rng.random((T, Y, X)) ** 1.5
                v[rng.random((T, Y, X)) < 0.03] = rng.choice([np.nan, 0.0, 25.0, np.inf, -1.0])
                z0 = np.exp(np.log(1e-3) + rng.random((T, Y, X)) * np.log(2e3))
                sh = rng.uniform(-0.1, 0.6, (T, Y, X))
                sh[rng.random((T, Y, X)) < 0.03] = rng.choice([np.nan, 0.0, 5.0])
                heights = {100: v, 10: 0.7 * v} if rng.random() < 0.5 else {100: v}
                xds = F.dataset(dict({f"wnd{h}m": a for h, a in heights.items()}, roughness=z0, wnd_shear_exp=sh), time, x, y)
                to_h = float(rng.choice([80.0, 30.0, 120.5, 100.0, 10.9, 250.0]))
                method = str(rng.choice(["power", "logarithmic"]))
                ref = windmod.extrapolate_wind_speed(xds, to_h, method=method).transpose("time", "y", "x").values
                if int(to_h) in heights:
                    got = heights[int(to_h)]
                else:
                    hs = np.asarray(list(heights))
                    from_h = int(hs[np.argmin(np.abs(hs - to_h))])
                    got = orc.extrapolate_wind_speed(heights[from_h], z0 if method == "logarithmic" else sh, to_h, from_h, method)
                F.compare(got, ref, f"case {case} extrapolate {method} to {to_h} from {sorted(heights)}", stats)
                tb = res.get_windturbineconfig(str(rng.choice(["Vestas_V112_3MW", "Enercon_E101_3000kW", "Siemens_SWT_107_3600kW"])))
                ref = conv.convert_wind(xds, tb, method).transpose("time", "y", "x").values
                hub = float(tb["hub_height"])
                if int(hub) in heights:
                    hubv = heights[int(hub)]
                else:
                    hs = np.asarray(list(heights))
                    from_h = int(hs[np.argmin(np.abs(hs - hub))])
                    hubv = orc.extrapolate_wind_speed(heights[from_h], z0 if method == "logarithmic" else sh, hub, from_h, method)
                got = np.interp(hubv, np.asarray(tb["V"], float), np.asarray(tb["POW"], float) / float(tb["P"]))
                what = f"convert_wind {method} hub {hub}"
            else:
                tk = 283 + 15 * rng.standard_normal((T, Y, X))
                tk[rng.random((T, Y, X)) < 0.05] = np.nan
                kind = str(rng.choice(["temperature", "soil", "dewpoint", "cop_air", "cop_soil", "cop_custom", "cooling"]))
                if kind == "temperature":
                    ref, got = conv.convert_temperature(F.dataset(dict(temperature=tk), time, x, y)).values, orc.convert_temperature(tk)
                elif kind == "soil":
                    ref = conv.convert_soil_temperature(F.dataset({"soil temperature": tk}, time, x, y)).values
                    got = orc.convert_soil_temperature(tk)
                elif kind == "dewpoint":
                    ref = conv.convert_dewpoint_temperature(F.dataset({"dewpoint temperature": tk}, time, x, y)).values
                    got = orc.convert_temperature(tk)
                elif kind.startswith("cop"):
                    source = "soil" if kind == "cop_soil" else "air"
                    var = "soil temperature" if source == "soil" else "temperature"
                    sink = float(rng.choice([55.0, 35.0, 70.5]))
                    cs = (None, None, None) if kind != "cop_custom" else (float(rng.uniform(5, 9)), float(rng.uniform(-0.2, -0.1)), float(rng.uniform(5e-4, 8e-4)))
                    d = (6.81, -0.121, 0.000630) if source == "air" else (8.77, -0.150, 0.000734)
                    full = tuple(d[i] if c is None else c for i, c in enumerate(cs))
                    ref = conv.convert_coefficient_of_performance(F.dataset({var: tk}, time, x, y), source, sink, *full).values
                    got = orc.convert_coefficient_of_performance(tk, source, sink, *cs)
                else:
                    shift = float(rng.choice([0.0, 2.0, -7.0]))
                    thr, a, c0 = float(rng.choice([23.0, 18.5])), float(rng.choice([1.0, 0.4])), float(rng.choice([0.0, 2.0]))
                    ref = conv.convert_cooling_demand(F.dataset(dict(temperature=tk), time, x, y), thr, a, c0, shift).values
                    ptr, _ = orc.day_groups(time, shift)
                    got = orc.convert_cooling_demand(tk, ptr, thr, a, c0)
                what = f"thermo {kind}"
            F.compare(got, ref, f"case {case} ({T},{Y},{X}) {what}", stats)
    print(f"{n} cases, {stats['fails']} mismatches, worst error {stats['worst']:.3e} of the rtol 1e-10 / atol 1e-12 max allowance"
          + (f" ({stats['worst_what']})" if stats.get("worst_what") else "") + f"; families {seen}")
    return 1 if stats["fails"] else 0


if __name__ == "__main__":
    sys.exit(main())
