#!/usr/bin/env python3
"""
Randomised differential test of the pv option space (GPU vs the NumPy oracle, rtol 1e-10):
tracking x trigon model x panel x orientation kind x dataset flavour (direct/diffuse or influx-only,
albedo or outflux) x grid shape x aggregation, with hostile values mixed in (NaN / zero / negative
radiation, sun exactly at the horizon, exactly in the panel azimuth, zenith).  Run on the GPU box:

    python tests/fuzz_pv_options.py [n_cases] [seed]
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from atlite_amd import Cutout, Dataset  # noqa: E402
from atlite_amd.resource import get_solarpanelconfig  # noqa: E402
from oracle import atlite_oracle as orc  # noqa: E402
from tests import helpers as H  # noqa: E402

TRACK = [None, "horizontal", "tilted_horizontal", "vertical", "dual"]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    worst = 0.0
    worst_case = ""
    fails = 0
    for case in range(n):
        T, Y, X = int(rng.integers(5, 40)), int(rng.integers(1, 12)), int(rng.integers(2, 40))
        ds = H.pv_dataset(T, Y, X, seed=int(rng.integers(1 << 30)))
        ds = {k: v.reshape(T, Y, X).copy() for k, v in ds.items()}
        # hostile values
        for k in ("influx_direct", "influx_diffuse", "temperature", "albedo"):
            m = rng.random((T, Y, X)) < 0.02
            ds[k][m] = rng.choice([np.nan, 0.0, -5.0, 1e4])
        m = rng.random((T, Y, X)) < 0.02
        ds["solar_altitude"][m] = rng.choice([0.0, np.radians(1.0), np.pi / 2, -0.3, np.nan])
        m = rng.random((T, Y, X)) < 0.02
        ds["solar_azimuth"][m] = np.pi  # exactly the panel azimuth: tan(rotation) = 0 for the trackers
        m = rng.random((T, Y, X)) < 0.01
        ds["influx_toa"][m] = 0.0
        flavour = rng.choice(["split", "influx", "outflux", "sarah"])
        if flavour in ("outflux", "sarah"):  # "sarah": total influx AND outflux (the fast family's influx head)
            ds["outflux"] = (ds["influx_direct"] + ds["influx_diffuse"]) * ds["albedo"]
            del ds["albedo"]
        if flavour in ("influx", "sarah"):
            ds["influx"] = ds["influx_direct"] + ds["influx_diffuse"]
            ds["humidity"] = rng.random((T, Y, X))
            del ds["influx_direct"], ds["influx_diffuse"]
        # a quarter of the cases carry no stored solar angles: in-kernel solar position (the oracle
        # computes the angles the way pv/solar_position.py does)
        computed_sp = bool(rng.random() < 0.25)
        trk = TRACK[int(rng.integers(5))]
        tm = str(rng.choice(["simple", "other"]))
        cs = str(rng.choice(["simple", "enhanced"]))
        panel = str(rng.choice(["CSi", "CdTe", "KANENA"]))
        okind = str(rng.choice(["const", "latitude_optimal", "latitude"]))
        x, y = H.grid(Y, X)
        ods = ds  # what the oracle sees
        if computed_sp:
            alt, az = orc.solar_position(H.times(T), x, y, "0h")
            ods = dict(ds, solar_altitude=alt, solar_azimuth=az)
            ds = {k: v for k, v in ds.items() if not k.startswith("solar_")}
        c = Cutout(Dataset(ds, dict(time=H.times(T), y=y, x=x)))
        if okind == "const":
            sl, az = float(rng.choice([0.0, 30.0, 90.0, rng.random() * 90])), float(rng.choice([180.0, 0.0, rng.random() * 360]))
            ospec = {"slope": sl, "azimuth": az}
            ori = dict(slope=np.radians(sl), azimuth=np.radians(az))
        else:
            ospec = okind
            lat = np.radians(y)
            o = orc.orientation_latitude_optimal(lat) if okind == "latitude_optimal" else orc.orientation_latitude(lat)
            ori = dict(slope=np.broadcast_to(np.asarray(o["slope"])[:, None], (Y, X)),
                       azimuth=np.broadcast_to(np.asarray(o["azimuth"], dtype=float).reshape(-1, 1) if np.ndim(o["azimuth"])
                                               else np.asarray(o["azimuth"], dtype=float), (Y, X)))
        what = str(rng.choice(["pv", "irradiation", "thermal"]))
        pcfg = get_solarpanelconfig(panel)
        try:
            import warnings

            with np.errstate(all="ignore"), warnings.catch_warnings():
                warnings.simplefilter("ignore", DeprecationWarning)  # the reference warns about the compute branch
                if what == "pv":
                    got = c.pv(panel=panel, orientation=ospec, tracking=trk, trigon_model=tm, clearsky_model=cs,
                               aggregate_time=None).values
                    ref = orc.convert_pv_general(ods, pcfg, ori, trk, tm, cs)
                elif what == "irradiation":
                    q = str(rng.choice(["total", "direct", "diffuse", "ground"]))
                    got = c.irradiation(orientation=ospec, irradiation=q, tracking=trk, trigon_model=tm,
                                        clearsky_model=cs, aggregate_time=None).values
                    ref = orc.convert_irradiation(ods, ori, trk, q, tm, cs)
                else:
                    got = c.solar_thermal(orientation=ospec, trigon_model=tm, clearsky_model=cs, aggregate_time=None).values
                    ref = orc.convert_solar_thermal(ods, ori, tm, cs)
        except Exception as e:  # noqa: BLE001
            print(f"case {case}: {what} {trk} {tm} {cs} {panel} {okind} {flavour} ({T},{Y},{X}) RAISED {type(e).__name__}: {e}")
            fails += 1
            continue
        ref = np.broadcast_to(ref, got.shape)
        scale = np.nanmax(np.abs(ref)) if np.isfinite(ref).any() else 1.0
        nan_mismatch = int((np.isnan(got) != np.isnan(ref)).sum())
        # the repository's tolerance: |got - ref| <= rtol |ref| + atol max|ref|, rtol 1e-10,
        # atol 1e-12; reported as a fraction of that allowance (1.0 = at the limit)
        with np.errstate(all="ignore"):
            err = np.abs(got - ref) / (1e-10 * np.abs(ref) + 1e-12 * max(scale, 1e-300))
        err = np.where(np.isnan(ref) | np.isnan(got) | (got == ref), 0.0, err)
        e = float(err.max()) if err.size else 0.0
        inf_mismatch = int((np.isinf(got) != np.isinf(ref)).sum())
        bad = e > 1.0 or nan_mismatch or inf_mismatch
        if e > worst:
            i = np.unravel_index(np.argmax(err), err.shape)
            worst_case = f"{what} trk={trk} {tm} {cs} {panel} ori={okind} {flavour} ({T},{Y},{X}) at {i}: got {got[i]!r} ref {ref[i]!r} scale {scale:.3g}"
        worst = max(worst, e)
        if bad:
            fails += 1
            i = np.unravel_index(np.argmax(err), err.shape)
            print(f"case {case}: {what} trk={trk} {tm} {cs} {panel} ori={okind} {flavour} sp={computed_sp} ({T},{Y},{X}) error {e:.3e} of the allowance "
                  f"nan-mismatch {nan_mismatch} inf-mismatch {inf_mismatch} at {i}: got {got[i]!r} ref {ref[i]!r}")
    print(f"{n} cases, {fails} failures, worst error {worst:.3e} of the allowance: {worst_case}")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
