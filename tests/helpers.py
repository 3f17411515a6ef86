"""Shared builders of small synthetic ERA5-shaped inputs for the tests (host NumPy)."""
import numpy as np
import pandas as pd
import scipy.sparse as sp

from oracle import atlite_oracle as orc

CSI = dict(model="huld", name="CSi", c_temp_amb=1, c_temp_irrad=0.035, r_tamb=293, r_tmod=298,
           r_irradiance=1000, k_1=-0.017162, k_2=-0.040289, k_3=-0.004681, k_4=0.000148, k_5=0.000169,
           k_6=0.000005, inverter_efficiency=0.9)

V112 = dict(
    V=np.array([0, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 25, 25], dtype=float),
    POW=np.array([0.000, 0.000, 0.005, 0.150, 0.300, 0.525, 0.905, 1.375, 1.950, 2.580, 2.960, 3.050, 3.060,
                  3.060, 0.000]),
    hub_height=80.0,
    P=3.06,
)


def grid(Y, X):
    x = -25.0 + (70.0 / X) * np.arange(X)
    y = 30.0 + (42.0 / Y) * np.arange(Y)
    return x, y


def times(T, start="2013-01-01"):
    return pd.date_range(start, periods=T, freq="h")


def pv_dataset(T, Y, X, seed=0, start="2013-01-01"):
    """(T, S) float64 arrays of the 7 ERA5 pv variables, physically consistent."""
    rng = np.random.default_rng(seed)
    x, y = grid(Y, X)
    t = times(T, start)
    alt, az = orc.solar_position(t, x, y, "-30min")
    toa = 1361.0 * np.maximum(np.sin(alt), 0.0)
    kt = 0.2 + 0.55 * rng.random((T, Y, X))
    fd = 0.3 + 0.5 * rng.random((T, Y, X))
    ds = dict(
        influx_direct=toa * kt * fd,
        influx_diffuse=toa * kt * (1 - fd),
        influx_toa=toa,
        albedo=0.05 + 0.3 * rng.random((T, Y, X)),
        temperature=283.15 + 10 * rng.standard_normal((T, Y, X)),
        solar_altitude=alt,
        solar_azimuth=az,
    )
    return {k: np.ascontiguousarray(v.reshape(T, Y * X)) for k, v in ds.items()}


def wind_dataset(T, Y, X, seed=0):
    rng = np.random.default_rng(seed)
    u = rng.random((T, Y * X))
    wnd = 8.0 * np.sqrt(-np.log1p(-u)) * (2 / np.sqrt(np.pi))
    rough = np.exp(np.log(1e-3) + rng.random((T, Y * X)) * np.log(1.5e3))
    shear = 0.05 + 0.3 * rng.random((T, Y * X))
    return dict(wnd100m=wnd, roughness=rough, wnd_shear_exp=shear)


def blob_matrix(N, Y, X, seed=0, overlap=True):
    """Sparse N x (Y*X) indicator-like matrix: each row a fuzzy disc of cells, weights in (0,1]."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:Y, 0:X]
    rows, cols, vals = [], [], []
    for n in range(N):
        cy, cx = rng.uniform(0, Y), rng.uniform(0, X)
        r = rng.uniform(0.8, 1.6) * np.sqrt(Y * X / (np.pi * max(N, 1)))
        d = np.hypot(yy - cy, xx - cx)
        w = np.clip(r + 0.5 - d, 0.0, 1.0)
        j = np.flatnonzero(w.ravel() > 0)
        rows += [n] * len(j)
        cols += list(j)
        vals += list(w.ravel()[j])
    return sp.csr_matrix((vals, (rows, cols)), shape=(N, Y * X))


def runoff_post_inputs():
    """Seeded inputs of the runoff() post-processing vectors (tests/golden/make_golden.py freezes the reference's
    outputs for exactly these; two years + a stub of hourly runoff on a 3 x 4 grid, 3 shapes)."""
    r = np.random.default_rng(20240917)
    t2 = pd.date_range("2012-01-01", "2014-01-05", freq="h", inclusive="left")  # 2012 (leap), 2013, 96 h of 2014
    Y2, X2 = 3, 4
    season = 1.0 + 0.8 * np.sin(2 * np.pi * (np.asarray(t2.dayofyear) - 100.0) / 365.0)
    ro2 = -1e-4 * np.log1p(-r.random((len(t2), Y2, X2))) * season[:, None, None]
    height2 = 2000 * r.random((Y2, X2))
    M2 = sp.csr_matrix(np.where(r.random((3, Y2 * X2)) < 0.6, r.random((3, Y2 * X2)), 0.0))
    x2 = 5.0 + 0.25 * np.arange(X2)
    y2 = 45.0 + 0.25 * np.arange(Y2)
    return ro2, height2, M2, ["AT", "CH", "NO"], t2, y2, x2


def runoff_post_sample(T):
    return np.unique(np.concatenate([np.arange(0, 240), np.arange(0, T, 37), np.arange(8700, 8900), np.arange(T - 240, T)]))


def xarray_stand_in():
    """The eager xarray stand-in the golden generator runs the reference under (tests/golden/refshim.py), as a
    module object: xarray itself cannot be installed in this image, so tests of the product's xarray bridge
    (LabeledArray.to_xarray, Dataset.from_xarray, the gateway's isinstance branches) patch it in as a test double."""
    import importlib.util
    from pathlib import Path

    spec = importlib.util.spec_from_file_location("_atl_refshim", Path(__file__).parent / "golden" / "refshim.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod._make_xarray()


def orientation_follow_sun(lon, lat, solar_position):
    """A user orientation callback that reads the sun (pv/orientation.py:104-107 passes solar_position): the panel
    azimuth follows the sun's, the slope is the (clipped) zenith angle.  Written against what both xarray and the
    product's LabeledArray offer (.values, .dims, .coords, the constructor keywords), so make_golden.py runs it inside the reference and the tests inside
    the product."""
    alt = solar_position["altitude"]
    slope = np.clip(np.pi / 2 - np.asarray(alt.values), 0.1, 1.4)  # (time, y, x)
    # a labelled array of the caller's own kind (xr.DataArray / the stand-in / LabeledArray share these keywords)
    return dict(slope=type(alt)(slope, dims=alt.dims, coords=alt.coords), azimuth=solar_position["azimuth"])

