"""
Host-side, per-time pieces of the solar position algorithm.

``SolarPosition`` (atlite/pv/solar_position.py:71-121) is separable: the almanac quantities
``n, L, g, l, ep, ra, dec`` depend on time only, the hour angle ``h`` on (time, x), and only the
last step (altitude, azimuth) is cube-sized.  The (T)- and (T,X)-sized parts are evaluated
here with NumPy/pandas exactly as the reference writes them (the Julian day is formed by
pandas in float64, which matters at the 1e-10 level: SURVEY.md Appendix A), the cube-sized
part runs on the device.
"""

from __future__ import annotations

import numpy as np
import pandas as pd


def almanac(time, time_shift="0h"):
    """(T,) arrays: right ascension, declination, lmst0 [deg] (solar_position.py:71-97)."""
    t = pd.DatetimeIndex(time) + pd.to_timedelta(time_shift)
    n = np.asarray(t.to_julian_date(), dtype=np.float64) - 2451545.0
    hour = np.asarray(t.hour)
    minute = np.asarray(t.minute)
    L = 280.460 + 0.9856474 * n
    g = np.radians(357.528 + 0.9856003 * n)
    l = np.radians(L + 1.915 * np.sin(g) + 0.020 * np.sin(2 * g))  # noqa: E741
    ep = np.radians(23.439 - 4e-7 * n)
    ra = np.arctan2(np.cos(ep) * np.sin(l), np.cos(l))
    lmst0 = (6.697375 + (hour + minute / 60.0) + 0.0657098242 * n) * 15.0
    dec = np.arcsin(np.sin(ep) * np.sin(l))
    return dict(ra=ra, dec=dec, lmst0=lmst0)


def hour_angle(time, lon_deg, time_shift="0h"):
    """(T, X) hour angle in radians and the (T,) declination (solar_position.py:92-97)."""
    a = almanac(time, time_shift)
    lmst = a["lmst0"][:, None] + np.asarray(lon_deg, dtype=np.float64)[None, :]
    h = (np.radians(lmst) - a["ra"][:, None] + np.pi) % (2 * np.pi) - np.pi
    return h, a["dec"]


def position(time, lon_deg, lat_deg, time_shift="0h"):
    """(T, Y, X) altitude and azimuth in radians on the HOST (solar_position.py:100-114).  Only used to hand a
    ``solar_position`` object to user orientation callbacks of datasets that do not store the angles; the
    conversion itself evaluates this part inside the kernel."""
    h, dec = hour_angle(time, lon_deg, time_shift)
    lat = np.radians(np.asarray(lat_deg, dtype=np.float64))[None, :, None]
    dec, h = dec[:, None, None], h[:, None, :]
    with np.errstate(invalid="ignore", divide="ignore"):
        alt = np.arcsin(np.clip(np.sin(dec) * np.sin(lat) + np.cos(dec) * np.cos(lat) * np.cos(h), -1.0, 1.0))
        az = np.arccos(np.clip((np.sin(dec) * np.cos(lat) - np.cos(dec) * np.sin(lat) * np.cos(h)) / np.cos(alt), -1.0, 1.0))
    az = np.where(h <= 0, az, 2 * np.pi - az)
    return alt, az

