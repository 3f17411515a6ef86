"""
Panel orientation callbacks: ``get_orientation`` and the ``make_*`` factories of
atlite/pv/orientation.py:13-88.  A callback has the signature
``(lon, lat, solar_position) -> dict(slope=..., azimuth=...)`` with angles in radians; ``lon``
and ``lat`` arrive in radians as 1-d LabeledArrays over ``x`` / ``y`` (orientation.py:104-107).
The device kernel consumes the result as scalars or as one (slope, azimuth) pair per cell.
"""

from __future__ import annotations

import sys

import numpy as np

from ..labeled import LabeledArray


def get_orientation(name, **params):
    """
    -`slope` is the angle between ground and panel.
    -`azimuth` is the clockwise angle from North, i.e. azimuth = 180 faces exactly South.
    """
    if isinstance(name, dict):
        params = dict(name)
        name = params.pop("name", "constant")
    return getattr(sys.modules[__name__], f"make_{name}")(**params)


def make_constant(slope, azimuth):
    slope = np.radians(slope)
    azimuth = np.radians(azimuth)

    def constant(lon, lat, solar_position):
        return dict(slope=slope, azimuth=azimuth)

    return constant


def make_latitude(azimuth=180):
    azimuth = np.radians(azimuth)

    def latitude(lon, lat, solar_position):
        return dict(slope=lat, azimuth=azimuth)

    return latitude


def make_latitude_optimal():
    """Tilt rule of thumb by latitude band, equator-facing (orientation.py:26-69)."""

    def latitude_optimal(lon, lat, solar_position):
        alat = np.abs(np.asarray(lat.values, dtype=np.float64))
        slope = np.where(
            alat <= np.radians(25),
            0.87 * alat,
            np.where(alat <= np.radians(50), 0.76 * alat + np.radians(0.31), np.radians(40.0)),
        )
        azimuth = np.where(np.asarray(lat.values) < 0, 0, np.pi).astype(np.float64)
        return dict(
            slope=LabeledArray(slope, lat.dims, lat.coords),
            azimuth=LabeledArray(azimuth, lat.dims, lat.coords),
        )

    return latitude_optimal
