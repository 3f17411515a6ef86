"""
``atlite.wind`` next to the hot path: ``extrapolate_wind_speed`` (atlite/wind.py:23-125) as an operation of its own.
Inside ``Cutout.wind()`` the extrapolation is fused with the power-curve interpolation in one kernel; called directly
it runs the same kernel without a power curve (``atl_wind_params.n_knots = 0``) and returns the (time, y, x) wind
speed at ``to_height``.
"""

from __future__ import annotations

from . import convert as _convert


def extrapolate_wind_speed(ds, to_height, from_height=None, method="logarithmic"):
    """
    Extrapolate the wind speed from a given height above ground to another (signature, variable selection, error
    messages and attributes as atlite/wind.py:23-125).  If ``ds`` already holds ``wnd{to_height}m`` it is returned as
    it is; otherwise ``wnd{from_height}m`` (default: the stored height closest to ``to_height``) is extrapolated with
    the "logarithmic" law (needs ``roughness``) or the "power" law (needs ``wnd_shear_exp``) on the GPU.
    """
    to_name = f"wnd{int(to_height):0d}m"
    if to_name in ds:  # fast lane (wind.py:77-79)
        return ds[to_name]
    spec = _convert._WindSpeedSpec(ds, to_height, from_height, method)
    return _convert._finish(_convert._per_cell(spec, ds))
