"""
A duck-typed cutout: exactly what the hot path touches on ``atlite.Cutout`` - ``.data``,
``.grid``, ``.indicatormatrix`` and the conversion methods bound as attributes
(atlite/cutout.py:355-376, 492-515, 653-689).  ``Cutout(path)`` opens an existing NetCDF-4
cutout file through the native reader (``atlite_amd.io``; variables stay on disk and are streamed
to the device), like ``atlite.Cutout(path)`` on a prepared cutout (cutout.py:143,151-153).
Creating / preparing cutouts (CDS downloads, GIS reprojection) is outside the hot path and not
provided; alternatively build a ``Dataset`` from arrays you already hold (host NumPy, torch CUDA
tensors or DeviceArrays), or pass an ``xarray.Dataset`` where xarray is installed.
"""

from __future__ import annotations

import os

import numpy as np
import pandas as pd

from . import convert as _convert
from . import gis, labeled
from .labeled import Dataset


class Cutout:
    def __init__(self, data=None, crs=4326, path=None, devices=None):
        import os

        # devices=[0, 1, ...]: conversions on this cutout shard their time axis over these GPUs
        # (atlite_amd.multigpu); None -> atlite_amd.set_devices() / ATLITE_HIP_DEVICES / one device
        self.devices = None if devices is None else [int(d) for d in devices]

        if path is not None and data is None:
            data = path
        if isinstance(data, (str, os.PathLike)):
            from .io import open_cutout

            if not os.path.exists(data):
                raise NotImplementedError(
                    f"{data}: no such cutout file; creating / preparing cutouts is not part of atlite_amd")
            self.path = os.fspath(data)
            data = open_cutout(data)
        if labeled.xr is not None and isinstance(data, labeled.xr.Dataset):
            data = Dataset.from_xarray(data)
        if not isinstance(data, Dataset):
            raise TypeError("Cutout needs a cutout file path, an atlite_amd.Dataset or an xarray.Dataset")
        self.data = data
        self.crs = crs

    # -- geometry (atlite/cutout.py:250-376) ------------------------------------------------
    @property
    def coords(self):
        return self.data.coords

    @property
    def shape(self):
        return len(self.coords["y"]), len(self.coords["x"])

    @property
    def dx(self):
        x = self.coords["x"]
        return float((x[-1] - x[0]) / (len(x) - 1)) if len(x) > 1 else 1.0

    @property
    def dy(self):
        y = self.coords["y"]
        return float((y[-1] - y[0]) / (len(y) - 1)) if len(y) > 1 else 1.0

    @property
    def extent(self):
        x, y = self.coords["x"], self.coords["y"]
        return np.array([x[0] - self.dx / 2, x[-1] + self.dx / 2, y[0] - self.dy / 2, y[-1] + self.dy / 2])

    @property
    def bounds(self):
        e = self.extent
        return np.array([e[0], e[2], e[1], e[3]])

    @property
    def grid(self):
        """Frame with cell-centre columns 'x' and 'y', cells in y-major order (cutout.py:369-376)."""
        xs, ys = np.meshgrid(self.coords["x"], self.coords["y"])
        return pd.DataFrame({"x": np.ravel(xs), "y": np.ravel(ys)})

    def indicatormatrix(self, shapes, shapes_crs=4326, where=None):
        """Share of every grid cell lying in every shape, sparse (N x Y*X) (cutout.py:492-515).

        where: "device" (the areas are line integrals evaluated on the GPU), "host" (the C++ polygon clipper) or
        None = ATLITE_HIP_INDICATOR, default "device".  Both implement the same contract (1e-13 of a cell apart);
        the conversion itself has no host path either way."""
        if shapes_crs != self.crs:
            raise NotImplementedError("reprojection of shapes needs pyproj; pass shapes in the cutout's crs")
        where = where or os.environ.get("ATLITE_HIP_INDICATOR", "device")
        if where not in ("device", "host"):
            raise ValueError(f"where must be 'device' or 'host', not {where!r}")
        ctx = None
        if where == "device":
            from .device import default_context

            ctx = default_context()
        cache = self.__dict__.setdefault("_indicator_cache", {})
        return gis.compute_indicatormatrix(self.coords["x"], self.coords["y"], shapes, ctx=ctx, cache=cache)

    def uniform_layout(self):
        from .labeled import LabeledArray

        Y, X = self.shape
        return LabeledArray(np.ones((Y, X)), ("y", "x"), {"y": self.coords["y"], "x": self.coords["x"]},
                            name="Capacity")

    # -- conversion methods bound like the reference does (cutout.py:653-689) -----------------
    convert_and_aggregate = _convert.convert_and_aggregate
    pv = _convert.pv
    irradiation = _convert.irradiation
    solar_thermal = _convert.solar_thermal
    wind = _convert.wind
    heat_demand = _convert.heat_demand
    cooling_demand = _convert.cooling_demand
    temperature = _convert.temperature
    soil_temperature = _convert.soil_temperature
    dewpoint_temperature = _convert.dewpoint_temperature
    coefficient_of_performance = _convert.coefficient_of_performance
    runoff = _convert.runoff

    def __repr__(self):
        t = self.coords["time"]
        return (f"<Cutout (amd) x={self.coords['x'][0]:.2f}-{self.coords['x'][-1]:.2f} "
                f"y={self.coords['y'][0]:.2f}-{self.coords['y'][-1]:.2f} time={t[0]}..{t[-1]} "
                f"vars={list(self.data.data_vars)}>")
