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


class Affine(tuple):
    """The six coefficients (a, b, c, d, e, f) of x' = a*col + b*row + c, y' = d*col + e*row + f; stands in for
    rasterio's Affine, which is not a dependency here."""

    def __new__(cls, a, b, c, d, e, f):
        return super().__new__(cls, (float(a), float(b), float(c), float(d), float(e), float(f)))

    a, b, c, d, e, f = (property(lambda self, i=i: self[i]) for i in range(6))

    def __mul__(self, colrow):
        col, row = colrow
        return (self[0] * col + self[1] * row + self[2], self[3] * col + self[4] * row + self[5])


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
        return round(float((x[-1] - x[0]) / (len(x) - 1)), 8) if len(x) > 1 else 1.0  # cutout.py:314-320

    @property
    def dy(self):
        y = self.coords["y"]
        return round(float((y[-1] - y[0]) / (len(y) - 1)), 8) if len(y) > 1 else 1.0

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

    def indicatormatrix(self, shapes, shapes_crs=4326, where=None, _share=False):
        """Share of every grid cell lying in every shape, sparse (N x Y*X) (cutout.py:492-515).

        where: "device" (the areas are line integrals evaluated on the GPU), "host" (the C++ polygon clipper) or
        None = ATLITE_HIP_INDICATOR, default "device".  Both implement the same contract (1e-13 of a cell apart);
        the conversion itself has no host path either way.  ``shapes_crs`` other than the cutout's: one of the projections
        written out in ``atlite_amd.crs`` (EPSG:3035, 3857, UTM zones) for a cutout in geographic coordinates."""
        where = where or os.environ.get("ATLITE_HIP_INDICATOR", "device")
        if where not in ("device", "host"):
            raise ValueError(f"where must be 'device' or 'host', not {where!r}")
        ctx = None
        if where == "device":
            from .device import default_context

            ctx = default_context()
        cache = self.__dict__.setdefault("_indicator_cache", {})
        # shapes in another crs: their vertices are moved into the cutout's (atlite/gis.py:130), then the same clippers
        return gis.compute_indicatormatrix(self.coords["x"], self.coords["y"], shapes, ctx=ctx, cache=cache,
                                           shapes_crs=shapes_crs, grid_crs=self.crs, share=_share)

    def uniform_layout(self):
        from .labeled import LabeledArray

        Y, X = self.shape
        return LabeledArray(np.ones((Y, X)), ("y", "x"), {"y": self.coords["y"], "x": self.coords["x"]},
                            name="Capacity")

    def _layout(self, values, name=None):
        from .labeled import LabeledArray

        return _convert._finish(LabeledArray(np.asarray(values, dtype=np.float64), ("y", "x"),
                                             {"y": self.coords["y"], "x": self.coords["x"]}, name=name))

    def area(self, crs=None):
        """Area per grid cell, (y, x) (cutout.py:537-558).  In the cutout's own crs the reference's
        ``grid.to_crs(crs).area`` is the planar area of the cell boxes, dx * dy; any other crs needs pyproj."""
        if crs is not None and crs != self.crs:
            raise NotImplementedError("areas in another crs need pyproj; only the cutout's own crs is supported")
        return self._layout(np.full(self.shape, abs(self.dx * self.dy)))

    def uniform_density_layout(self, capacity_density, crs=None):
        """Capacity layout from a uniform capacity density (cutout.py:566-585)."""
        a = self.area(crs)
        return self._layout(capacity_density * np.asarray(a.values))

    def layout_from_capacity_list(self, data, col="Capacity"):
        """Capacity layout from a list of plants with columns 'x', 'y' and ``col``: every entry is added to the grid
        cell nearest to its coordinate, entries outside the cutout to the border cells (cutout.py:596-642)."""
        xg, yg = np.asarray(self.coords["x"]), np.asarray(self.coords["y"])
        px, py = np.asarray(data["x"], dtype=np.float64), np.asarray(data["y"], dtype=np.float64)

        def nearest(grid, v):
            i = np.clip(np.searchsorted(grid, v, side="left"), 0, len(grid) - 1)
            # like the reference, i == 0 compares with grid[-1]: an entry at or below the first coordinate lands in the LAST
            # cell (index -1) - kept, so that layouts equal the reference's for the same list
            return i - (v - grid[i - 1] < grid[i] - v)

        ix, iy = nearest(xg, px), nearest(yg, py)
        out = np.zeros(self.shape)
        np.add.at(out, (iy, ix), np.asarray(data[col], dtype=np.float64))
        return self._layout(out, name=col)

    def sel(self, path=None, bounds=None, buffer=0, **kwargs):
        """Select parts of the cutout (cutout.py:387-414): ``bounds`` = (x1, y1, x2, y2) with an optional ``buffer``
        around them, and / or label selections along time, y, x as for ``xarray.Dataset.sel`` (``Dataset.sel``).
        Returns a new Cutout; nothing is written (``path`` only names it)."""
        if bounds is not None:
            x1, y1, x2, y2 = (float(b) for b in bounds)
            if buffer > 0:  # shapely's box(*bounds).buffer(buffer).bounds: the box grown by the buffer on every side
                x1, y1, x2, y2 = x1 - buffer, y1 - buffer, x2 + buffer, y2 + buffer
            kwargs.update(x=slice(x1, x2), y=slice(y1, y2))
        out = Cutout(self.data.sel(**kwargs), crs=self.crs, devices=self.devices)
        if path is not None:
            out.path = os.fspath(path)
        elif getattr(self, "path", None):
            out.path = self.path
        return out

    def merge(self, other, path=None, **kwargs):
        """Merge the variables of two cutouts on the same grid and time axis into one (cutout.py:416-450).  The
        reference hands differing coordinates to ``xarray.merge`` (an outer join filled with NaN); here both cutouts
        must share their coordinates.  ``module`` and ``prepared_features`` attributes are united like there."""
        assert isinstance(other, Cutout)
        for k in ("time", "y", "x"):
            if not np.array_equal(np.asarray(self.coords[k]), np.asarray(other.coords[k])):
                raise NotImplementedError(f"merging cutouts with different {k!r} coordinates needs xarray's outer join")
        both = set(self.data.data_vars) & set(other.data.data_vars)
        for k in both:
            if not np.array_equal(np.asarray(self.data[k].values), np.asarray(other.data[k].values), equal_nan=True):
                raise ValueError(f"conflicting values for variable {k!r} on objects to be combined")
        attrs = {**self.data.attrs, **other.data.attrs}
        mods = [m for c in (self, other) for m in np.atleast_1d(c.module if c.module is not None else []).tolist()]
        attrs["module"] = list(dict.fromkeys(mods))
        feats = list(dict.fromkeys(list(self.prepared_features.index.unique("feature")) + list(other.prepared_features.index.unique("feature"))))
        attrs["prepared_features"] = feats
        variables = {k: self.data[k] for k in self.data.data_vars}
        variables.update({k: other.data[k] for k in other.data.data_vars if k not in variables})
        data = Dataset({}, {k: self.coords[k] for k in ("time", "y", "x")}, attrs, chunked=self.data.chunked or other.data.chunked)
        for k, la in variables.items():
            data[k] = labeled.LabeledArray(la.data, la.dims, attrs=la.attrs, name=la.name)
        out = Cutout(data, crs=self.crs, devices=self.devices)
        if path is not None:
            out.path = os.fspath(path)
        return out

    def equals(self, other):
        """Same coordinates and variables, value by value (NaN == NaN); the path is ignored (cutout.py:587-594)."""
        if not isinstance(other, Cutout):
            return NotImplemented
        a, b = self.data, other.data
        if set(a.data_vars) != set(b.data_vars):
            return False
        for k in ("time", "y", "x"):
            if (k in a.coords) != (k in b.coords) or (k in a.coords and not np.array_equal(np.asarray(a.coords[k]), np.asarray(b.coords[k]))):
                return False
        for k in a.data_vars:
            u, v = a[k], b[k]
            if u.dims != v.dims or not np.array_equal(np.asarray(u.values), np.asarray(v.values), equal_nan=True):
                return False
        return True

    # -- descriptive properties (cutout.py:216-248, 284-345) ----------------------------------
    @property
    def name(self):
        p = getattr(self, "path", None)
        return os.path.splitext(os.path.basename(p))[0] if p else None

    @property
    def module(self):
        return self.data.attrs.get("module")

    @property
    def chunks(self):
        c = getattr(getattr(self.data, "file", None), "chunks", None)
        return c if c else None

    @property
    def transform(self):
        """(a, b, c, d, e, f) of the affine cell-index -> coordinate map, the argument order of rasterio's Affine."""
        return Affine(self.dx, 0.0, float(self.coords["x"][0]) - self.dx / 2, 0.0, self.dy, float(self.coords["y"][0]) - self.dy / 2)

    @property
    def transform_r(self):
        return Affine(self.dx, 0.0, float(self.coords["x"][0]) - self.dx / 2, 0.0, -self.dy, float(self.coords["y"][-1]) + self.dy / 2)

    @property
    def dt(self):
        t = self.coords["time"]
        return pd.infer_freq(t) if len(t) >= 3 else None

    @property
    def prepared_features(self):
        """Series of the variables indexed by (module, feature) where the variables carry those attributes
        (cutout.py:335-345); variables without them are listed under the dataset's module and their own name."""
        idx = []
        for v in self.data:
            at = getattr(self.data[v], "attrs", None) or {}
            idx.append((at.get("module", self.module), at.get("feature", v)))
        index = pd.MultiIndex.from_tuples(idx, names=["module", "feature"]) if idx else pd.MultiIndex.from_arrays([[], []], names=["module", "feature"])
        return pd.Series(list(self.data), index, dtype=object)

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
