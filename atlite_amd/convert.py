"""
Host-side mirror of atlite's conversion interface for the hot path:

* ``convert_and_aggregate``  - the gateway, atlite/convert.py:59-276
* ``pv`` / ``wind`` / ``heat_demand`` / ``runoff`` - technology wrappers, convert.py:857-936,
  665-744, 421-471, 1037-1084 (same signatures, defaults, warnings and exception classes)
* ``convert_pv`` / ``convert_wind`` / ``convert_heat_demand`` / ``convert_runoff`` - the
  ``convert_func`` callables the wrappers hand to the gateway.

Where the reference builds a lazy dask graph and executes it with ``.load()``, this module
resolves the configuration on the host and issues ONE fused call into ``libatlite_hip.so``
(convert + indicator-matrix aggregation + optional time reduction, on the GPU).  Known
converters are recognised by identity and never materialise the converted cube; an unknown
``convert_func`` is called on the host dataset and its result is aggregated on the device with
``atl_spmm_csr``.  There is no CPU fallback.
"""

from __future__ import annotations

import datetime as dt
import logging
import os
import re
import warnings
from pathlib import Path

import numpy as np
import pandas as pd
import scipy.sparse as sp

from . import labeled
from .device import DeviceArray, default_context
from .gis import spdiag
from .labeled import Dataset, LabeledArray
from .pv.orientation import get_orientation
from .resource import get_solarpanelconfig, get_windturbineconfig, windturbine_smooth

logger = logging.getLogger(__name__)


# --------------------------------------------------------------------------------------
# converter descriptors
# --------------------------------------------------------------------------------------
class _Spec:
    """
    A resolved conversion: knows its output time axis and how to launch itself, either on the whole
    dataset (``run``) or slab by slab for host-resident data (``atlite_amd.streaming``):
    ``time_vars`` / ``static_vars`` name the inputs, ``slab_edges`` cuts the time axis,
    ``for_slab`` returns the spec restricted to a slab, ``out_slots`` its output slot range,
    ``prepare`` uploads per-call constant tables once.
    """

    name = None
    attrs = {}
    time_vars = ()
    static_vars = ()
    aligned_ok = True  # may run on a line-aligned plan (the library still refuses what it cannot re-address: the gateway falls back)

    def time_coord(self, ds):
        return ds.coords["time"]

    def n_slots(self, ds):
        return len(self.time_coord(ds))

    def run(self, ctx, ds, plan, time_agg, out=None):  # -> DeviceArray
        raise NotImplementedError

    # -- slab interface ----------------------------------------------------------------------
    def slab_edges(self, T, steps):
        return [(a, min(a + steps, T)) for a in range(0, T, steps)]

    def for_slab(self, t0, t1):
        return self

    def out_slots(self, t0, t1):
        return t0, t1

    def prepare(self, ctx, ds):
        pass

    # -- multi-GPU interface (atlite_amd.multigpu) -------------------------------------------------
    def shard_edges(self, T, n):
        """Edges of ``n`` contiguous, balanced time shards."""
        from .distributed import time_partition

        return time_partition(T, n)


def _device_group(ds, ctx, names):
    """The device copies of the variables one conversion reads (``Dataset.device_group``: cubes the dataset uploads itself
    share one slot-interleaved allocation)."""
    group = getattr(ds, "device_group", None)
    if group is not None:
        return group(ctx, names)
    return {n: ds.device(ctx, n) for n in dict.fromkeys(names)}


def _need(ds, names, exc, msg):
    for n in names:
        if n not in ds:
            raise exc(msg)


class _SolarPositionView:
    """What orientation callbacks receive as ``solar_position`` (the reference hands over the Dataset SolarPosition
    returns, pv/orientation.py:107): ``view["altitude"]`` / ``view["azimuth"]`` (or ``.altitude`` / ``.azimuth``)
    are (time, y, x) LabeledArrays in radians on the HOST - copied from the device or computed on first access
    only, since the shipped orientation factories never look at them."""

    _names = ("altitude", "azimuth")

    def __init__(self, ds):
        self._ds, self._cache = ds, {}

    def _get(self, name):
        if name not in self._names:
            raise KeyError(name)
        if not self._cache:
            ds = self._ds
            c = {k: ds.coords[k] for k in ("time", "y", "x")}
            if "solar_altitude" in ds and "solar_azimuth" in ds:  # the getter branch, solar_position.py:54-60
                alt, az = ds["solar_altitude"].values, ds["solar_azimuth"].values
            else:
                from . import solar

                alt, az = solar.position(ds.coords["time"], ds.coords["lon"], ds.coords["lat"], "0h")
            shape = tuple(len(c[k]) for k in ("time", "y", "x"))
            for n, v in (("altitude", alt), ("azimuth", az)):
                self._cache[n] = LabeledArray(np.asarray(v).reshape(shape), ("time", "y", "x"), c, {"units": "rad"}, n)
        return self._cache[name]

    def __getitem__(self, name):
        return self._get(name)

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        try:
            return self._get(name)
        except KeyError:
            raise AttributeError(name) from None

    def __contains__(self, name):
        return name in self._names

    def keys(self):
        return list(self._names)

    def __iter__(self):
        return iter(self._names)


class _PvSpec(_Spec):
    """convert_pv / convert_irradiation / convert_solar_thermal: one kernel family."""

    name = "specific generation"
    attrs = {"units": "kWh/kWp"}

    def __init__(self, ds, panel, orientation, tracking, trigon_model="simple", clearsky_model="simple",
                 irradiation="total", panel_model=None, thermal=None):
        if tracking not in (None, "horizontal", "tilted_horizontal", "vertical", "dual"):
            raise AssertionError(
                "Values describing tracking system must be None for no tracking,"
                + "'horizontal' for 1-axis horizontal tracking,"
                + "tilted_horizontal' for 1-axis horizontal tracking of tilted panle,"
                + "vertical' for 1-axis vertical tracking, or 'dual' for 2-axis tracking"
            )
        if irradiation not in ("total", "direct", "diffuse", "ground"):
            raise ValueError(f"irradiation must be 'total', 'direct', 'diffuse' or 'ground', not {irradiation!r}")
        self.options = dict(tracking=tracking, trigon_model="simple" if trigon_model == "simple" else "other",
                            irradiation=irradiation)
        self.vars = ["influx_toa"]
        if "influx" in ds:  # irradiation.py:202-205: Reindl split of the total influx
            if clearsky_model is None:
                clearsky_model = "enhanced" if ("temperature" in ds and "humidity" in ds) else "simple"
            if clearsky_model not in ("simple", "enhanced"):
                raise KeyError("`clearsky model` must be chosen from 'simple' and 'enhanced'")
            self.options["clearsky_model"] = clearsky_model
            self.vars += ["influx"] + (["humidity"] if clearsky_model == "enhanced" else [])
        elif "influx_direct" in ds and "influx_diffuse" in ds:
            self.vars += ["influx_direct", "influx_diffuse"]
        else:
            raise AssertionError(
                "Need either influx or influx_direct and influx_diffuse in the "
                "dataset. Check your cutout and dataset module."
            )
        if "albedo" in ds:
            self.vars.append("albedo")
        elif "outflux" in ds:
            self.vars.append("outflux")
        else:
            raise AssertionError(
                "Need either albedo or outflux as a variable in the dataset. "
                "Check your cutout and dataset module."
            )
        self.solar_tables = None
        if "solar_altitude" in ds and "solar_azimuth" in ds:
            self.vars += ["solar_altitude", "solar_azimuth"]
        else:
            # SolarPosition(ds) compute branch (pv/solar_position.py:62-121, no time shift): the (T)-
            # and (T,X)-sized parts on the host, the cube-sized part inside the kernel
            warnings.warn(
                """The calculation method and handling of solar position variables will change.
    The solar position will in the future be a permanent variables of a cutout.
    Recreate your cutout to remove this warning and permanently include the solar position variables into your cutout.""",
                DeprecationWarning,
            )
            from . import solar

            h, dec = solar.hour_angle(ds.coords["time"], ds.coords["lon"], "0h")
            lat = np.radians(ds.coords["lat"])
            self.solar_tables = dict(sin_dec=np.sin(dec), cos_dec=np.cos(dec), h=h, cos_h=np.cos(h),
                                     sin_lat=np.sin(lat), cos_lat=np.cos(lat))
        self.panel = dict(panel or {})
        if panel_model is None:
            panel_model = self.panel.get("model", "huld")
            if panel_model not in ("huld", "bofinger"):
                raise AssertionError(f"Unknown panel model: {panel_model}")
        self.options["panel_model"] = panel_model
        if thermal:
            self.options.update(thermal)
        if panel_model != "none" or self.options.get("clearsky_model") == "enhanced":
            _need(ds, ["temperature"], KeyError, "temperature")
            self.vars.append("temperature")
        if panel_model == "none":
            self.name = f"{irradiation} tilted"
            self.attrs = {"units": "W m**-2"}
        elif panel_model == "bofinger":
            self.name, self.attrs = "AC power", {}
        elif panel_model == "solar_thermal":
            self.name, self.attrs = None, {}
        # orientation callback evaluated on the host with radian lon / lat and the sun's position
        # (orientation.py:104-107); the shipped factories ignore the third argument, a user callback may
        # read solar_position["altitude"] / ["azimuth"] (fetched or computed only then) and return angles
        # that depend on time - those become two more cubes for the general kernel
        x, y = ds.coords["x"], ds.coords["y"]
        lon = LabeledArray(np.radians(ds.coords["lon"]), ("x",), {"x": x}, name="lon")
        lat = LabeledArray(np.radians(ds.coords["lat"]), ("y",), {"y": y}, name="lat")
        o = orientation(lon, lat, _SolarPositionView(ds))
        T = len(ds.coords["time"])
        self.slope = self._cellwise(o["slope"], T, len(y), len(x))
        self.azimuth = self._cellwise(o["azimuth"], T, len(y), len(x))
        self.per_time = any(np.ndim(v) == 2 for v in (self.slope, self.azimuth))
        if self.per_time:
            self.slope = np.ascontiguousarray(np.broadcast_to(self.slope, (T, len(y) * len(x))))
            self.azimuth = np.ascontiguousarray(np.broadcast_to(self.azimuth, (T, len(y) * len(x))))

    @staticmethod
    def _cellwise(v, T, Y, X):
        """Orientation angle -> float, (S,) per cell or (T, S) per time step and cell."""
        if labeled.xr is not None and isinstance(v, labeled.xr.DataArray):
            v = LabeledArray(v.values, v.dims)
        if isinstance(v, LabeledArray):
            a = np.asarray(v.values, dtype=np.float64)
            full = ("time", "y", "x")
            if any(d not in full for d in v.dims):
                raise ValueError(f"orientation angles must be over (time, y, x), got dims {v.dims}")
            a = np.transpose(a, [v.dims.index(d) for d in full if d in v.dims])
            a = a.reshape([n if d in v.dims else 1 for d, n in zip(full, (T, Y, X))])
            if "time" in v.dims:
                return np.ascontiguousarray(np.broadcast_to(a, (T, Y, X))).reshape(T, Y * X)
            return np.ascontiguousarray(np.broadcast_to(a[0], (Y, X))).reshape(-1)
        a = np.asarray(v, dtype=np.float64)
        if a.ndim == 0:
            return float(a)
        if a.ndim == 3:
            return np.ascontiguousarray(np.broadcast_to(a, (T, Y, X))).reshape(T, Y * X)
        return np.ascontiguousarray(np.broadcast_to(a, (Y, X))).reshape(-1)

    @property
    def time_vars(self):
        return tuple(dict.fromkeys(self.vars))

    @property
    def aligned_ok(self):
        """Line-aligned plans re-address the cubes by slot; the in-kernel solar position (per-time tables) and an
        orientation that follows the sun (the general kernel) index by the time step itself: not worth building the plan."""
        return self.solar_tables is None and not self.per_time

    def prepare(self, ctx, ds):
        """Upload the per-call constant tables once (per-cell orientation, solar position tables)."""
        S = len(ds.coords["y"]) * len(ds.coords["x"])
        if not isinstance(self.slope, DeviceArray) and np.ndim(self.slope) != np.ndim(self.azimuth):
            self.slope = np.broadcast_to(self.slope, (S,))  # mixed scalar / per-cell -> per-cell
            self.azimuth = np.broadcast_to(self.azimuth, (S,))
        if isinstance(self.slope, np.ndarray):
            self.slope, self.azimuth = ctx.upload(self.slope), ctx.upload(self.azimuth)
        if self.solar_tables is not None and isinstance(self.solar_tables["h"], np.ndarray):
            self.solar_tables = {k: ctx.upload(np.ascontiguousarray(v)) for k, v in self.solar_tables.items()}

    def for_slab(self, t0, t1):
        if self.solar_tables is None and not self.per_time:
            return self
        import copy

        def cut(v):  # device tables (after prepare): views; host tables: slices
            return v.slab(t0, t1) if isinstance(v, DeviceArray) else v[t0:t1]

        sub = copy.copy(self)
        if self.solar_tables is not None:
            tb = self.solar_tables
            sub.solar_tables = dict(tb, sin_dec=cut(tb["sin_dec"]), cos_dec=cut(tb["cos_dec"]), h=cut(tb["h"]),
                                    cos_h=cut(tb["cos_h"]))
        if self.per_time:  # the orientation cubes are cut along time with the inputs
            sub.slope, sub.azimuth = cut(self.slope), cut(self.azimuth)
        return sub

    def run(self, ctx, ds, plan, time_agg, out=None):
        T, S = len(ds.coords["time"]), len(ds.coords["y"]) * len(ds.coords["x"])
        inputs = _device_group(ds, ctx, self.vars)
        self.prepare(ctx, ds)
        params = dict(self.panel, slope=self.slope, azimuth=self.azimuth)
        return ctx.pv(inputs, params, T, S, plan=plan, time_agg=time_agg, solar_tables=self.solar_tables,
                      options=dict(self.options, row_len=len(ds.coords["x"])), out=out)


class _IrradiationSpec(_PvSpec):
    def __init__(self, ds, orientation, tracking=None, irradiation="total", trigon_model="simple",
                 clearsky_model="simple"):
        super().__init__(ds, None, orientation, tracking, trigon_model, clearsky_model, irradiation, panel_model="none")


class _SolarThermalSpec(_PvSpec):
    def __init__(self, ds, orientation, trigon_model, clearsky_model, c0, c1, t_store):
        super().__init__(ds, None, orientation, None, trigon_model, clearsky_model, "total", panel_model="solar_thermal",
                         thermal=dict(c0=c0, c1=c1, t_store_K=t_store + 273.15))  # convert.py:554


class _WindSpec(_Spec):
    name = "specific generation"
    attrs = {"units": "MWh/MWp"}

    def __init__(self, ds, turbine, interpolation_method):
        self.V = None if turbine["V"] is None else np.asarray(turbine["V"], dtype=np.float64)
        self.POWn = np.asarray(turbine["POW"] / turbine["P"], dtype=np.float64)  # convert.py:649
        to_height = turbine["hub_height"]
        # extrapolate_wind_speed, atlite/wind.py:75-117
        to_name = f"wnd{int(to_height):0d}m"
        if to_name in ds:
            self.method, self.wnd, self.aux, self.from_height = None, to_name, None, to_height
        else:
            heights = np.asarray([int(s[3:-1]) for s in ds if re.match(r"wnd\d+m", s)])
            if len(heights) == 0:
                raise AssertionError("Wind speed is not in dataset")
            from_height = heights[np.argmin(np.abs(heights - to_height))]
            self.wnd = f"wnd{int(from_height):0d}m"
            self.from_height = float(from_height)
            if interpolation_method == "logarithmic":
                if "roughness" not in ds:
                    raise RuntimeError(
                        "The logarithmic interpolation method requires surface roughness (roughness);\n"
                        "make sure you choose a compatible dataset like ERA5"
                    )
                self.method, self.aux = "logarithmic", "roughness"
            elif interpolation_method == "power":
                if "wnd_shear_exp" not in ds:
                    raise RuntimeError(
                        "The power law interpolation method requires a wind shear exponent (wnd_shear_exp);\n"
                        "make sure you choose a compatible dataset like ERA5 and update your cutout"
                    )
                self.method, self.aux = "power", "wnd_shear_exp"
            else:
                raise ValueError(
                    f"Interpolation method must be 'logarithmic' or 'power',  but is: {interpolation_method}"
                )
        self.to_height = float(to_height)

    @property
    def time_vars(self):
        return (self.wnd,) + ((self.aux,) if self.aux else ())

    def run(self, ctx, ds, plan, time_agg, out=None):
        T, S = len(ds.coords["time"]), len(ds.coords["y"]) * len(ds.coords["x"])
        g = _device_group(ds, ctx, self.time_vars)
        wnd, aux = g[self.wnd], (g[self.aux] if self.aux else None)
        return ctx.wind(wnd, aux, self.V, self.POWn, self.to_height, self.from_height, self.method, T, S,
                        plan=plan, time_agg=time_agg, out=out)


class _WindSpeedSpec(_WindSpec):
    """extrapolate_wind_speed as an operation of its own (atlite/wind.py:23-125): the wind converter without a power
    curve (atl_wind_params.n_knots = 0)."""

    attrs = {"units": "m s**-1"}

    def __init__(self, ds, to_height, from_height=None, method="logarithmic"):
        to_name = f"wnd{int(to_height):0d}m"
        if to_name in ds:  # fast lane (wind.py:77-79): the stored variable as it is, whatever from_height says
            from_height = None
        if from_height is not None:  # a given source height replaces the "closest height" rule
            if f"wnd{int(from_height):0d}m" not in ds:
                raise KeyError(f"wnd{int(from_height):0d}m")
            ds = {k: None for k in ds if not re.match(r"wnd\d+m", k) or k == f"wnd{int(from_height):0d}m"}
        super().__init__(ds, dict(V=None, POW=np.zeros(0), P=1.0, hub_height=to_height), method)
        self.V = self.POWn = None
        self.name = to_name
        if self.method is None:  # fast lane: the stored variable keeps its own attrs
            stored = ds[to_name] if hasattr(ds, "__getitem__") else None
            self.attrs = dict(getattr(stored, "attrs", None) or self.attrs)
        else:
            desc = "logarithmic method with roughness" if self.method == "logarithmic" else "power method with wind shear exponent"
            self.attrs = {"long name": f"extrapolated {to_height} m wind speed using {desc}  and {int(self.from_height)} m wind speed",
                          "units": "m s**-1"}


class _ThermoSpec(_Spec):
    """temperature / soil temperature / dewpoint temperature / COP (convert.py:292-401)."""

    attrs = {}

    def __init__(self, ds, var, fillna0=False, cop=None, name=None):
        _need(ds, [var], KeyError, var)
        self.var, self.fillna0, self.cop, self.name = var, fillna0, cop, name

    @property
    def time_vars(self):
        return (self.var,)

    def run(self, ctx, ds, plan, time_agg, out=None):
        T, S = len(ds.coords["time"]), len(ds.coords["y"]) * len(ds.coords["x"])
        return ctx.thermo(ds.device(ctx, self.var), T, S, fillna0=self.fillna0, cop=self.cop, plan=plan,
                          time_agg=time_agg, out=out)


def _cop_spec(ds, source, sink_T, c0, c1, c2):
    assert source in ["air", "soil"], NotImplementedError("'source' must be one of  ['air', 'soil']")
    if source == "air":
        d = (6.81, -0.121, 0.000630)
        var, fill = "temperature", False
    else:
        d = (8.77, -0.150, 0.000734)
        var, fill = "soil temperature", True
    c0, c1, c2 = (d[i] if v is None else v for i, v in enumerate((c0, c1, c2)))
    return _ThermoSpec(ds, var, fill, cop=(sink_T, c0, c1, c2))


class _HeatSpec(_Spec):
    name = "heat_demand"
    attrs = {}
    cooling = False
    aligned_ok = False  # day groups index the cube by the hour: no line-aligned plans

    def __init__(self, ds, threshold, a, constant, hour_shift):
        _need(ds, ["temperature"], KeyError, "temperature")
        self.threshold_K = threshold + 273.15  # convert.py:413
        self.a, self.constant = a, constant
        # T.resample(time="1D") on the shifted axis: calendar-day bins, label = day start
        t = ds.coords["time"] + pd.Timedelta(np.timedelta64(dt.timedelta(hours=hour_shift)))
        if len(t) == 0:
            self.day_ptr, self.days = np.zeros(1, dtype=np.int64), pd.DatetimeIndex([])
        else:
            day = t.floor("D")
            self.days = pd.date_range(day[0], day[-1], freq="D")
            edges = np.append(self.days.values, self.days.values[-1] + np.timedelta64(1, "D"))
            self.day_ptr = np.searchsorted(day.values, edges).astype(np.int64)

    time_vars = ("temperature",)

    def time_coord(self, ds):
        return self.days

    def n_slots(self, ds):
        return len(self.days)

    # shards and slabs are whole calendar days of the shifted axis
    def shard_edges(self, T, n):
        from .distributed import time_partition

        return [int(self.day_ptr[d]) for d in time_partition(len(self.day_ptr) - 1, n)]

    def slab_edges(self, T, steps):
        ptr, edges, a = self.day_ptr, [], 0
        while a < len(ptr) - 1:
            b = a + 1
            while b < len(ptr) - 1 and ptr[b + 1] - ptr[a] <= steps:
                b += 1
            edges.append((int(ptr[a]), int(ptr[b])))
            a = b
        return [e for e in edges if e[1] > e[0]] or ([(0, T)] if T else [])

    def for_slab(self, t0, t1):
        import copy

        d0, d1 = self.out_slots(t0, t1)
        sub = copy.copy(self)
        sub.day_ptr = self.day_ptr[d0 : d1 + 1] - t0
        sub.days = self.days[d0:d1]
        sub._d_day_ptr = None
        return sub

    def out_slots(self, t0, t1):
        # groups whose range lies in [t0, t1); empty groups at a slab boundary go to the earlier slab
        d0 = int(np.searchsorted(self.day_ptr[:-1], t0, side="left"))
        d1 = int(np.searchsorted(self.day_ptr[1:], t1, side="right"))
        return d0, d1

    def run(self, ctx, ds, plan, time_agg, out=None):
        T, S = len(ds.coords["time"]), len(ds.coords["y"]) * len(ds.coords["x"])
        return ctx.heat_demand(ds.device(ctx, "temperature"), self.day_ptr, self.threshold_K, self.a,
                               self.constant, T, S, plan=plan, time_agg=time_agg, cooling=self.cooling, out=out)


class _CoolSpec(_HeatSpec):
    name = "cooling_demand"
    cooling = True


class _RunoffSpec(_Spec):
    name = "runoff"
    attrs = {}

    def __init__(self, ds, weight_with_height=True):
        _need(ds, ["runoff"], KeyError, "runoff")
        if weight_with_height:
            _need(ds, ["height"], KeyError, "height")
        self.weight_with_height = weight_with_height

    time_vars = ("runoff",)

    @property
    def static_vars(self):
        return ("height",) if self.weight_with_height else ()

    def run(self, ctx, ds, plan, time_agg, out=None):
        T, S = len(ds.coords["time"]), len(ds.coords["y"]) * len(ds.coords["x"])
        h = ds.device(ctx, "height") if self.weight_with_height else None
        return ctx.runoff(ds.device(ctx, "runoff"), h, T, S, plan=plan, time_agg=time_agg, out=out)


def _execute(ctx, spec, ds, plan, time_agg):
    """Whole-dataset launch, or the slab pipeline for large host-resident inputs."""
    from . import streaming

    if streaming.wanted(ds, spec) and (plan is not None or time_agg in (None, "sum")):
        out = streaming.run(ctx, spec, ds, plan, None if plan is not None else time_agg)
        if plan is not None and time_agg is not None:  # small (N, T) result: reduce on the host
            la = LabeledArray(out.numpy(), ("i", "time"))
            return ctx.upload(_aggregate_time(la, time_agg).values)
        return out
    return spec.run(ctx, ds, plan, time_agg)


def _per_cell(spec, ds):
    """Run a spec without aggregation and wrap the (time, y, x) result (device-resident)."""
    ctx = default_context()
    out = _execute(ctx, spec, ds, None, None)
    Y, X = len(ds.coords["y"]), len(ds.coords["x"])
    tc = spec.time_coord(ds)
    return LabeledArray(out.reshape(len(tc), Y, X), ("time", "y", "x"),
                        {"time": tc, "y": ds.coords["y"], "x": ds.coords["x"]}, dict(spec.attrs), spec.name)


def convert_pv(ds, panel, orientation, tracking, trigon_model="simple", clearsky_model="simple"):
    """convert.py:840-854; returns the per-cell 'specific generation' cube (kWh/kWp)."""
    return _per_cell(_PvSpec(ds, panel, orientation, tracking, trigon_model, clearsky_model), ds)


def convert_irradiation(ds, orientation, tracking=None, irradiation="total", trigon_model="simple",
                        clearsky_model="simple"):
    """convert.py:748-767; irradiation on the tilted surface (W m**-2)."""
    return _per_cell(_IrradiationSpec(ds, orientation, tracking, irradiation, trigon_model, clearsky_model), ds)


def convert_solar_thermal(ds, orientation, trigon_model, clearsky_model, c0, c1, t_store):
    """convert.py:550-574; solar thermal collector output."""
    return _per_cell(_SolarThermalSpec(ds, orientation, trigon_model, clearsky_model, c0, c1, t_store), ds)


def convert_wind(ds, turbine, interpolation_method):
    """convert.py:634-662; per-cell 'specific generation' (MWh/MWp)."""
    return _per_cell(_WindSpec(ds, turbine, interpolation_method), ds)


def convert_heat_demand(ds, threshold, a, constant, hour_shift):
    """convert.py:405-418; per-cell daily 'heat_demand'."""
    return _per_cell(_HeatSpec(ds, threshold, a, constant, hour_shift), ds)


def convert_cooling_demand(ds, threshold, a, constant, hour_shift):
    """convert.py:475-490; per-cell daily 'cooling_demand'."""
    return _per_cell(_CoolSpec(ds, threshold, a, constant, hour_shift), ds)


def convert_temperature(ds):
    """convert.py:292-299."""
    return _per_cell(_ThermoSpec(ds, "temperature", name="temperature"), ds)


def convert_soil_temperature(ds):
    """convert.py:307-318 (NaN over sea -> 0 so that it does not contribute to the aggregation)."""
    return _per_cell(_ThermoSpec(ds, "soil temperature", fillna0=True, name="soil temperature"), ds)


def convert_dewpoint_temperature(ds):
    """convert.py:326-330."""
    return _per_cell(_ThermoSpec(ds, "dewpoint temperature", name="dewpoint temperature"), ds)


def convert_coefficient_of_performance(ds, source, sink_T, c0, c1, c2):
    """convert.py:338-364."""
    return _per_cell(_cop_spec(ds, source, sink_T, c0, c1, c2), ds)


def convert_runoff(ds, weight_with_height=True):
    """convert.py:1028-1034."""
    return _per_cell(_RunoffSpec(ds, weight_with_height), ds)


_KNOWN = {convert_pv: _PvSpec, convert_wind: _WindSpec, convert_heat_demand: _HeatSpec,
          convert_runoff: _RunoffSpec, convert_irradiation: _IrradiationSpec,
          convert_solar_thermal: _SolarThermalSpec, convert_cooling_demand: _CoolSpec,
          convert_temperature: lambda ds: _ThermoSpec(ds, "temperature", name="temperature"),
          convert_soil_temperature: lambda ds: _ThermoSpec(ds, "soil temperature", fillna0=True, name="soil temperature"),
          convert_dewpoint_temperature: lambda ds: _ThermoSpec(ds, "dewpoint temperature", name="dewpoint temperature"),
          convert_coefficient_of_performance: _cop_spec}


class _CubeSpec(_Spec):
    """Result of an arbitrary user ``convert_func``: an already converted (time, y, x) cube."""

    def __init__(self, da, ds):
        if labeled.xr is not None and isinstance(da, labeled.xr.DataArray):
            da = LabeledArray(da.transpose("time", "y", "x").values, ("time", "y", "x"),
                              {"time": da.coords["time"].values}, dict(da.attrs), da.name)
        if not isinstance(da, LabeledArray):
            da = LabeledArray(np.asarray(da), ("time", "y", "x"), {"time": ds.coords["time"]})
        if da.dims != ("time", "y", "x"):
            vals = da.data if isinstance(da.data, DeviceArray) else None
            if vals is None:
                da = da.transpose("time", "y", "x")
            else:
                raise ValueError("device-resident results must have dims (time, y, x)")
        self.da = da
        self.name, self.attrs = da.name, dict(da.attrs)

    def time_coord(self, ds):
        return pd.DatetimeIndex(self.da.coords["time"]) if "time" in self.da.coords else ds.coords["time"]

    def run(self, ctx, ds, plan, time_agg, out=None):
        d = ctx.asdevice(self.da.data)
        T = d.shape[0]
        d = d.reshape(T, -1)
        if plan is None:
            if time_agg is None:
                return d
            # identity conversion + time reduction on the device (runoff kernel without height)
            return ctx.runoff(d, None, T, d.shape[1], plan=None, time_agg=time_agg)
        return ctx.spmm(plan, d, time_agg=time_agg, out=out)


# --------------------------------------------------------------------------------------
# gateway
# --------------------------------------------------------------------------------------
def _aggregate_time(da, method):
    """convert.py:51-56."""
    if method == "sum":
        return da.sum("time", keep_attrs=True)
    elif method == "mean":
        return da.mean("time", keep_attrs=True)
    return da


def _index_coords(index):
    """utils.ensure_coords (atlite/utils.py:22-36): -> (dim name, coordinate values)."""
    if labeled.xr is not None and isinstance(index, labeled.xr.Coordinates):
        dims = list(index.dims)
        if len(dims) > 1:
            raise ValueError(f"index must have a single dimension, not: {tuple(dims)}")
        return dims[0], index[dims[0]].values
    if isinstance(index, pd.Index):
        return index.name or "dim_0", index
    raise ValueError(f"index must be a pandas index or xarray coordinates, not: {index}")


def _as_dataset(data):
    if isinstance(data, Dataset):
        return data
    if labeled.xr is not None and isinstance(data, labeled.xr.Dataset):
        return Dataset.from_xarray(data)
    raise TypeError(f"cutout.data must be an atlite_amd.Dataset (or an xarray.Dataset), not {type(data)}")


def _finish(la):
    """Return a real xarray.DataArray when xarray is available, else the LabeledArray."""
    if labeled.xr is not None:
        return la.to_xarray()
    return la


def convert_and_aggregate(
    cutout,
    convert_func,
    matrix=None,
    index=None,
    layout=None,
    shapes=None,
    shapes_crs=4326,
    per_unit=False,
    return_capacity=False,
    aggregate_time="legacy",
    capacity_factor=False,
    capacity_factor_timeseries=False,
    show_progress=False,
    dask_kwargs={},
    **convert_kwds,
):
    """
    Convert and aggregate a weather-based renewable generation time-series on the GPU.

    Same parameters and return values as atlite's gateway (convert.py:59-159):
    ``matrix`` (N x S sparse / dense / labelled), ``index``, ``layout`` ((y, x) labelled array),
    ``shapes`` (+ ``shapes_crs``), ``per_unit``, ``return_capacity``, ``aggregate_time`` in
    {"sum", "mean", "legacy", None}, deprecated ``capacity_factor`` /
    ``capacity_factor_timeseries``; ``show_progress`` and ``dask_kwargs`` are accepted and
    ignored (there is no dask graph to execute).  ``**convert_kwds`` go to ``convert_func``.
    """
    if aggregate_time not in ("sum", "mean", "legacy", None):
        raise ValueError(f"aggregate_time must be 'sum', 'mean', 'legacy', or None, got {aggregate_time!r}")

    if aggregate_time == "legacy":
        warnings.warn(
            "aggregate_time='legacy' is deprecated and will be removed in a "
            "future release. Pass 'sum', 'mean', or None explicitly.",
            FutureWarning,
            stacklevel=2,
        )

    if capacity_factor or capacity_factor_timeseries:
        if aggregate_time != "legacy":
            raise ValueError(
                "Cannot use 'aggregate_time' together with deprecated "
                "'capacity_factor' or 'capacity_factor_timeseries'."
            )
        if capacity_factor:
            warnings.warn("capacity_factor is deprecated. Use aggregate_time='mean' instead.", FutureWarning,
                          stacklevel=2)
            aggregate_time = "mean"
        if capacity_factor_timeseries:
            warnings.warn("capacity_factor_timeseries is deprecated. Use aggregate_time=None instead.",
                          FutureWarning, stacklevel=2)
            aggregate_time = None

    # private: a callable (ctx, (N, T) DeviceArray, time coordinate) -> DeviceArray applied to the aggregated series while
    # it is still in HBM (runoff()'s post-processing, convert.py:1046-1082); only honoured where the (dim, time) series is
    # the call's result as it leaves the device
    device_post = convert_kwds.pop("_device_post", None)
    func_name = convert_func.__name__.replace("convert_", "")
    logger.info(f"Convert and aggregate '{func_name}'.")
    ds = _as_dataset(cutout.data)
    if convert_func in _KNOWN:
        spec = _KNOWN[convert_func](ds, **convert_kwds)
    else:
        spec = _CubeSpec(convert_func(cutout.data, **convert_kwds), ds)

    ctx = default_context()
    Y, X = len(ds.coords["y"]), len(ds.coords["x"])
    no_args = all(v is None for v in [layout, shapes, matrix])
    # more than one device selected (ATLITE_HIP_DEVICES / set_devices / Cutout(devices=)): the time axis
    # is sharded across them inside this call (atlite_amd.multigpu); results come back on the host
    from . import multigpu

    devs = multigpu.devices_for(cutout)
    multi = devs is not None and len(devs) > 1 and not isinstance(spec, _CubeSpec)

    if no_args:
        if per_unit or return_capacity:
            raise ValueError("One of `matrix`, `shapes` and `layout` must be given for `per_unit` or `return_capacity`")
        agg = "sum" if aggregate_time == "legacy" else aggregate_time
        out = multigpu.group(devs).run(spec, ds, None, X, agg) if multi else _execute(ctx, spec, ds, None, agg)
        if multi and agg is not None:
            out = ctx.upload(out)  # (S,) - keeps the single-device code below unchanged
        if agg is None:
            tc = spec.time_coord(ds)
            res = LabeledArray(out.reshape(len(tc), Y, X), ("time", "y", "x"),
                               {"time": tc, "y": ds.coords["y"], "x": ds.coords["x"]}, dict(spec.attrs), spec.name)
        else:
            res = LabeledArray(out.numpy().reshape(Y, X), ("y", "x"), {"y": ds.coords["y"], "x": ds.coords["x"]},
                               dict(spec.attrs), spec.name)
        # the reference returns a computed DataArray here too (convert.py:200-211); without xarray the
        # series stays a LabeledArray over device memory (copied to the host on first .values)
        return _finish(res)

    if matrix is not None:
        if shapes is not None:
            raise ValueError("Passing matrix and shapes is ambiguous. Pass only one of them.")
        is_xr = labeled.xr is not None and isinstance(matrix, labeled.xr.DataArray)
        if isinstance(matrix, LabeledArray) or is_xr:
            sdim = matrix.dims[1]
            mx, my = np.asarray(matrix.coords["x"]), np.asarray(matrix.coords["y"])
            if is_xr:
                mx, my = matrix.coords["x"].values, matrix.coords["y"].values
            g = cutout.grid
            if not (np.array_equal(mx, g["x"].values) and np.array_equal(my, g["y"].values)):
                raise ValueError("Matrix spatial coordinates not aligned with cutout spatial coordinates.")
            if index is None:
                d0 = matrix.dims[0]
                c0 = matrix.coords[d0].values if is_xr else matrix.coords.get(d0, np.arange(matrix.shape[0]))
                index = pd.Index(np.asarray(c0), name=d0)
            del sdim
            matrix = np.asarray(matrix.values)
        if not matrix.ndim == 2:
            raise ValueError("Matrix not 2-dimensional.")
        matrix = sp.csr_matrix(matrix)

    if shapes is not None:
        # pandas Series / GeoSeries / GeoDataFrame-like: the result is labelled with their index (convert.py:236-238)
        if index is None and isinstance(getattr(shapes, "index", None), pd.Index):
            index = shapes.index
        try:  # the cutout's cached matrix itself (not modified below: products create new matrices)
            matrix = cutout.indicatormatrix(shapes, shapes_crs, _share=True)
        except TypeError:  # a duck-typed cutout without the private switch
            matrix = cutout.indicatormatrix(shapes, shapes_crs)
        if not sp.isspmatrix_csr(matrix):
            matrix = sp.csr_matrix(matrix)

    if layout is not None:
        is_xr = labeled.xr is not None and isinstance(layout, labeled.xr.DataArray)
        assert isinstance(layout, LabeledArray) or is_xr
        lay = _reindex_layout(layout, ds)
        if matrix is None:
            matrix = sp.csr_matrix(lay[None, :])
        else:
            matrix = sp.csr_matrix(matrix) * spdiag(lay)

    assert isinstance(matrix, sp.csr_matrix)
    if matrix.shape[1] != Y * X:
        raise ValueError(f"matrix has {matrix.shape[1]} columns but the cutout has {Y * X} grid cells")

    dim, index_vals = _index_coords(pd.RangeIndex(matrix.shape[0]) if index is None else index)

    # per-unit needs the series on the host anyway (fillna(0) precedes the time reduction)
    on_device_time = aggregate_time if (aggregate_time in ("sum", "mean") and not per_unit and not multi) else None
    if multi:  # every device aggregates its time shard; the (N x T) series is gathered (RCCL) to the host
        out = multigpu.group(devs).run(spec, ds, matrix, X, None)
    else:
        # (the slot stride of the dataset's device copies only steers the plan's tile shape)
        out = None
        if getattr(spec, "aligned_ok", False) and _aligned_plan_wanted(ds, matrix, Y * X):
            # the caller's own contiguous device cubes on a grid whose slots do not start on 128-byte lines: the
            # line-aligned plan (atl_agg_create_aligned); conversions it does not cover answer with an error -> ordinary plan
            try:
                out = _execute(ctx, spec, ds, ctx.plan(matrix, row_len=X, aligned=True), on_device_time)
            except NotImplementedError:  # ATL_E_UNSUPPORTED: this converter / these cubes cannot be re-addressed
                out = None
        if out is None:
            plan = ctx.plan(matrix, row_len=X, ld=getattr(ds, "_slot_stride", lambda: None)())
            out = _execute(ctx, spec, ds, plan, on_device_time)  # the plan stays in ctx's cache
        if device_post is not None and on_device_time is None and not per_unit and aggregate_time in (None, "legacy"):
            out = device_post(ctx, out, spec.time_coord(ds))
            device_post.applied = True
        out = out.numpy()
    tc = spec.time_coord(ds)
    attrs = {}

    if per_unit or return_capacity:
        caps = np.asarray(matrix.sum(-1)).flatten()
        capacity = LabeledArray(caps, (dim,), {dim: index_vals}, {"units": "MW"})

    if on_device_time is not None:
        results = LabeledArray(out, (dim,), {dim: index_vals}, attrs)
    else:
        results = LabeledArray(out, (dim, "time"), {dim: index_vals, "time": tc}, attrs)
        if ds.chunked:  # dask branch of aggregate_matrix returns (time, dim)
            results = results.transpose("time", dim)

    if per_unit:
        with np.errstate(invalid="ignore", divide="ignore"):
            cap = np.where(caps != 0, caps, np.nan)
            shape = [1] * results.ndim
            shape[results.dims.index(dim)] = -1
            vals = results.values / cap.reshape(shape)
        results = LabeledArray(np.where(np.isnan(vals), 0.0, vals), results.dims, results.coords)
        results.attrs["units"] = "p.u."
    else:
        results.attrs["units"] = "MW"

    if aggregate_time != "legacy" and on_device_time is None:
        results = _aggregate_time(results, aggregate_time)

    if return_capacity:
        return _finish(results), _finish(capacity)
    return _finish(results)


def _aligned_plan_wanted(ds, matrix, S):
    """A line-aligned plan pays when the dataset's cubes are the caller's own contiguous device arrays and their slots do
    not start on 128-byte lines; 16 / gcd(S, 16) copies of the matrix must fit the plan's 65535 rows
    (``ATLITE_HIP_ALIGNED_PLANS=0`` switches it off)."""
    import math

    if os.environ.get("ATLITE_HIP_ALIGNED_PLANS", "1") == "0" or S % 16 == 0 or S < 16:
        return False
    if not getattr(ds, "_caller_layout", lambda: False)():
        return False
    return matrix.shape[0] * (16 // math.gcd(S, 16)) < 65536


def _reindex_layout(layout, ds):
    """layout.reindex_like(cutout.data).stack(spatial=["y","x"]) (convert.py:244) -> (S,) array."""
    if labeled.xr is not None and isinstance(layout, labeled.xr.DataArray):
        layout = LabeledArray(layout.values, layout.dims, {d: layout.coords[d].values for d in layout.dims})
    la = layout if layout.dims == ("y", "x") else layout.transpose("y", "x")
    vals = np.asarray(la.values, dtype=np.float64)
    for ax, d in enumerate(("y", "x")):
        have = pd.Index(np.asarray(la.coords[d])) if d in la.coords else pd.Index(ds.coords[d])
        want = pd.Index(ds.coords[d])
        if not have.equals(want):
            idx = have.get_indexer(want)
            taken = np.take(vals, np.where(idx < 0, 0, idx), axis=ax)
            mask = (idx < 0).reshape([-1 if i == ax else 1 for i in range(2)])
            vals = np.where(mask, np.nan, taken)
    return np.ascontiguousarray(vals).reshape(-1)


# --------------------------------------------------------------------------------------
# technology wrappers (signatures as in atlite/convert.py)
# --------------------------------------------------------------------------------------
def pv(cutout, panel, orientation, tracking=None, clearsky_model=None, **params):
    """Solar PV generation time-series (convert.py:857-936)."""
    if isinstance(panel, (str, Path)):
        panel = get_solarpanelconfig(panel)
    if not callable(orientation):
        orientation = get_orientation(orientation)
    return cutout.convert_and_aggregate(
        convert_func=convert_pv,
        panel=panel,
        orientation=orientation,
        tracking=tracking,
        clearsky_model=clearsky_model,
        **params,
    )


def irradiation(cutout, orientation, irradiation="total", tracking=None, clearsky_model=None, **params):
    """Total / direct / diffuse / ground irradiation on a tilted surface (convert.py:770-836)."""
    if not callable(orientation):
        orientation = get_orientation(orientation)
    return cutout.convert_and_aggregate(
        convert_func=convert_irradiation,
        orientation=orientation,
        tracking=tracking,
        irradiation=irradiation,
        clearsky_model=clearsky_model,
        **params,
    )


def solar_thermal(cutout, orientation={"slope": 45.0, "azimuth": 180.0}, trigon_model="simple",
                  clearsky_model="simple", c0=0.8, c1=3.0, t_store=80.0, **params):
    """Solar thermal collector time series (convert.py:577-631)."""
    if not callable(orientation):
        orientation = get_orientation(orientation)
    return cutout.convert_and_aggregate(
        convert_func=convert_solar_thermal,
        orientation=orientation,
        trigon_model=trigon_model,
        clearsky_model=clearsky_model,
        c0=c0,
        c1=c1,
        t_store=t_store,
        **params,
    )


def wind(cutout, turbine, smooth=False, add_cutout_windspeed=False, interpolation_method="logarithmic", **params):
    """Wind generation time-series (convert.py:665-744)."""
    turbine = get_windturbineconfig(turbine, add_cutout_windspeed=add_cutout_windspeed)
    if smooth:
        turbine = windturbine_smooth(turbine, params=smooth)
    return cutout.convert_and_aggregate(
        convert_func=convert_wind,
        turbine=turbine,
        interpolation_method=interpolation_method,
        **params,
    )


def heat_demand(cutout, threshold=15.0, a=1.0, constant=0.0, hour_shift=0.0, **params):
    """Daily heat demand by the degree-day approximation (convert.py:421-471)."""
    return cutout.convert_and_aggregate(
        convert_func=convert_heat_demand,
        threshold=threshold,
        a=a,
        constant=constant,
        hour_shift=hour_shift,
        **params,
    )


def cooling_demand(cutout, threshold=23.0, a=1.0, constant=0.0, hour_shift=0.0, **params):
    """Daily cooling demand by the degree-day approximation (convert.py:493-546)."""
    return cutout.convert_and_aggregate(
        convert_func=convert_cooling_demand,
        threshold=threshold,
        a=a,
        constant=constant,
        hour_shift=hour_shift,
        **params,
    )


def temperature(cutout, **params):
    """Outside temperature in deg C (convert.py:302-303)."""
    return cutout.convert_and_aggregate(convert_func=convert_temperature, **params)


def soil_temperature(cutout, **params):
    """Soil temperature in deg C, NaN (sea) -> 0 (convert.py:321-322)."""
    return cutout.convert_and_aggregate(convert_func=convert_soil_temperature, **params)


def dewpoint_temperature(cutout, **params):
    """Dewpoint temperature in deg C (convert.py:333-335)."""
    return cutout.convert_and_aggregate(convert_func=convert_dewpoint_temperature, **params)


def coefficient_of_performance(cutout, source="air", sink_T=55.0, c0=None, c1=None, c2=None, **params):
    """Heat-pump COP from air or soil temperature (convert.py:367-401)."""
    return cutout.convert_and_aggregate(
        convert_func=convert_coefficient_of_performance,
        source=source,
        sink_T=sink_T,
        c0=c0,
        c1=c1,
        c2=c2,
        **params,
    )


class _RunoffPost:
    """convert.py:1046-1082 on the (shapes x time) result WHILE IT IS ON THE DEVICE (ctx.rolling_mean / quantile / zero_below /
    normalize_rows -> atl_rolling_mean, atl_order_statistic, atl_zero_below, atl_normalize_rows): rolling mean over `smooth`
    steps (min_periods=1), values below the `lower_threshold_quantile` quantile of all values set to 0, scaling to reported
    yearly totals over the full years the series and `normalize_using_yearly` share.  Called by the gateway with the
    aggregated series before its download (`applied`), or by runoff() on an uploaded copy of a result that reached the
    host another way (per_unit, several devices, xarray in / out)."""

    def __init__(self, smooth, lower_threshold_quantile, normalize_using_yearly, dim_coords=None):
        self.smooth = 24 * 7 if smooth is True else smooth
        self.q = 5e-3 if lower_threshold_quantile is True else lower_threshold_quantile
        self.norm = normalize_using_yearly
        self.dim_coords = dim_coords  # labels of the rows (needed by the yearly normalisation); set by runoff()
        self.applied = False

    def __bool__(self):
        return self.smooth is not None or self.q is not None or self.norm is not None

    def yearly(self, tc):
        """-> (time mask (T,) of the full years shared with the reported totals, reported total per row label)."""
        nidx = self.norm.index
        nidx = nidx.year if isinstance(nidx, pd.DatetimeIndex) else nidx.astype(int)
        tyear = pd.Series(pd.to_datetime(tc).year)
        years = tyear.value_counts().loc[lambda x: x > 8700].index.intersection(nidx)
        assert len(years), "Need at least a full year of data (more is better)"
        lo, hi = min(years), max(years)
        ref = self.norm.copy()
        ref.index = nidx
        return ((tyear >= lo) & (tyear <= hi)).values, ref.loc[lo:hi].sum()

    def __call__(self, ctx, series, tc):
        if self.smooth is not None:
            series = ctx.rolling_mean(series, int(self.smooth), 1)
        if self.q is not None:
            series = ctx.zero_below(series, ctx.quantile(series, self.q))
        if self.norm is not None:
            tmask, ref = self.yearly(tc)
            series = ctx.normalize_rows(series, tmask, ref.reindex(pd.Index(self.dim_coords)).values)
        return series


def runoff(cutout, smooth=None, lower_threshold_quantile=None, normalize_using_yearly=None, **params):
    """
    Runoff (optionally height-weighted) aggregated to shapes, with the reference's post-processing of the small
    (shapes x time) result (convert.py:1037-1084) - on the device, before the result is downloaded.

    One deliberate deviation, in ``smooth``: a +-inf value in the series counts as MISSING in the rolling mean (what
    pandas' rolling does), where the reference's ``rolling(time=w, min_periods=1).mean()`` - xarray -> bottleneck
    ``move_mean`` - keeps the inf in the window's running sum: inf while it is inside, NaN for the rest of the row once
    it has left (inf - inf).  Runoff is finite; NaN is skipped in both.  The running sums restart every 256 steps with
    compensated additions, so finite results can differ from bottleneck's in the last bits (rtol 1e-10 holds).
    """
    post = _RunoffPost(smooth, lower_threshold_quantile, normalize_using_yearly)
    if post:
        # row labels of the result, for the yearly normalisation (the gateway's own rule: index, else the shapes' index)
        idx = params.get("index")
        if idx is None and isinstance(getattr(params.get("shapes"), "index", None), pd.Index):
            idx = params["shapes"].index
        if idx is None and hasattr(params.get("matrix"), "dims"):
            m = params["matrix"]
            idx = np.asarray(m.coords[m.dims[0]].values if hasattr(m.coords[m.dims[0]], "values") else m.coords[m.dims[0]])
        post.dim_coords = idx
        params = dict(params, _device_post=post)
    result = cutout.convert_and_aggregate(convert_func=convert_runoff, **params)
    if not post or post.applied:
        return result
    # the series reached the host another way (per_unit, a time reduction, several devices): the same device routines on a copy
    cap = None
    if "return_capacity" in params.keys() and isinstance(result, tuple):
        result, cap = result
    is_xr = labeled.xr is not None and isinstance(result, labeled.xr.DataArray)
    la = result if not is_xr else LabeledArray(result.values, result.dims,
                                               {d: result.coords[d].values for d in result.dims},
                                               dict(result.attrs), result.name)
    ax = la.get_axis_num("time")  # (a result without a time axis cannot be smoothed: the reference raises as well)
    # any rank: a per-cell (time, y, x) cube (no shapes) is smoothed along time per cell and thresholded on the quantile of
    # ALL its values, as the reference's rolling(time=...) / values.ravel() do (convert.py:1046-1062); time goes last, the
    # other axes are the rows of the device routines
    vals = np.moveaxis(np.asarray(la.values, dtype=np.float64), ax, -1)
    lead, T_ = vals.shape[:-1], vals.shape[-1]
    if post.norm is not None:
        if la.ndim != 2:
            raise ValueError("normalize_using_yearly needs a (shapes x time) result: pass shapes or a matrix")
        dim = la.dims[1 - ax]
        if post.dim_coords is None or len(post.dim_coords) != la.shape[1 - ax]:
            post.dim_coords = la.coords.get(dim, np.arange(la.shape[1 - ax]))
    ctx = default_context()
    out = post(ctx, ctx.upload(np.ascontiguousarray(vals.reshape(-1, T_))), la.coords["time"]).numpy()
    la = LabeledArray(np.moveaxis(out.reshape(lead + (T_,)), -1, ax), la.dims, la.coords, la.attrs, la.name)
    out = _finish(la) if is_xr or labeled.xr is not None else la
    return (out, cap) if cap is not None else out