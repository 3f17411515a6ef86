"""
Stand-in modules that let the reference's OWN hot-path source files
(/root/reference/atlite/{convert,aggregate,wind,resource,utils}.py and atlite/pv/*.py) execute
in an environment without xarray / dask / geopandas / rasterio / shapely / pyproj.

Only used by ``make_golden.py`` (in the build container, where /root/reference is mounted) to
freeze golden input/output vectors.  Nothing here is imported by the product or by the tests
that run on the GPU box.

* ``xarray``  -> a small eager DataArray / Dataset / Coordinates implementation with xarray's
  semantics for exactly the operations those files use (broadcasting by dimension name in
  first-appearance order, clip = np.clip, fillna, where, nan-skipping sum/mean,
  resample("1D").mean, stack/transpose/expand_dims/reindex_like, apply_ufunc).
* ``dask.array`` -> NumPy ufuncs (eager); ``dask.array.core.Array`` is a class nothing is an
  instance of, so ``aggregate_matrix`` takes its NumPy branch.
* GIS / IO packages -> inert stubs (their functions are never called on this path).
"""

from __future__ import annotations

import importlib
import importlib.abc
import importlib.machinery
import sys
import types

import numpy as np
import pandas as pd

REFERENCE = "/root/reference"

# ======================================================================================
# mini xarray
# ======================================================================================


def _is_da(x):
    return isinstance(x, DataArray)


class _Coords(dict):
    """Ordered mapping coordinate name -> DataArray (1-d index coords or aux coords)."""

    @property
    def dims(self):
        out = []
        for v in self.values():
            for d in v.dims:
                if d not in out:
                    out.append(d)
        return tuple(out)

    @property
    def sizes(self):
        out = {}
        for v in self.values():
            out.update(dict(zip(v.dims, v.shape)))
        return out


class Coordinates(_Coords):
    """xr.Coordinates({name: pandas.Index}) as used by atlite.utils.ensure_coords."""

    def __init__(self, mapping=None):
        super().__init__()
        for k, v in (mapping or {}).items():
            if _is_da(v):
                self[k] = v
            else:
                self[k] = DataArray(np.asarray(v), dims=[k], _index=True)

    def assign(self, **kw):
        new = Coordinates(dict(self))
        for k, v in kw.items():
            new[k] = v if _is_da(v) else DataArray(np.asarray(v), dims=[k], _index=True)
        return new


def _broadcast(a_vals, a_dims, out_dims):
    """View of a_vals with axes ordered/expanded to out_dims."""
    a_vals = np.asarray(a_vals)
    order = [a_dims.index(d) for d in out_dims if d in a_dims]
    v = np.transpose(a_vals, order) if order != list(range(len(order))) else a_vals
    shape = [v.shape[[d for d in out_dims if d in a_dims].index(d)] if d in a_dims else 1 for d in out_dims]
    return v.reshape(shape)


class _DT:
    def __init__(self, da):
        self._da = da

    def _field(self, name):
        idx = pd.DatetimeIndex(self._da.values.ravel())
        return DataArray(np.asarray(getattr(idx, name)).reshape(self._da.shape), dims=self._da.dims,
                         coords=self._da._coords, name=name)

    hour = property(lambda self: self._field("hour"))
    minute = property(lambda self: self._field("minute"))


class DataArray:
    __array_priority__ = 50

    def __init__(self, data=None, coords=None, dims=None, name=None, attrs=None, _index=False):
        if isinstance(data, pd.Series) and coords is None and dims is not None and len(dims) == 1:
            coords = {dims[0]: np.asarray(data.index)}  # xarray: a pandas object brings its index along
        data = data.values if _is_da(data) else np.asarray(data)
        if dims is None:
            if isinstance(coords, _Coords):
                dims = coords.dims
            elif isinstance(coords, dict):
                dims = tuple(coords.keys())
            else:
                dims = tuple(f"dim_{i}" for i in range(data.ndim))
        self.dims = tuple(dims)
        assert len(self.dims) == data.ndim, (self.dims, data.shape)
        self._values = data
        self.name = name
        self.attrs = dict(attrs or {})
        self._coords = _Coords()
        if coords is not None:
            for k, v in coords.items():
                if _is_da(v):
                    if all(d in self.dims for d in v.dims):
                        self._coords[k] = v
                elif isinstance(v, tuple) and len(v) == 2 and v[0] in self.dims:  # xarray's (dim, values) form
                    self._coords[k] = DataArray(np.asarray(v[1]), dims=[v[0]], _index=(k == v[0]))
                elif k in self.dims:
                    self._coords[k] = DataArray(np.asarray(v), dims=[k], _index=True)
        if _index:
            self._coords[self.dims[0]] = self

    # -- basic properties -------------------------------------------------------------------
    values = property(lambda self: self._values)
    data = property(lambda self: self._values)
    shape = property(lambda self: self._values.shape)
    ndim = property(lambda self: self._values.ndim)
    dtype = property(lambda self: self._values.dtype)
    sizes = property(lambda self: dict(zip(self.dims, self.shape)))
    coords = property(lambda self: self._coords)
    dt = property(lambda self: _DT(self))

    @property
    def indexes(self):
        return {d: pd.Index(self._coords[d].values) for d in self.dims if d in self._coords}

    def __array__(self, dtype=None, copy=None):
        return self._values if dtype is None else self._values.astype(dtype)

    def __len__(self):
        return self.shape[0]

    def __repr__(self):
        return f"<shim.DataArray {self.name!r} {self.sizes}>"

    def _new(self, values, dims=None, name="__same__", attrs=None, coords=None):
        dims = self.dims if dims is None else tuple(dims)
        c = self._coords if coords is None else coords
        c = _Coords({k: v for k, v in c.items() if all(d in dims for d in v.dims)})
        return DataArray(values, coords=c, dims=dims, name=self.name if name == "__same__" else name, attrs=attrs)

    # -- arithmetic with broadcasting by name -------------------------------------------------
    def _binary(self, other, op, reflexive=False):
        if isinstance(other, Dataset):
            return NotImplemented
        if _is_da(other):
            for d in self.dims:  # xarray aligns (inner join) on shared indexed dimensions
                if d in other.dims and d in self._coords and d in other._coords:
                    a_i, b_i = pd.Index(self._coords[d].values), pd.Index(other._coords[d].values)
                    if not a_i.equals(b_i):
                        common = a_i.intersection(b_i)
                        return self.reindex(**{d: common})._binary(other.reindex(**{d: common}), op, reflexive)
            out_dims = list(self.dims) + [d for d in other.dims if d not in self.dims]
            if reflexive:
                out_dims = list(other.dims) + [d for d in self.dims if d not in other.dims]
            a = _broadcast(self._values, list(self.dims), out_dims)
            b = _broadcast(other._values, list(other.dims), out_dims)
            coords = _Coords(dict(other._coords))
            coords.update(self._coords)
            name = self.name if self.name == other.name else None
        else:
            out_dims = list(self.dims)
            if isinstance(other, pd.Timedelta):
                other = other.to_timedelta64()
            a, b = self._values, other
            coords = self._coords
            name = self.name
        with np.errstate(all="ignore"):
            vals = op(b, a) if reflexive else op(a, b)
        return DataArray(vals, coords=_Coords({k: v for k, v in coords.items() if all(d in out_dims for d in v.dims)}),
                         dims=out_dims, name=name)

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if method != "__call__":
            return NotImplemented
        if len(inputs) == 1:
            with np.errstate(all="ignore"):
                return self._new(ufunc(self._values, **kwargs))
        if len(inputs) == 2:
            a, b = inputs
            if _is_da(a):
                return a._binary(b, lambda x, y: ufunc(x, y, **kwargs))
            return b._binary(a, lambda x, y: ufunc(x, y, **kwargs), reflexive=True)
        return NotImplemented

    def __neg__(self):
        return self._new(-self._values)

    def __invert__(self):
        return self._new(~self._values)

    def __abs__(self):
        return self._new(np.abs(self._values))

    def __bool__(self):
        return bool(self._values)

    def any(self):
        return self._new(np.any(self._values), dims=[])

    # -- xarray methods the reference uses ------------------------------------------------------
    def where(self, cond, other=np.nan):
        c = cond if _is_da(cond) else DataArray(np.asarray(cond), dims=self.dims)
        o = other if _is_da(other) else None
        out_dims = list(self.dims) + [d for d in c.dims if d not in self.dims]
        if o is not None:
            out_dims += [d for d in o.dims if d not in out_dims]
        a = _broadcast(self._values, list(self.dims), out_dims)
        cc = _broadcast(c._values, list(c.dims), out_dims)
        oo = _broadcast(o._values, list(o.dims), out_dims) if o is not None else other
        coords = _Coords(dict(c._coords))
        coords.update(self._coords)
        return DataArray(np.where(cc, a, oo), coords=coords, dims=out_dims, name=self.name, attrs=self.attrs)

    def fillna(self, value):
        v = value._values if _is_da(value) else value
        return self._new(np.where(np.isnan(self._values), v, self._values), attrs=self.attrs)

    def clip(self, min=None, max=None):
        out = self
        with np.errstate(all="ignore"):
            if min is not None:
                out = out._binary(min, np.maximum) if _is_da(min) else out._new(np.maximum(out._values, min))
                # np.clip semantics: NaN in the data propagates (np.maximum does that already)
            if max is not None:
                out = out._binary(max, np.minimum) if _is_da(max) else out._new(np.minimum(out._values, max))
        out.name = self.name
        out.attrs = dict(self.attrs)
        return out

    def rename(self, new):
        if isinstance(new, dict):
            dims = [new.get(d, d) for d in self.dims]
            coords = _Coords({new.get(k, k): v.rename(new) if k != self.name else v for k, v in self._coords.items()
                              if v is not self})
            out = DataArray(self._values, coords=coords, dims=dims, name=new.get(self.name, self.name),
                            attrs=self.attrs)
            return out
        return self._new(self._values, name=new, attrs=self.attrs)

    def transpose(self, *dims):
        dims = list(dims) if dims else list(self.dims[::-1])
        return self._new(np.transpose(self._values, [self.dims.index(d) for d in dims]), dims=dims, attrs=self.attrs)

    def chunk(self, *a, **k):
        return self

    def load(self, **k):
        return self

    def compute(self, **k):
        return self

    def assign_attrs(self, **kw):
        out = self._new(self._values, attrs=dict(self.attrs, **kw))
        return out

    def assign_coords(self, **kw):
        coords = _Coords(dict(self._coords))
        for k, v in kw.items():
            coords[k] = DataArray(v.values if _is_da(v) else np.asarray(v), dims=[k], _index=True)
        return DataArray(self._values, coords=coords, dims=self.dims, name=self.name, attrs=self.attrs)

    def _reduce(self, fn, dim, keep_attrs):
        ax = self.dims.index(dim)
        with np.errstate(all="ignore"):
            vals = fn(self._values, ax)
        return self._new(vals, dims=[d for d in self.dims if d != dim], attrs=self.attrs if keep_attrs else None)

    def sum(self, dim, keep_attrs=False):
        # xarray: skipna=True for floats -> nansum
        return self._reduce(lambda v, ax: np.nansum(v, axis=ax), dim, keep_attrs)

    def mean(self, dim, keep_attrs=False):
        # xarray nanops.nanmean: sum of NaN-zeroed values / count of valid
        def f(v, ax):
            valid = ~np.isnan(v)
            return np.sum(np.where(valid, v, 0.0), axis=ax) / np.sum(valid, axis=ax)

        return self._reduce(f, dim, keep_attrs)

    def stack(self, **kw):
        ((new, old),) = kw.items()
        old = list(old)
        keep = [d for d in self.dims if d not in old]
        v = np.transpose(self._values, [self.dims.index(d) for d in keep + old])
        v = v.reshape(v.shape[: len(keep)] + (-1,))
        return DataArray(v, coords=_Coords({k: c for k, c in self._coords.items() if all(d in keep for d in c.dims)}),
                         dims=keep + [new], name=self.name, attrs=self.attrs)

    def expand_dims(self, dim):
        return DataArray(self._values[None, ...], dims=[dim] + list(self.dims), name=self.name, attrs=self.attrs)

    def reindex_like(self, other):
        vals = self._values
        for ax, d in enumerate(self.dims):
            if d in other.coords:
                want = pd.Index(np.asarray(other.coords[d].values))
                have = pd.Index(np.asarray(self._coords[d].values))
                if not have.equals(want):
                    idx = have.get_indexer(want)
                    taken = np.take(vals, np.where(idx < 0, 0, idx), axis=ax).astype(float)
                    mask = (idx < 0).reshape([-1 if i == ax else 1 for i in range(vals.ndim)])
                    vals = np.where(mask, np.nan, taken)
        coords = _Coords({d: other.coords[d] for d in self.dims if d in other.coords})
        return DataArray(vals, coords=coords, dims=self.dims, name=self.name, attrs=self.attrs)

    def get_axis_num(self, dim):
        return self.dims.index(dim)

    def reindex(self, **kw):
        vals, coords = self._values, _Coords(dict(self._coords))
        for d, want in kw.items():
            ax = self.dims.index(d)
            want = pd.Index(want.values if _is_da(want) else np.asarray(want))
            have = pd.Index(np.asarray(self._coords[d].values))
            idx = have.get_indexer(want)
            taken = np.take(vals, np.where(idx < 0, 0, idx), axis=ax)
            if (idx < 0).any():
                mask = (idx < 0).reshape([-1 if i == ax else 1 for i in range(vals.ndim)])
                taken = np.where(mask, np.nan, taken.astype(float))
            vals = taken
            coords[d] = DataArray(np.asarray(want), dims=[d], _index=True)
        return DataArray(vals, coords=coords, dims=self.dims, name=self.name, attrs=self.attrs)

    def sel(self, **kw):
        out = self
        for d, key in kw.items():
            idx = pd.Index(np.asarray(out._coords[d].values))
            assert isinstance(key, slice), "the stand-in selects label slices only"
            if isinstance(idx, pd.DatetimeIndex) or np.issubdtype(np.asarray(idx).dtype, np.datetime64):
                idx = pd.DatetimeIndex(idx)
            pos = idx.slice_indexer(key.start, key.stop)  # label based, both ends inclusive, partial date strings
            ax = out.dims.index(d)
            vals = np.take(out._values, np.arange(len(idx))[pos], axis=ax)
            coords = _Coords(dict(out._coords))
            coords[d] = DataArray(np.asarray(idx)[pos], dims=[d], _index=True)
            out = DataArray(vals, coords=coords, dims=out.dims, name=out.name, attrs=out.attrs)
        return out

    def rolling(self, min_periods=None, center=False, **kw):
        ((dim, window),) = kw.items()
        assert not center
        return _Rolling(self, dim, int(window), min_periods)

    def resample(self, **kw):
        ((dim, freq),) = kw.items()
        assert freq == "1D"
        return _Resample(self, dim)

    def to_frame(self, *a, **k):
        raise NotImplementedError


def _mk_op(npop):
    def fwd(self, other):
        return self._binary(other, npop)

    def rev(self, other):
        return self._binary(other, npop, reflexive=True)

    return fwd, rev


for _n, _f in dict(add=np.add, sub=np.subtract, mul=np.multiply, truediv=np.true_divide, pow=(lambda a, b: a ** b),
                   mod=np.mod, and_=np.logical_and, or_=np.logical_or).items():
    _fw, _rv = _mk_op(_f)
    _name = _n.rstrip("_")
    setattr(DataArray, f"__{_name}__", _fw)
    setattr(DataArray, f"__r{_name}__", _rv)
for _n, _f in dict(lt=np.less, le=np.less_equal, gt=np.greater, ge=np.greater_equal, eq=np.equal,
                   ne=np.not_equal).items():
    setattr(DataArray, f"__{_n}__", _mk_op(_f)[0])
DataArray.__hash__ = object.__hash__


class _Rolling:
    """da.rolling(time=w, min_periods=m).mean(): trailing window, NaN-skipping, like pandas' rolling."""

    def __init__(self, da, dim, window, min_periods):
        self.da, self.dim, self.window = da, dim, window
        self.min_periods = window if min_periods is None else min_periods

    def mean(self):
        da = self.da
        ax = da.dims.index(self.dim)
        v = np.moveaxis(np.asarray(da.values, dtype=float), ax, 0)
        # xarray without bottleneck: NaN-padded window view along the dimension, then nanmean over the window
        pad = np.full((self.window - 1,) + v.shape[1:], np.nan)
        win = np.lib.stride_tricks.sliding_window_view(np.concatenate([pad, v], axis=0), self.window, axis=0)
        valid = ~np.isnan(win)
        cnt = valid.sum(axis=-1)
        with np.errstate(all="ignore"):
            out = np.where(valid, win, 0.0).sum(axis=-1) / cnt
        out = np.where(cnt >= self.min_periods, out, np.nan)
        return da._new(np.moveaxis(out, 0, ax))


class _Resample:
    """DataArray.resample(time="1D"): calendar-day bins from the first to the last day."""

    def __init__(self, da, dim):
        self.da, self.dim = da, dim

    def mean(self, dim=None):
        da = self.da
        ax = da.dims.index(self.dim)
        t = pd.DatetimeIndex(da.coords[self.dim].values)
        day = t.floor("D")
        labels = pd.date_range(day[0], day[-1], freq="D")
        out = np.full(da.shape[:ax] + (len(labels),) + da.shape[ax + 1 :], np.nan)
        v = np.moveaxis(da.values, ax, 0)
        o = np.moveaxis(out, ax, 0)
        with np.errstate(all="ignore"):
            for i, lab in enumerate(labels):
                sel = np.flatnonzero(day == lab)
                if len(sel):
                    blk = v[sel]
                    valid = ~np.isnan(blk)
                    o[i] = np.sum(np.where(valid, blk, 0.0), axis=0) / np.sum(valid, axis=0)
        coords = _Coords({k: c for k, c in da.coords.items() if k != self.dim})
        coords[self.dim] = DataArray(labels.values, dims=[self.dim], _index=True)
        return DataArray(out, coords=coords, dims=da.dims, name=da.name)


class Dataset:
    def __init__(self, data_vars=None, coords=None, attrs=None):
        self._coords = _Coords()
        for k, v in (coords or {}).items():
            if _is_da(v):
                self._coords[k] = v
            elif isinstance(v, tuple):
                self._coords[k] = DataArray(np.asarray(v[1]), dims=[v[0]] if isinstance(v[0], str) else list(v[0]))
            else:
                self._coords[k] = DataArray(np.asarray(v), dims=[k], _index=True)
        self._vars = {}
        self.attrs = dict(attrs or {})
        for k, v in (data_vars or {}).items():
            self[k] = v

    def __setitem__(self, k, v):
        if not _is_da(v):
            v = DataArray(np.asarray(v), dims=[])
        for ck, cv in v.coords.items():
            if ck not in self._coords and cv is not v:
                self._coords[ck] = cv
        self._vars[k] = v

    def __getitem__(self, key):
        if isinstance(key, (set, list, tuple)):
            return Dataset({k: self._vars[k] for k in key}, coords=self._coords, attrs=self.attrs)
        if key in self._vars:
            v = self._vars[key]
            coords = _Coords({k: c for k, c in self._coords.items() if all(d in v.dims for d in c.dims)})
            return DataArray(v.values, coords=coords, dims=v.dims, name=key, attrs=v.attrs)
        if key in self._coords:
            c = self._coords[key]
            coords = _Coords({k: cc for k, cc in self._coords.items() if all(d in c.dims for d in cc.dims)})
            return DataArray(c.values, coords=coords, dims=c.dims, name=key, attrs=c.attrs)
        raise KeyError(key)

    def __contains__(self, key):
        return key in self._vars or key in self._coords

    def __iter__(self):
        return iter(self._vars)

    data_vars = property(lambda self: self._vars)
    coords = property(lambda self: self._coords)
    chunksizes = property(lambda self: {})

    @property
    def indexes(self):
        return {k: pd.Index(v.values) for k, v in self._coords.items() if v.dims == (k,)}

    @property
    def sizes(self):
        out = dict(self._coords.sizes)
        for v in self._vars.values():
            out.update(v.sizes)
        return out

    def __getattr__(self, key):
        if key.startswith("_"):
            raise AttributeError(key)
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key)

    def rename(self, mapping):
        return Dataset({mapping.get(k, k): v for k, v in self._vars.items()}, coords=self._coords, attrs=self.attrs)

    def load(self, **k):
        return self

    def keys(self):
        return self._vars.keys()


def apply_ufunc(func, *args, input_core_dims=None, output_core_dims=None, output_dtypes=None, dask=None,
                dask_gufunc_kwargs=None, **kw):
    (a,) = args
    return a._new(func(a.values))


def date_range(start, periods=None, freq=None, **kw):
    return pd.date_range(start, periods=periods, freq=freq, **kw)


def _assert_identical(a, b):
    assert type(a) is type(b), (type(a), type(b))
    assert a.dims == b.dims and a.name == b.name, (a.dims, b.dims, a.name, b.name)
    np.testing.assert_array_equal(a.values, b.values)
    assert a.attrs == b.attrs, (a.attrs, b.attrs)


def _make_xarray():
    m = types.ModuleType("xarray")
    m.DataArray, m.Dataset, m.Coordinates = DataArray, Dataset, Coordinates
    m.apply_ufunc, m.date_range = apply_ufunc, date_range
    m.testing = types.SimpleNamespace(assert_identical=_assert_identical,
                                      assert_allclose=lambda a, b, **k: np.testing.assert_allclose(a.values, b.values, **k))
    m.__version__ = "0.0.shim"
    return m


# ======================================================================================
# dask + inert stubs
# ======================================================================================
class _Inert:
    """Attribute sink: any attribute / call returns another inert object."""

    def __init__(self, name="inert"):
        self._n = name

    def __getattr__(self, k):
        if k.startswith("__") and k.endswith("__"):
            raise AttributeError(k)
        return _Inert(f"{self._n}.{k}")

    def __call__(self, *a, **k):
        return _Inert(f"{self._n}()")

    def __mro_entries__(self, bases):
        return (object,)


class _StubModule(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__") and k.endswith("__"):
            raise AttributeError(k)
        return _Inert(f"{self.__name__}.{k}")


_STUB_ROOTS = ("geopandas", "rasterio", "pyproj", "shapely", "cdsapi", "cfgrib", "netCDF4", "numexpr", "bottleneck",
               "toolz", "progressbar")


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path=None, target=None):
        if name.split(".")[0] in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def _make_dask():
    dask = types.ModuleType("dask")
    dask.__path__ = []
    dask.compute = lambda *a, **k: a
    dask.delayed = lambda f=None, **k: f
    da = types.ModuleType("dask.array")
    da.__path__ = []
    for n in ("sin", "cos", "tan", "arcsin", "arccos", "arctan", "arctan2", "radians", "degrees", "sqrt", "fmin",
              "fmax", "absolute", "maximum", "minimum", "mod", "logical_and", "logical_or", "exp", "log", "where"):
        setattr(da, n, getattr(np, n))
    core = types.ModuleType("dask.array.core")

    class Array:  # nothing is ever an instance: aggregate_matrix takes the NumPy branch
        pass

    core.Array = Array
    da.core = core
    da.Array = Array
    diag = types.ModuleType("dask.diagnostics")

    class ProgressBar:
        def __init__(self, *a, **k):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    diag.ProgressBar = ProgressBar
    utils = types.ModuleType("dask.utils")
    utils.SerializableLock = _Inert("SerializableLock")
    dask.array, dask.diagnostics, dask.utils = da, diag, utils
    return {"dask": dask, "dask.array": da, "dask.array.core": core, "dask.diagnostics": diag, "dask.utils": utils}


_installed = False


def install():
    """Register the stand-ins and an empty ``atlite`` package rooted at the reference sources."""
    global _installed
    if _installed:
        return
    sys.modules["xarray"] = _make_xarray()
    sys.modules.update(_make_dask())
    sys.meta_path.insert(0, _StubFinder())
    # bypass atlite/__init__.py (it imports Cutout -> rasterio/geopandas at module level)
    pkg = types.ModuleType("atlite")
    pkg.__path__ = [f"{REFERENCE}/atlite"]
    sys.modules["atlite"] = pkg
    dsets = types.ModuleType("atlite.datasets")  # utils.py:16 only needs the name
    dsets.modules = {}
    dsets.__path__ = []
    sys.modules["atlite.datasets"] = dsets
    _installed = True


def reference(module):
    """Import a reference module by dotted name, e.g. reference('atlite.convert')."""
    install()
    return importlib.import_module(module)
