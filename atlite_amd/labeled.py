"""
Minimal labelled-array containers.

The reference passes ``xarray`` objects across the hot-path boundary (``cutout.data`` is an
``xr.Dataset``, results are ``xr.DataArray``; atlite/convert.py:59-276).  xarray is not part
of this image, so the host layer works on two small stand-ins that carry exactly what the
path needs - values (host ``numpy`` or device ``DeviceArray``), dimension names, coordinates,
attrs, name - and converts to / from real xarray objects when xarray is importable
(``LabeledArray.to_xarray``, ``Dataset.from_xarray``).
"""

from __future__ import annotations

import os

import numpy as np
import pandas as pd

try:  # optional
    import xarray as xr
except Exception:  # pragma: no cover - xarray is absent from the build image
    xr = None


def _is_device(x):
    return type(x).__name__ == "DeviceArray"


def _is_lazy(x):
    """File-backed variable (atlite_amd.io.FileArray): kept as is, read on demand."""
    return getattr(x, "is_file_array", False)


class LabeledArray:
    """``values`` + ``dims`` + ``coords`` + ``attrs`` + ``name`` (a tiny DataArray stand-in)."""

    def __init__(self, values, dims, coords=None, attrs=None, name=None):
        self._values = values if (_is_device(values) or _is_lazy(values)) else np.asarray(values)
        self.dims = tuple(dims)
        if len(self.dims) != len(self._values.shape):
            raise ValueError(f"dims {self.dims} do not match shape {self._values.shape}")
        self.coords = {}
        for k, v in (coords or {}).items():
            self.coords[k] = v if isinstance(v, pd.Index) else np.asarray(v)
        self.attrs = dict(attrs or {})
        self.name = name

    # -- data access ----------------------------------------------------------------------
    @property
    def data(self):
        """Underlying storage: numpy array or DeviceArray."""
        return self._values

    @property
    def values(self):
        if _is_device(self._values):
            return self._values.numpy()
        if _is_lazy(self._values):
            return np.asarray(self._values)
        return self._values

    def __array__(self, dtype=None, copy=None):
        v = self.values
        return v if dtype is None else v.astype(dtype)

    @property
    def shape(self):
        return tuple(self._values.shape)

    @property
    def ndim(self):
        return len(self.dims)

    @property
    def sizes(self):
        return dict(zip(self.dims, self.shape))

    @property
    def dtype(self):
        return self._values.dtype

    def get_axis_num(self, dim):
        return self.dims.index(dim)

    # -- light algebra (host, small results only) -------------------------------------------
    def _like(self, values, dims=None, drop=()):
        dims = self.dims if dims is None else tuple(dims)
        coords = {k: v for k, v in self.coords.items() if k in dims and k not in drop}
        return LabeledArray(values, dims, coords, self.attrs, self.name)

    def transpose(self, *dims):
        if not dims:
            dims = self.dims[::-1]
        order = [self.dims.index(d) for d in dims]
        return self._like(np.transpose(self.values, order), dims)

    def _reduce(self, fn, dim):
        ax = self.dims.index(dim)
        with np.errstate(invalid="ignore", divide="ignore"):
            vals = fn(self.values, ax)
        return self._like(vals, [d for d in self.dims if d != dim], drop=(dim,))

    def sum(self, dim, keep_attrs=True):
        return self._reduce(lambda v, ax: np.nansum(v, axis=ax), dim)

    def mean(self, dim, keep_attrs=True):
        def f(v, ax):
            cnt = np.sum(~np.isnan(v), axis=ax)
            return np.nansum(v, axis=ax) / cnt

        return self._reduce(f, dim)

    def isel(self, **idx):
        vals, dims, coords = self.values, list(self.dims), dict(self.coords)
        for d, i in idx.items():
            ax = dims.index(d)
            vals = np.take(vals, i, axis=ax)
            if np.ndim(i) == 0:
                dims.pop(ax)
                coords.pop(d, None)
            elif d in coords:
                coords[d] = np.asarray(coords[d])[i]
        return LabeledArray(vals, dims, coords, self.attrs, self.name)

    def rename(self, name):
        out = self._like(self._values)
        out.name = name
        return out

    def sel(self, **labels):
        """Label-based selection along dimensions that carry a coordinate: a label, a list of labels or a slice of
        labels (both ends included), like ``xarray.DataArray.sel``."""
        pos = {}
        for d, lab in labels.items():
            if d not in self.coords:
                raise KeyError(f"no coordinate along {d!r}")
            idx = pd.Index(self.coords[d])
            if isinstance(lab, slice):
                lo, hi = idx.slice_locs(lab.start, lab.stop)
                pos[d] = np.arange(lo, hi)
            elif np.ndim(lab) == 0 or isinstance(lab, str):
                loc = idx.get_loc(lab)
                pos[d] = int(loc) if isinstance(loc, (int, np.integer)) else (
                    np.arange(loc.start, loc.stop) if isinstance(loc, slice) else np.flatnonzero(loc))
            else:
                loc = idx.get_indexer(pd.Index(lab))
                if (loc < 0).any():
                    raise KeyError(f"not all values found in index {d!r}")
                pos[d] = loc
        return self.isel(**pos)

    # elementwise arithmetic on the (small, host) results: operands are aligned by dimension NAME, new dimensions are
    # appended in order of first appearance, attributes are dropped - xarray's rules for the cases that arise here
    def _binary(self, other, op, reflexive=False):
        a = self.values
        if isinstance(other, LabeledArray):
            dims = list(self.dims) + [d for d in other.dims if d not in self.dims]

            def expand(la):
                v = la.values
                order = [la.dims.index(d) for d in dims if d in la.dims]
                v = np.transpose(v, order)
                return v.reshape([v.shape[[d for d in dims if d in la.dims].index(d)] if d in la.dims else 1 for d in dims])

            for d in set(self.dims) & set(other.dims):
                if self.sizes[d] != other.sizes[d]:
                    raise ValueError(f"operands disagree along {d!r}: {self.sizes[d]} vs {other.sizes[d]}")
            a, b = expand(self), expand(other)
            coords = {**{k: v for k, v in other.coords.items() if k in dims}, **{k: v for k, v in self.coords.items() if k in dims}}
            name = self.name if self.name == other.name else None
        else:
            dims, b, coords, name = self.dims, (other.numpy() if _is_device(other) else other), self.coords, self.name
        with np.errstate(all="ignore"):
            out = op(b, a) if reflexive else op(a, b)
        return LabeledArray(out, dims, {k: v for k, v in coords.items() if k in dims}, None, name)

    def __neg__(self):
        return LabeledArray(-self.values, self.dims, self.coords, None, self.name)

    def __abs__(self):
        return LabeledArray(np.abs(self.values), self.dims, self.coords, None, self.name)

    def max(self, dim=None):
        return np.nanmax(self.values) if dim is None else self._reduce(lambda v, ax: np.nanmax(v, axis=ax), dim)

    def min(self, dim=None):
        return np.nanmin(self.values) if dim is None else self._reduce(lambda v, ax: np.nanmin(v, axis=ax), dim)

    def to_pandas(self):
        v = self.values
        if self.ndim == 1:
            return pd.Series(v, index=pd.Index(self.coords.get(self.dims[0], np.arange(len(v))), name=self.dims[0]),
                             name=self.name)
        if self.ndim == 2:
            return pd.DataFrame(
                v,
                index=pd.Index(self.coords.get(self.dims[0], np.arange(v.shape[0])), name=self.dims[0]),
                columns=pd.Index(self.coords.get(self.dims[1], np.arange(v.shape[1])), name=self.dims[1]),
            )
        raise ValueError("to_pandas supports 1-d and 2-d arrays")

    def to_xarray(self):
        if xr is None:
            raise ImportError("xarray is not installed")
        return xr.DataArray(self.values, dims=self.dims, coords={k: (k, np.asarray(v)) for k, v in self.coords.items()},
                            attrs=self.attrs, name=self.name)

    def __repr__(self):
        where = "device" if _is_device(self._values) else "file" if _is_lazy(self._values) else "host"
        return f"<LabeledArray {self.name!r} {self.sizes} [{where}] attrs={self.attrs}>"


def _install_operators():
    import operator as _op

    for name, fn in (("add", _op.add), ("sub", _op.sub), ("mul", _op.mul), ("truediv", _op.truediv), ("pow", _op.pow),
                     ("lt", _op.lt), ("le", _op.le), ("gt", _op.gt), ("ge", _op.ge)):
        setattr(LabeledArray, f"__{name}__", lambda self, other, fn=fn: self._binary(other, fn))
        if name in ("add", "sub", "mul", "truediv", "pow"):
            setattr(LabeledArray, f"__r{name}__", lambda self, other, fn=fn: self._binary(other, fn, reflexive=True))
    LabeledArray.__array_priority__ = 1000  # ndarray <op> LabeledArray defers to the methods above


_install_operators()


class Dataset:
    """
    Dict of (time, y, x) / (y, x) variables sharing coordinates - what the hot path reads from
    ``cutout.data`` (variable names as in atlite/datasets/era5.py:47-60).

    ``chunked=True`` marks a dataset that plays the role of a dask-backed cutout (file-loaded
    cutouts always are, atlite/cutout.py:143,151-153): aggregated results then come back as
    ``(time, <index>)`` like the reference's dask branch, otherwise ``(<index>, time)``
    (atlite/aggregate.py:21-35).
    """

    def __init__(self, data_vars, coords, attrs=None, chunked=False, repack=None, static=False):
        # repack: copy (time, y, x) variables the CALLER holds on the device into the library's padded, slot-interleaved
        # layout on first use when the cell count is not a multiple of 16 (contiguous cubes off the 128-byte line grid run
        # the fused kernels 15-35 % slower: one extra line per 1-KiB wave load).  Costs one device-to-device copy and the
        # cubes' size in HBM again; pays off from the second conversion on.  Default: $ATLITE_HIP_REPACK == "1".
        self.repack = (os.environ.get("ATLITE_HIP_REPACK", "0") == "1") if repack is None else bool(repack)
        # static: the caller promises not to rewrite the DEVICE arrays it hands in while this dataset is in use (a cutout's
        # data never changes, atlite/cutout.py:151-153), so what is derived from their contents - the night early-out's day
        # maps - may be cached with them.  The library's own copies of host / file data always are.
        self.static = bool(static)
        self.coords = {}
        for k, v in coords.items():
            self.coords[k] = pd.DatetimeIndex(v) if k == "time" else np.asarray(v, dtype=np.float64)
        if "lon" not in self.coords and "x" in self.coords:
            self.coords["lon"] = self.coords["x"]  # atlite/gis.py:73
        if "lat" not in self.coords and "y" in self.coords:
            self.coords["lat"] = self.coords["y"]
        self.attrs = dict(attrs or {})
        self.chunked = bool(chunked)
        self._vars = {}
        for k, v in data_vars.items():
            self[k] = v
        self._device_cache = {}

    @classmethod
    def from_xarray(cls, ds, chunked=None):
        coords = {k: ds.coords[k].values for k in ("time", "y", "x") if k in ds.coords}
        if chunked is None:
            chunked = bool(getattr(ds, "chunks", None))
        dv = {}
        for k in ds.data_vars:
            da = ds[k]
            dims = tuple(d for d in ("time", "y", "x") if d in da.dims)
            dv[k] = LabeledArray(da.transpose(*dims).values, dims)
        return cls(dv, coords, dict(ds.attrs), chunked=chunked)

    def __setitem__(self, name, value):
        if isinstance(value, LabeledArray):
            la = value
        else:
            shape = tuple(value.shape)
            T, Y, X = (len(self.coords.get(k, ())) for k in ("time", "y", "x"))
            if len(shape) == 3:
                dims = ("time", "y", "x")
            elif len(shape) == 2 and shape == (Y, X):
                dims = ("y", "x")
            elif len(shape) == 2 and shape == (T, Y * X):
                value = value.reshape(T, Y, X)
                dims = ("time", "y", "x")
            elif len(shape) == 1 and shape == (Y * X,):
                value = value.reshape(Y, X)
                dims = ("y", "x")
            else:
                raise ValueError(f"cannot infer dims of variable {name!r} with shape {shape}")
            la = LabeledArray(value, dims)
        la.coords = {d: self.coords[d] for d in la.dims if d in self.coords}
        if la.name is None:
            la.name = name
        self._vars[name] = la
        self._invalidate(name)

    def __delitem__(self, name):
        del self._vars[name]
        self._invalidate(name)

    def _invalidate(self, name):
        """A variable was replaced, added or removed: its device copy and every cached time shard of the dataset
        (the multi-GPU executor's per-rank ``isel_time`` views and THEIR device copies) are stale."""
        if hasattr(self, "_device_cache"):
            self._device_cache.pop(name, None)
        self.__dict__.pop("_shard_cache", None)

    def __getitem__(self, name):
        if name in self._vars:
            return self._vars[name]
        if name in self.coords:
            c = self.coords[name]
            dim = {"lon": "x", "lat": "y"}.get(name, name)
            return LabeledArray(np.asarray(c), (dim,), {dim: self.coords[dim]}, name=name)
        raise KeyError(name)

    def __contains__(self, name):
        return name in self._vars

    def __iter__(self):
        return iter(self._vars)

    def keys(self):
        return self._vars.keys()

    @property
    def data_vars(self):
        return self._vars

    @property
    def sizes(self):
        return {k: len(self.coords[k]) for k in ("time", "y", "x") if k in self.coords}

    @property
    def indexes(self):
        return {k: pd.Index(v) for k, v in self.coords.items() if k in ("time", "y", "x")}

    def isel_time(self, start, stop):
        """
        The dataset restricted to time steps ``[start, stop)`` without copying or reading anything:
        host arrays and device arrays become views, file-backed variables lazy row ranges.  One
        rank's shard in a multi-GPU run: ``ds.isel_time(*edges[rank:rank + 2])`` with the edges of
        ``atlite_amd.distributed.time_partition``.
        """
        T = len(self.coords["time"])
        start, stop = int(start), int(stop)
        if not 0 <= start <= stop <= T:
            raise IndexError(f"time range [{start}, {stop}) outside the dataset's {T} steps")
        coords = dict(self.coords)
        coords["time"] = self.coords["time"][start:stop]
        out = Dataset({}, coords, self.attrs, chunked=self.chunked, repack=self.repack)
        for k, la in self._vars.items():
            if "time" not in la.dims:
                out[k] = la
                continue
            d = la.data
            if _is_device(d) or _is_lazy(d):
                d = d.slab(start, stop)
            else:
                d = d[start:stop]
            out[k] = LabeledArray(d, la.dims, attrs=la.attrs, name=la.name)
        if hasattr(self, "file"):
            out.file = self.file
        return out

    def sel(self, **indexers):
        """
        Label-based selection along ``time`` / ``y`` / ``x`` like ``xarray.Dataset.sel`` for what cutouts are cut
        with (atlite/cutout.py:387-414): a ``slice`` of labels (both ends inclusive), a single label, or a sequence
        of labels.  A contiguous time range stays a view (device arrays and file-backed variables included); spatial
        selections and scattered time steps copy - device-resident variables come back as host arrays then.
        """
        unknown = set(indexers) - {"time", "y", "x"}
        if unknown:
            raise KeyError(f"cannot select along {sorted(unknown)}: a cutout has time, y and x")
        pos = {}
        for dim, ix in indexers.items():
            idx = pd.Index(self.coords[dim])
            if isinstance(ix, slice):
                if ix.step not in (None, 1):
                    raise NotImplementedError("label slices with a step")
                lo, hi = idx.slice_locs(ix.start, ix.stop)
                pos[dim] = np.arange(lo, hi)
            elif np.ndim(ix) == 0 or isinstance(ix, str):
                loc = idx.get_loc(ix)  # (a date string selects its whole period, like pandas / xarray)
                pos[dim] = np.arange(loc.start, loc.stop) if isinstance(loc, slice) else np.atleast_1d(np.arange(len(idx))[loc])
            else:
                want = pd.DatetimeIndex(ix) if dim == "time" else pd.Index(np.asarray(ix, dtype=np.float64))
                loc = idx.get_indexer(want)
                if (loc < 0).any():
                    raise KeyError(f"not all values found in index {dim!r}")
                pos[dim] = loc
        out = self
        tp = pos.pop("time", None)
        if tp is not None:
            contiguous = len(tp) > 0 and np.array_equal(tp, np.arange(tp[0], tp[0] + len(tp)))
            if contiguous or len(tp) == 0:
                out = out.isel_time(int(tp[0]) if len(tp) else 0, int(tp[0]) + len(tp) if len(tp) else 0)
                tp = None
        if tp is None and not pos:
            return out
        coords = dict(out.coords)
        for dim, p in ([("time", tp)] if tp is not None else []) + list(pos.items()):
            coords[dim] = out.coords[dim][p]
            for alias, of in (("lon", "x"), ("lat", "y")):
                if of == dim and alias in coords:
                    coords[alias] = coords[dim]
        new = Dataset({}, coords, out.attrs, chunked=out.chunked, repack=out.repack)
        for k, la in out._vars.items():
            v = np.asarray(la.data.numpy() if _is_device(la.data) else la.data)
            for ax, dim in enumerate(la.dims):
                p = tp if dim == "time" else pos.get(dim)
                if p is not None:
                    v = np.take(v, p, axis=ax)
            new[k] = LabeledArray(np.ascontiguousarray(v), la.dims, attrs=la.attrs, name=la.name)
        return new

    def pin(self):
        """Page-lock the host arrays in place so that the slab pipeline (atlite_amd.streaming) DMAs
        them at PCIe rate without re-registering on every call.  Undone by ``unpin()`` / deletion."""
        from . import _lib

        lib = _lib.load()
        self._pinned = getattr(self, "_pinned", {})
        for k, la in self._vars.items():
            a = la.data
            if isinstance(a, np.ndarray) and a.nbytes and a.flags.c_contiguous and k not in self._pinned:
                if lib.atl_host_register(a.ctypes.data, a.nbytes) == 0:
                    self._pinned[k] = (a.ctypes.data, a.nbytes)
        return self

    def unpin(self):
        from . import _lib

        for ptr, _ in getattr(self, "_pinned", {}).values():
            _lib.load().atl_host_unregister(ptr)
        self._pinned = {}

    def pinned_ranges(self):
        return list(getattr(self, "_pinned", {}).values())

    def __del__(self):
        try:
            self.unpin()
        except Exception:
            pass

    def device(self, ctx, name):
        """DeviceArray of variable ``name`` flattened to (T, S) or (S,); uploads host data once."""
        la = self._vars[name]
        if name not in self._device_cache or self._device_cache[name].ctx is not ctx:
            self._device_cache[name] = self._to_device(ctx, la)
        d = self._device_cache[name]
        if la.dims == ("time", "y", "x"):
            return d.reshape(d.shape[0], -1) if d.ndim == 3 else d
        return d.reshape(-1)

    def device_group(self, ctx, names):
        """
        ``{name: DeviceArray}`` of the variables ONE conversion call reads.  The (time, y, x) ones among them that this
        dataset uploads itself go into one slot-interleaved allocation (``device.SlotPool``: the variables of a time step
        side by side - the fused kernels stream 5-9 % faster from it than from separate allocations) and share its
        slot stride; a later call that needs a different set regroups on the device.  Variables the caller already holds
        on a device stay where they are (contiguous), and so does everything with ``ATLITE_HIP_INTERLEAVE=0``.
        """
        from .device import interleave_enabled

        names = list(dict.fromkeys(names))
        cubes = [n for n in names if self._vars[n].dims == ("time", "y", "x")]
        if len(cubes) > 1 and interleave_enabled() and not self._caller_layout():
            self._pool_cubes(ctx, cubes)
        return {n: self.device(ctx, n) for n in names}

    def _pool_cubes(self, ctx, cubes):
        from .device import SlotPool, pitch_for

        cache = self._device_cache
        have = {n: cache[n] for n in cubes if n in cache and cache[n].ctx is ctx}
        T, S = len(self.coords["time"]), len(self.coords["y"]) * len(self.coords["x"])
        for p in {id(q): q for q in (getattr(d, "_pool", None) for d in have.values()) if q is not None}.values():
            if all(n in p.names for n in cubes) and (p.T, p.S) == (T, S):
                for n in cubes:  # a replaced variable returns to the slot it had
                    if not (n in have and getattr(have[n], "_pool", None) is p):
                        cache[n] = self._fill(ctx, p.view(n), n, have.get(n))
                return
        p = SlotPool(ctx, T, S, cubes, pitch_for(S))
        for n in cubes:
            cache[n] = self._fill(ctx, p.view(n), n, have.get(n))

    def _fill(self, ctx, view, name, resident=None):
        """Variable ``name`` into ``view``: from its device copy if it has one (another layout), else from the host / file."""
        from ._lib import check

        T, S = view.shape
        from .device import root_block

        root_block(view).__dict__.pop("_day_maps", None)  # what was derived from the block's old contents
        if resident is not None:
            r = resident.reshape(T, S)
            check(ctx.lib.atl_copy_2d(ctx.handle, view.ptr, (view.ld or S) * 8, r.ptr, (r.ld or S) * 8, S * 8, T, 2, 0))
            ctx.sync()  # the old copy may be freed as soon as the cache lets go of it
            return view
        from .device import mark_static

        x = self._vars[name].data
        if getattr(x, "is_file_array", False):
            return mark_static(x.to_device(ctx, out=view))
        if _is_device(x) or (type(x).__module__.startswith("torch") and getattr(x, "is_cuda", False)):
            r = ctx.asdevice(x).reshape(T, S)  # the caller's own device cube (repack): copied on the device
            check(ctx.lib.atl_copy_2d(ctx.handle, view.ptr, (view.ld or S) * 8, r.ptr, (r.ld or S) * 8, S * 8, T, 2, 0))
            ctx.sync()
            return view
        if type(x).__module__.startswith("torch") and hasattr(x, "data_ptr"):
            x = x.numpy()
        return mark_static(ctx.upload(np.asarray(x).reshape(T, -1), out=view))

    def _caller_layout(self):
        """True when a (time, y, x) variable already lives on a device in the caller's own (contiguous) layout, which every
        cube of a call must then share."""
        if self.repack and (len(self.coords["y"]) * len(self.coords["x"])) % 16 != 0:
            return False  # the library makes padded copies of the caller's device cubes (see __init__)
        return any(la.dims == ("time", "y", "x") and (_is_device(la.data) or (type(la.data).__module__.startswith("torch")
                                                                               and getattr(la.data, "is_cuda", False)))
                   for la in self._vars.values())

    def _slot_stride(self):
        """Cells between the slots of the device copies this dataset makes of its (time, y, x) variables: padded to a
        128-byte line when the cell count is not a multiple of 16 (``device.pitch_for``) - unless a variable already
        lives on a device in the caller's own (contiguous) layout, which every cube of a call must then share."""
        from .device import pitch_for

        if self._caller_layout():
            return None
        return pitch_for(len(self.coords["y"]) * len(self.coords["x"]))

    def _to_device(self, ctx, la):
        ld = self._slot_stride() if la.dims == ("time", "y", "x") else None
        x = la.data
        if ld is not None and (_is_device(x) or (type(x).__module__.startswith("torch") and getattr(x, "is_cuda", False))):
            d = ctx.asdevice(x)  # repack: a padded copy of the caller's device cube
            return ctx._relayout(d.reshape(d.shape[0], -1), ld)
        from .device import mark_static

        if _is_device(x) or (type(x).__module__.startswith("torch") and getattr(x, "is_cuda", False)):
            d = ctx.asdevice(x)  # the caller's own device memory, as it lies
            return mark_static(d) if self.static and _is_device(x) else d
        if ld is None:
            return mark_static(ctx.asdevice(x))
        T = x.shape[0]

        if getattr(x, "is_file_array", False):  # inflate + decode straight into the padded rows
            return mark_static(x.to_device(ctx, ld=ld))
        if type(x).__module__.startswith("torch") and hasattr(x, "data_ptr"):
            x = x.numpy()
        return mark_static(ctx.upload(np.asarray(x).reshape(T, -1), ld=ld))

    def __repr__(self):
        return f"<Dataset {self.sizes} vars={list(self._vars)} chunked={self.chunked}>"
