"""
Cutout files: open a NetCDF-4 cutout the way ``atlite.Cutout(path)`` does
(atlite/cutout.py:143,151-153 - ``xr.open_dataset(path, chunks={"time": 100})``), without xarray,
netCDF4 or libhdf5: the container is parsed by the library itself (``atl_nc_*``,
include/atlite_hip.h), variables stay on disk as lazy ``FileArray`` objects and are inflated
chunk by chunk on host threads, DMA'd in their on-disk dtype and un-shuffled / widened to fp64 /
CF-decoded on the device while the conversion kernels work on the previous time slab
(``atlite_amd.streaming``).  SURVEY.md section 8 row f-4.
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np
import pandas as pd

from . import _lib
from ._lib import check

_TIME_UNITS_NS = {
    "days": 86_400_000_000_000, "day": 86_400_000_000_000, "d": 86_400_000_000_000,
    "hours": 3_600_000_000_000, "hour": 3_600_000_000_000, "h": 3_600_000_000_000, "hrs": 3_600_000_000_000,
    "minutes": 60_000_000_000, "minute": 60_000_000_000, "min": 60_000_000_000,
    "seconds": 1_000_000_000, "second": 1_000_000_000, "s": 1_000_000_000, "sec": 1_000_000_000,
    "milliseconds": 1_000_000, "microseconds": 1_000, "nanoseconds": 1,
}


def decode_time(values, units, calendar="standard"):
    """CF time -> DatetimeIndex ("<unit> since <reference>", standard / proleptic_gregorian calendars)."""
    if calendar not in (None, "", "standard", "gregorian", "proleptic_gregorian"):
        raise NotImplementedError(f"calendar {calendar!r} is not supported")
    unit, sep, ref = units.partition(" since ")
    unit = unit.strip().lower()
    if not sep or unit not in _TIME_UNITS_NS:
        raise ValueError(f"cannot decode time units {units!r}")
    ref = pd.Timestamp(ref.strip())
    if ref.tzinfo is not None:
        ref = ref.tz_convert("UTC").tz_localize(None)
    v = np.asarray(values, dtype=np.float64)
    per = _TIME_UNITS_NS[unit]
    whole = np.floor(v)
    ns = whole.astype(np.int64) * per + np.round((v - whole) * per).astype(np.int64)
    return pd.DatetimeIndex(ref.value + ns)


def _text(fn, *args):
    need = C.c_int64()
    check(fn(*args, None, 0, C.byref(need)))
    if need.value <= 0:
        return None
    buf = C.create_string_buffer(int(need.value))
    check(fn(*args, buf, len(buf), C.byref(need)))
    return buf.value.decode("utf-8", "replace")


class NcVariable:
    """Metadata of one variable (``atl_nc_inquire`` + ``atl_nc_dims``)."""

    def __init__(self, file, name):
        info = _lib.NcVar()
        check(file.lib.atl_nc_inquire(file.handle, name.encode(), C.byref(info)))
        self.name = name
        self.ndim = int(info.ndim)
        self.shape = tuple(int(v) for v in info.shape[: self.ndim])
        self.chunks = tuple(int(v) for v in info.chunk[: self.ndim])
        self.dtype = _lib.NC_DTYPES.get(int(info.dtype))  # on-disk dtype name, None = not numeric
        self.big_endian = bool(info.big_endian)
        self.layout = {0: "compact", 1: "contiguous", 2: "chunked"}.get(int(info.layout), "unsupported")
        self.shuffle, self.fletcher32 = bool(info.shuffle), bool(info.fletcher32)
        self.deflate = int(info.deflate) - 1 if info.deflate else None
        self.scale_factor = float(info.scale_factor) if info.has_scale else None
        self.add_offset = float(info.add_offset) if info.has_scale else None
        self.fill_value = float(info.fill_value) if info.has_fill else None
        self.missing_value = float(info.missing_value) if info.has_missing else None
        self.n_chunks, self.stored_bytes = int(info.n_chunks), int(info.stored_bytes)
        d = _text(file.lib.atl_nc_dims, file.handle, name.encode())
        self.dims = tuple(d.split("\n")) if d is not None and self.ndim else ()

    def __repr__(self):
        return f"<NcVariable {self.name!r} {self.dtype} {dict(zip(self.dims, self.shape))} chunks={self.chunks}>"


class NcFile:
    """A NetCDF-4 / HDF5 file opened by the native reader (host only; no GPU needed to inspect or read)."""

    def __init__(self, path):
        self.lib = _lib.load()
        self.path = os.fspath(path)
        h = C.c_void_p()
        check(self.lib.atl_nc_open(self.path.encode(), C.byref(h)))
        self.handle = h
        names = _text(self.lib.atl_nc_list, h) or ""
        self.variables = {n: NcVariable(self, n) for n in names.split("\n") if n}

    def attr(self, var, name):
        """Attribute as str, float, ndarray or None (``var=None``: global attribute)."""
        v = var.encode() if var else None
        out = np.empty(64)
        n = C.c_int64()
        rc = self.lib.atl_nc_att_double(self.handle, v, name.encode(), out.ctypes.data, out.size, C.byref(n))
        if rc == 0 and n.value:
            if n.value > out.size:
                out = np.empty(n.value)
                check(self.lib.atl_nc_att_double(self.handle, v, name.encode(), out.ctypes.data, out.size, C.byref(n)))
            return float(out[0]) if n.value == 1 else out[: n.value].copy()
        if rc == 0:
            return None
        return _text(self.lib.atl_nc_att_text, self.handle, v, name.encode())

    def read(self, name, start=0, count=None):
        """Rows ``[start, start+count)`` of a variable, CF-decoded to fp64 on the host."""
        var = self.variables[name]
        if var.ndim == 0:
            raise ValueError(f"variable {name!r} is a scalar")
        count = var.shape[0] - start if count is None else count
        out = np.empty((count,) + var.shape[1:], dtype=np.float64)
        check(self.lib.atl_nc_read_host(self.handle, name.encode(), int(start), int(count), out.ctypes.data))
        return out

    def read_slab(self, ctx, name, start, count, dptr, n_threads=0, ld=0):
        """Rows -> fp64 block at device pointer ``dptr`` (``ld``: cells between its slots, 0 = contiguous); visible in the
        order of the context's COPY stream."""
        check(self.lib.atl_nc_read_slab_ld(ctx.handle, int(ld or 0), self.handle, name.encode(), int(start), int(count), int(dptr),
                                           int(n_threads)))

    def read_slabs(self, ctx, names, start, count, dptrs, n_threads=0, ld=0):
        """The same rows of several variables -> fp64 blocks at the device pointers ``dptrs``: one call, and - when their chunks
        are zlib streams - ONE device launch that inflates all of them (``atl_nc_read_slabs``)."""
        n = len(names)
        c_names = (C.c_char_p * n)(*[s.encode() for s in names])
        c_outs = (C.c_void_p * n)(*[int(p) for p in dptrs])
        check(self.lib.atl_nc_read_slabs_ld(ctx.handle, int(ld or 0), self.handle, n, c_names, int(start), int(count), c_outs,
                                            int(n_threads)))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.atl_nc_close(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __repr__(self):
        return f"<NcFile {self.path!r} {list(self.variables)}>"


class FileArray:
    """
    A variable that lives in the file: shape / dtype of the DECODED array (fp64), rows read on
    demand.  ``np.asarray(a)`` and ``a[t0:t1]`` read on the host; ``read_slab`` feeds the device;
    ``slab(t0, t1)`` is a lazy view of a row range (a rank's time shard, ``Dataset.isel_time``).
    """

    is_file_array = True

    def __init__(self, file, name, row0=0, rows=None):
        self.file, self.name = file, name
        self.var = file.variables[name]
        full = self.var.shape
        self.row0 = int(row0)
        rows = full[0] - self.row0 if rows is None else int(rows)
        if not (0 <= self.row0 and 0 <= rows and self.row0 + rows <= full[0]):
            raise IndexError(f"rows [{self.row0}, {self.row0 + rows}) outside {name!r} with {full[0]} rows")
        self.shape = (rows,) + full[1:]
        self.dtype = np.dtype(np.float64)
        self.ndim = len(self.shape)

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    @property
    def nbytes(self):
        return self.size * 8

    def slab(self, start, stop):
        start, stop = int(start), int(stop)
        if not 0 <= start <= stop <= self.shape[0]:
            raise IndexError(f"slab [{start}, {stop}) outside {self.shape[0]} rows")
        return FileArray(self.file, self.name, self.row0 + start, stop - start)

    def __array__(self, dtype=None, copy=None):
        a = self.file.read(self.name, self.row0, self.shape[0])
        return a if dtype is None else a.astype(dtype)

    def __getitem__(self, key):
        if isinstance(key, slice):
            start, stop, step = key.indices(self.shape[0])
            if step == 1:
                return self.file.read(self.name, self.row0 + start, max(stop - start, 0))
        return np.asarray(self)[key]

    def read_slab(self, ctx, t0, t1, dptr, ld=0):
        self.file.read_slab(ctx, self.name, self.row0 + t0, t1 - t0, dptr, ld=ld)

    def to_device(self, ctx, block_bytes=256 << 20, ld=None, out=None):
        """Whole variable as a DeviceArray (rows in blocks so that the pinned staging stays bounded); ``ld``: a
        (time, y, x) variable as a (T, S) block with slots ``ld`` cells apart (``device.pitch_for``); ``out``: an
        existing (T, S) block to fill instead (its own slot stride: a ``device.SlotPool`` view)."""
        from ._lib import check

        row = max(int(np.prod(self.shape[1:], dtype=np.int64)), 1)
        if out is not None:
            assert len(self.shape) == 3 and out.shape == (self.shape[0], row), (self.shape, out.shape)
            ld = out.ld
        pitched = ld is not None and len(self.shape) == 3 and int(ld) > row
        if out is None:
            out = ctx.empty_pitched((self.shape[0], row), int(ld)) if pitched else ctx.empty(self.shape)
        stride = int(ld) if pitched else row
        step = max(1, block_bytes // (row * 8))
        if self.var.layout == "chunked":
            ct = self.var.chunks[0]
            if self.var.deflate is not None and os.environ.get("ATLITE_HIP_INFLATE", "") in ("", "device"):
                # the chunks' zlib streams are inflated on the device, one wavefront per stream (atl_nc_read_slab): a call should
                # carry thousands of them - up to 1.5 GiB of on-disk dtype per call (staging: two slots of that)
                es = np.dtype(self.var.dtype).itemsize if self.var.dtype else 8
                step = max(step, int(os.environ.get("ATLITE_HIP_INFLATE_BLOCK", 3 << 29)) // max(row * es, 1))
            step = max(ct, step // ct * ct)
        for t0 in range(0, self.shape[0], step):
            t1 = min(self.shape[0], t0 + step)
            self.read_slab(ctx, t0, t1, out.ptr + t0 * stride * 8, ld=stride if pitched else 0)  # where the rows go
        ctx.copy_barrier()
        return out

    def __repr__(self):
        rows = f"rows {self.row0}:{self.row0 + self.shape[0]} of " if self.shape[0] != self.var.shape[0] else ""
        return f"<FileArray {rows}{self.name!r} {self.shape} in {self.file.path!r}>"


def open_cutout(path, chunked=True):
    """
    ``Dataset`` view of a cutout file: coordinates and static (y, x) fields are read eagerly,
    (time, y, x) variables stay lazy.  ``chunked=True`` mirrors the reference, whose file-backed
    cutouts are always dask arrays, so aggregated results come back as (time, <index>)
    (atlite/aggregate.py:21-35).
    """
    from .labeled import Dataset, LabeledArray

    f = NcFile(path)
    names = f.variables

    def coord(*cands):
        for c in cands:
            if c in names and names[c].ndim == 1 and names[c].dtype:
                return c
        return None

    tn, yn, xn = coord("time", "valid_time"), coord("y", "lat", "latitude"), coord("x", "lon", "longitude")
    if not (yn and xn):
        raise ValueError(f"{path}: no y/x (or lat/lon) coordinate variables found; is this a cutout?")
    coords = {"y": f.read(yn), "x": f.read(xn)}
    for k in ("y", "x"):
        if len(coords[k]) > 1 and not np.all(np.diff(coords[k]) > 0):
            raise NotImplementedError(
                f"{path}: coordinate {k!r} is not ascending; atlite cutouts store x and y ascending "
                "(atlite/gis.py:63-74) - re-save the dataset sorted, e.g. ds.sortby([\"y\", \"x\"])")
    if tn:
        units = f.attr(tn, "units")
        cal = f.attr(tn, "calendar")
        if not isinstance(units, str):
            raise ValueError(f"{path}: time variable {tn!r} has no CF 'units' attribute")
        coords["time"] = decode_time(f.read(tn), units, cal if isinstance(cal, str) else "standard")
    for extra in ("lon", "lat"):
        if extra in names and names[extra].ndim == 1 and extra not in (yn, xn):
            coords[extra] = f.read(extra)
    attrs = {}
    for a in ("module", "prepared_features", "dx", "dy", "chunksize_time"):
        v = f.attr(None, a)
        if v is not None:
            attrs[a] = v
    ds = Dataset({}, coords, attrs, chunked=chunked)
    T, Y, X = (len(coords.get(k, ())) for k in ("time", "y", "x"))
    want3, want2 = (tn, yn, xn), (yn, xn)
    for n, v in names.items():
        if not v.dtype or n in (tn, yn, xn):
            continue
        if v.ndim == 3 and v.shape == (T, Y, X):
            if all(v.dims) and v.dims != want3:
                raise NotImplementedError(f"{path}: variable {n!r} has dims {v.dims}; (time, y, x) order is required")
            ds[n] = LabeledArray(FileArray(f, n), ("time", "y", "x"))
        elif v.ndim == 2 and v.shape == (Y, X):
            if all(v.dims) and v.dims != want2:
                raise NotImplementedError(f"{path}: variable {n!r} has dims {v.dims}; (y, x) order is required")
            ds[n] = LabeledArray(f.read(n), ("y", "x"))
    ds.file = f
    return ds


def describe(path):
    """Human-readable summary of a NetCDF-4 / HDF5 file (``python -m atlite_amd.io <file>``)."""
    f = NcFile(path)
    lines = [f"{f.path}: {os.path.getsize(f.path) / 1e6:.1f} MB, {len(f.variables)} variables"]
    for name, v in sorted(f.variables.items()):
        dims = ", ".join(f"{d or '?'}={n}" for d, n in zip(v.dims or ("?",) * v.ndim, v.shape))
        enc = []
        if v.layout == "chunked":
            enc.append("chunks " + "x".join(str(c) for c in v.chunks))
        else:
            enc.append(v.layout)
        if v.shuffle:
            enc.append("shuffle")
        if v.deflate is not None:
            enc.append(f"deflate {v.deflate}")
        if v.fletcher32:
            enc.append("fletcher32")
        if v.scale_factor is not None:
            enc.append(f"scale {v.scale_factor:g} offset {v.add_offset:g}")
        if v.fill_value is not None:
            enc.append(f"_FillValue {v.fill_value:g}")
        raw = int(np.prod(v.shape, dtype=np.int64)) * (np.dtype(v.dtype).itemsize if v.dtype else 0)
        ratio = f", {v.stored_bytes / raw:.2f} of raw" if raw and v.stored_bytes else ""
        lines.append(f"  {name:24s} {v.dtype or 'non-numeric':8s}{'>' if v.big_endian else ' '} ({dims})  "
                     f"[{'; '.join(enc)}]  {v.stored_bytes / 1e6:.2f} MB stored{ratio}")
    f.close()
    return "\n".join(lines)


if __name__ == "__main__":
    import sys

    for p in sys.argv[1:]:
        print(describe(p))
