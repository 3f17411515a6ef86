"""
Pipelined execution for cutouts that live in HOST memory (or do not fit in HBM).

The time axis is cut into slabs; while the conversion kernels work on slab ``i`` (compute
stream), the input variables of slab ``i+1`` are DMA'd from pinned host memory on the copy
stream into the other half of a double buffer; events order the two streams.  Device memory
is bounded by two slabs of every input variable, the end-to-end rate approaches the PCIe rate
(the kernels are ~100x faster than the link).  SURVEY.md 8 f-4; the reference reads its
cutouts chunk by chunk through dask (atlite/cutout.py:143) - this is the device-side analogue.

Three kinds of sources feed a slab buffer:
  * host fp64 arrays: one DMA (``atl_upload_async``);
  * host arrays of a narrower dtype (float32 is what xarray hands over for a real cutout): DMA'd
    as they are and widened on the device (``atl_upload_convert_async``) - half the PCIe bytes;
  * variables of a cutout FILE (``atlite_amd.io.FileArray``): chunks inflated on host threads,
    DMA'd in the on-disk dtype, un-shuffled / CF-decoded on the device (``atl_nc_read_slab``).

Used automatically by ``convert_and_aggregate`` for file-backed datasets and for host-resident ones
above ``ATLITE_HIP_STREAM_MIN_BYTES`` (default 256 MiB; ``ATLITE_HIP_STREAM=0/1`` forces it off/on).
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

from ._lib import NC_CODES, check

COMPUTE, COPY = 0, 1


def _host_array(la):
    """The variable's host-side source: ndarray or FileArray; None for device-resident data."""
    d = la.data
    return d if isinstance(d, np.ndarray) or getattr(d, "is_file_array", False) else None


def _is_file(a):
    return getattr(a, "is_file_array", False)


def _source(a, T, S):
    """Normalise one source: FileArray as is; ndarray C-contiguous (T, S) in a dtype the device decodes."""
    if _is_file(a):
        return a
    if a.dtype.str.lstrip("<=|") not in ("f8", "f4", "i1", "i2", "i4", "i8", "u1", "u2", "u4", "u8") or \
            a.dtype.byteorder == ">":
        a = a.astype(np.float64)
    return np.ascontiguousarray(a).reshape(T, S)


def wanted(ds, spec):
    """Stream iff every time-dependent input is a host ndarray and the total is large enough."""
    mode = os.environ.get("ATLITE_HIP_STREAM", "auto")
    if mode == "0" or not getattr(spec, "time_vars", None):
        return False
    arrs = [_host_array(ds[n]) for n in spec.time_vars]
    if any(a is None for a in arrs):
        return False
    if mode == "1" or any(_is_file(a) for a in arrs):
        return True
    return sum(a.nbytes for a in arrs) >= int(os.environ.get("ATLITE_HIP_STREAM_MIN_BYTES", 256 << 20))


class _Pinned:
    """Pins host arrays in place for the duration of a run (no-op if pinning fails)."""

    def __init__(self, lib, arrays, already=()):
        self.lib, self.ptrs = lib, []
        if os.environ.get("ATLITE_HIP_PIN", "1") == "0":
            return
        for a in arrays:
            lo, hi = a.ctypes.data, a.ctypes.data + a.nbytes
            if any(p <= lo and hi <= p + n for p, n in already):
                continue  # Dataset.pin() already page-locked this array
            if a.nbytes and lib.atl_host_register(a.ctypes.data, a.nbytes) == 0:
                self.ptrs.append(a.ctypes.data)

    def release(self):
        for p in self.ptrs:
            self.lib.atl_host_unregister(p)
        self.ptrs = []


class _SlabView:
    """Dataset look-alike for one slab: ``coords`` with the slab's times, ``device()`` of its buffers."""

    def __init__(self, ds, bufs, static, t0, t1):
        self.coords = dict(ds.coords)
        self.coords["time"] = ds.coords["time"][t0:t1]
        self._bufs, self._static, self._n = bufs, static, t1 - t0

    def device(self, ctx, name):
        if name in self._bufs:
            return self._bufs[name].slab(0, self._n)
        return self._static[name]


def run(ctx, spec, ds, plan, time_agg):
    """
    Execute ``spec`` slab by slab.  Returns a DeviceArray shaped like ``spec.run`` would return
    for ``time_agg=None`` ((N, slots) with a plan, (slots, S) without) or, for per-cell
    ``time_agg="sum"``, the (S,) sum.  Other time reductions are left to the caller.
    """
    assert time_agg in (None, "sum")
    lib = ctx.lib
    T = len(ds.coords["time"])
    S = len(ds.coords["y"]) * len(ds.coords["x"])
    host = {n: _source(_host_array(ds[n]), T, S) for n in spec.time_vars}
    # slab = 128 MiB of fp64 per variable (DMA sources), 512 MiB for file sources: one read call inflates
    # a slab's chunks in parallel, so a bigger slab keeps more host threads busy
    from_file = any(_is_file(a) for a in host.values())
    # ... and 4 GiB when the chunks are zlib streams that the DEVICE inflates (one wavefront per stream, the device holds 8192 of
    # them; a read is ONE launch that its own DMAs feed, so a year of the C2 grid is best read as one slab)
    on_device = from_file and os.environ.get("ATLITE_HIP_INFLATE", "") in ("", "device") and any(
        _is_file(a) and a.var.deflate is not None for a in host.values())
    slab_bytes = int(os.environ.get("ATLITE_HIP_SLAB_BYTES", (4 << 30) if on_device else (512 << 20) if from_file else (128 << 20)))
    steps = int(os.environ.get("ATLITE_HIP_SLAB_STEPS", 0)) or max(8, min(T, slab_bytes // max(S * 8, 1)) // 8 * 8)
    # file sources: whole chunks per slab, so that no chunk is inflated twice
    tchunk = max([a.var.chunks[0] for a in host.values() if _is_file(a) and a.var.layout == "chunked"], default=0)
    if tchunk and not os.environ.get("ATLITE_HIP_SLAB_STEPS") and steps < T:
        steps = max(tchunk, steps // tchunk * tchunk)
    edges = spec.slab_edges(T, steps)
    n_slots = spec.n_slots(ds)
    static = {n: ds.device(ctx, n) for n in getattr(spec, "static_vars", ())}
    spec.prepare(ctx, ds)
    max_len = max((b - a for a, b in edges), default=0)
    # slots padded to a 128-byte line when S is not a multiple of 16 cells (device.pitch_for) - the kernels then read
    # whole lines
    from .device import pitch_for

    ld = pitch_for(S)
    import time as _time

    _t0 = _time.perf_counter()
    _dbg = os.environ.get("ATLITE_HIP_STREAM_TIMING") == "1"
    # two sets of slab buffers (one is filled while the other is converted); a single slab needs one.  Fresh device memory
    # costs ~30 ms per GB to map (a year of the C2 grid: 0.6 s for two sets, twice the read itself), so the sets of the last
    # call stay with the context - up to $ATLITE_HIP_SLAB_CACHE bytes (default 48 GiB of the 288) - for the next one
    n_sets = 2 if len(edges) > 1 else 1
    key = (max(max_len, 1), S, ld, tuple(host))
    cache = ctx.__dict__.setdefault("_slab_cache", {})
    bufs = cache.pop(key, [])[:n_sets]
    while len(bufs) < n_sets:
        bufs.append({n: (ctx.empty_pitched((max(max_len, 1), S), ld) if ld else ctx.empty((max(max_len, 1), S))) for n in host})
    if _dbg:
        ctx.sync()
        print(f"[streaming] {len(edges)} slab(s) of <= {max_len} steps, buffers allocated in {(_time.perf_counter() - _t0) * 1e3:.1f} ms", flush=True)
    ev_ready, ev_done = [], []
    for _ in range(2):
        for lst in (ev_ready, ev_done):
            h = C.c_void_p()
            check(lib.atl_event_create(ctx.handle, C.byref(h)))
            lst.append(h)
    if plan is not None:
        out = ctx.empty((plan.shape[0], n_slots))
    elif time_agg is None:
        out = ctx.empty((n_slots, S))
    else:
        out, host_acc = None, np.zeros(S)
    pinned = _Pinned(lib, [a for a in host.values() if not _is_file(a)], getattr(ds, "pinned_ranges", lambda: [])())
    try:
        for i, (t0, t1) in enumerate(edges):
            b = i % len(bufs)
            _t1 = _time.perf_counter()
            if i >= len(bufs):
                check(lib.atl_stream_wait_event(ctx.handle, COPY, ev_done[b]))
            # the file-backed variables of the slab in ONE read per file (atl_nc_read_slabs: one device launch inflates the chunk
            # streams of all of them)
            by_file = {}
            for n, a in host.items():
                if _is_file(a):
                    by_file.setdefault((id(a.file), a.row0), (a.file, a.row0, []))[2].append(n)
            for fobj, row0, names in by_file.values():
                fobj.read_slabs(ctx, [host[n].name for n in names], row0 + t0, t1 - t0, [bufs[b][n].ptr for n in names], ld=ld or 0)
            for n, a in host.items():
                if _is_file(a):
                    continue
                blk = a[t0:t1]
                if a.dtype == np.float64 and ld:
                    check(lib.atl_copy_2d(ctx.handle, bufs[b][n].ptr, ld * 8, blk.ctypes.data, S * 8, S * 8, t1 - t0, 0, 1))
                elif a.dtype == np.float64:
                    check(lib.atl_upload_async(ctx.handle, bufs[b][n].ptr, blk.ctypes.data, blk.nbytes))
                elif ld:
                    check(lib.atl_upload_convert_2d_async(ctx.handle, bufs[b][n].ptr, ld, blk.ctypes.data,
                                                          NC_CODES[a.dtype.name], t1 - t0, S))
                else:
                    check(lib.atl_upload_convert_async(ctx.handle, bufs[b][n].ptr, blk.ctypes.data,
                                                       NC_CODES[a.dtype.name], blk.size))
            _t2 = _time.perf_counter()
            check(lib.atl_event_record(ctx.handle, ev_ready[b], COPY))
            check(lib.atl_stream_wait_event(ctx.handle, COMPUTE, ev_ready[b]))
            if _dbg:
                print(f"[streaming] slab {i}: reads enqueued in {(_t2 - _t1) * 1e3:.1f} ms, copy stream observed after {(_time.perf_counter() - _t2) * 1e3:.1f} ms more", flush=True)
            view = _SlabView(ds, bufs[b], static, t0, t1)
            sub = spec.for_slab(t0, t1)
            s0, s1 = spec.out_slots(t0, t1)
            if plan is not None:
                sub.run(ctx, view, plan, None, out=(out.ptr + s0 * 8, n_slots))
            elif time_agg is None:
                sub.run(ctx, view, None, None, out=(out.ptr + s0 * S * 8, S))
            else:
                host_acc += sub.run(ctx, view, None, "sum").numpy()  # (S,) per slab: tiny
            check(lib.atl_event_record(ctx.handle, ev_done[b], COMPUTE))
        ctx.sync()
        if _dbg:
            print(f"[streaming] all slabs done {(_time.perf_counter() - _t0) * 1e3:.1f} ms after the start", flush=True)
    finally:
        ctx.sync()
        cap = int(os.environ.get("ATLITE_HIP_SLAB_CACHE", 48 << 30))
        size = sum(a.nbytes for b_ in bufs for a in b_.values())
        if size <= cap:
            for k_ in list(cache):  # one shape at a time: a different cutout replaces it
                del cache[k_]
            cache[key] = bufs
        pinned.release()
        for h in ev_ready + ev_done:
            lib.atl_event_destroy(h)
    return out if out is not None else ctx.upload(host_acc)
