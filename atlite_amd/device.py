"""
Device-side objects: a HIP context, fp64 device arrays, aggregation plans and thin typed
wrappers of the C-ABI entry points (``include/atlite_hip.h``).  Host logic that mirrors the
reference's Python interface lives in ``atlite_amd.convert``.
"""

from __future__ import annotations

import ctypes as C
import math
import os
import threading

import numpy as np

from . import _lib
from ._lib import TIME_MEAN, TIME_NONE, TIME_SUM, check

_TIME_CODES = {None: TIME_NONE, "sum": TIME_SUM, "mean": TIME_MEAN, "sum_count": _lib.TIME_SUM_COUNT}


def pitch_for(S):
    """Slot stride (cells) of the library's own device copies of a (T, S) cube: slots padded to a whole 128-byte line
    when S is not a multiple of 16 cells, so that every slot starts on a line (the kernels own and stream whole lines:
    DESIGN.md section 3, round 3).  None = contiguous.  ``ATLITE_HIP_PITCH=0`` switches the padding off."""
    S = int(S)
    if S % 16 == 0 or S == 0 or os.environ.get("ATLITE_HIP_PITCH", "1") == "0":
        return None
    return (S + 15) // 16 * 16


class _PinnedBlock:
    """Page-locked host memory (``atl_pinned_alloc``) behind a NumPy array: ``np.asarray(block)`` views it through
    ``__array_interface__`` and keeps the block alive as the array's base; when the last view dies the memory goes back to a
    small pool (page-locking costs ~100 us per allocation, a warm ``Cutout.pv()`` result is downloaded in less).

    Back-pressure: at most ``BUDGET`` bytes (``ATLITE_HIP_PINNED_BUDGET``, default 2 GiB) are page-locked at a time - blocks
    users still hold plus the pool; past that ``_host_array`` hands out ordinary memory.  Every download into a block is
    synchronous (``atl_download`` / ``atl_copy_2d`` return after the stream has drained), so a block that comes back here has
    no transfer in flight.  The pool is emptied at interpreter exit."""

    _pool = {}  # nbytes -> [pointers]
    _pooled = 0
    _live = 0  # bytes page-locked right now (held by users + pooled)
    _lock = threading.Lock()
    CAP = 512 << 20  # bytes kept for reuse
    BUDGET = int(os.environ.get("ATLITE_HIP_PINNED_BUDGET", 2 << 30))

    def __init__(self, lib, shape, dtype):
        self.lib = lib
        self.ptr = None
        self.nbytes = max(int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize, 1)
        with _PinnedBlock._lock:
            free = _PinnedBlock._pool.get(self.nbytes)
            self.ptr = free.pop() if free else None
            if self.ptr is not None:
                _PinnedBlock._pooled -= self.nbytes
            elif _PinnedBlock._live + self.nbytes > _PinnedBlock.BUDGET:
                raise MemoryError("page-locked result budget exhausted")
            else:
                _PinnedBlock._live += self.nbytes
        if self.ptr is None:
            p = C.c_void_p()
            try:
                check(lib.atl_pinned_alloc(self.nbytes, C.byref(p)))
            except Exception:
                with _PinnedBlock._lock:
                    _PinnedBlock._live -= self.nbytes
                raise
            self.ptr = p.value
        self.__array_interface__ = {"shape": tuple(int(v) for v in shape), "typestr": np.dtype(dtype).str,
                                    "data": (self.ptr, False), "version": 3}

    def __del__(self):
        try:
            if self.ptr is None:
                return
            with _PinnedBlock._lock:
                if _PinnedBlock._pool is not None and _PinnedBlock._pooled + self.nbytes <= _PinnedBlock.CAP:
                    _PinnedBlock._pool.setdefault(self.nbytes, []).append(self.ptr)
                    _PinnedBlock._pooled += self.nbytes
                    return
                _PinnedBlock._live -= self.nbytes
            self.lib.atl_pinned_free(self.ptr)
        except Exception:
            pass

    @staticmethod
    def trim(lib=None):
        """Free every pooled block (interpreter exit; tests)."""
        with _PinnedBlock._lock:
            ptrs = [p for v in (_PinnedBlock._pool or {}).values() for p in v]
            held = _PinnedBlock._pooled
            if _PinnedBlock._pool is not None:
                _PinnedBlock._pool.clear()
            _PinnedBlock._pooled = 0
            _PinnedBlock._live -= held
        if ptrs:
            lib = lib or _lib.load()
            for p in ptrs:
                lib.atl_pinned_free(p)


def _trim_pinned_at_exit():
    try:
        _PinnedBlock.trim()
    except Exception:
        pass


import atexit  # noqa: E402

atexit.register(_trim_pinned_at_exit)


def _host_array(lib, shape, dtype):
    """Destination of a download: page-locked for results between 64 KiB and 1 GiB (``ATLITE_HIP_PINNED_RESULTS=0``: never;
    ordinary memory as well once ``_PinnedBlock.BUDGET`` bytes are page-locked)."""
    nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
    if (64 << 10) <= nbytes <= (1 << 30) and os.environ.get("ATLITE_HIP_PINNED_RESULTS", "1") != "0":
        try:
            return np.asarray(_PinnedBlock(lib, shape, dtype))
        except (MemoryError, _lib.AtliteHipError):
            pass
    return np.empty(shape, dtype=dtype)


class DeviceArray:
    """An array resident in HBM (fp64 unless stated): C-contiguous, or - ``ld`` set - a (T, S) or (T, Y, X) block whose
    slots (first-axis items, contiguous in themselves) lie ``ld`` elements apart: padded slots, a cube of a
    slot-interleaved ``SlotPool``; only the first ``prod(shape[1:])`` elements of a slot are this array's data."""

    def __init__(self, ctx, ptr, shape, dtype=np.float64, owner=None, owned=True, ld=None):
        self.ctx = ctx
        self.ptr = int(ptr)
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self._owner = owner  # keeps a parent allocation (or a torch tensor) alive
        self._owned = owned and owner is None
        inner = math.prod(self.shape[1:]) if len(self.shape) > 1 else 0
        self.ld = None if ld is None or len(self.shape) < 2 or int(ld) == inner else int(ld)
        assert self.ld is None or self.ld > inner

    @property
    def size(self):
        return math.prod(self.shape)

    @property
    def nbytes(self):
        return self.size * self.dtype.itemsize

    @property
    def ndim(self):
        return len(self.shape)

    def numpy(self):
        out = _host_array(self.ctx.lib, self.shape, self.dtype)
        if out.size and self.ld is not None:
            es = self.dtype.itemsize
            inner = out.size // self.shape[0]
            check(self.ctx.lib.atl_copy_2d(self.ctx.handle, out.ctypes.data, inner * es, self.ptr, self.ld * es,
                                           inner * es, self.shape[0], 1, 0))
        elif out.size:
            check(self.ctx.lib.atl_download(self.ctx.handle, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)

    def slab(self, start, stop):
        """View of rows [start, stop) along the first axis (no copy)."""
        start, stop = int(start), int(stop)
        assert 0 <= start <= stop <= self.shape[0]
        row = (self.ld if self.ld is not None else math.prod(self.shape[1:])) * self.dtype.itemsize
        return DeviceArray(
            self.ctx, self.ptr + start * row, (stop - start,) + self.shape[1:], self.dtype, owner=self, ld=self.ld
        )

    def reshape(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        shape = [int(v) for v in shape]
        if -1 in shape:
            known = math.prod(v for v in shape if v != -1)
            shape[shape.index(-1)] = self.size // known if known else 0
        assert math.prod(shape) == self.size, (shape, self.shape)
        if self.ld is not None:
            if tuple(shape) == self.shape:
                return self
            if len(shape) < 2 or shape[0] != self.shape[0]:  # only the inside of a slot is contiguous
                raise ValueError(f"a pitched {self.shape} block cannot be viewed as {tuple(shape)}; download it first")
            a = DeviceArray(self.ctx, self.ptr, shape, self.dtype, owner=self, ld=self.ld)
            if hasattr(self, "_pool"):
                a._pool = self._pool
            return a
        return DeviceArray(self.ctx, self.ptr, shape, self.dtype, owner=self)

    def no_recycle(self):
        """This block is (or will be) touched by a stream the context does not own - a communicator's, another library's:
        when it is released it goes back to the driver (``atl_free`` synchronises) instead of the context's pool."""
        self._no_recycle = True
        return self

    def free(self):
        if self._owned and self.ptr:
            if getattr(self, "_no_recycle", False) or not self.ctx._recycle(self.ptr, self.nbytes):
                check(self.ctx.lib.atl_free(self.ctx.handle, self.ptr))
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def __repr__(self):
        pitch = "" if self.ld is None else f", ld={self.ld}"
        return f"DeviceArray(shape={self.shape}{pitch}, dtype={self.dtype}, device={self.ctx.device})"


def root_block(a):
    """The allocation a DeviceArray view ultimately points into (views keep their parent alive in ``_owner``)."""
    while isinstance(a._owner, DeviceArray):
        a = a._owner
    return a


def mark_static(a):
    """The library filled this block from host or file data and nobody else holds a pointer to write through: what is
    derived from its CONTENTS (the early-out's day maps, ``Context.pv``) may be kept with it."""
    root_block(a)._static = True
    return a


def interleave_enabled():
    """``ATLITE_HIP_INTERLEAVE=0``: every device copy in an allocation of its own (the layout before round 3)."""
    return os.environ.get("ATLITE_HIP_INTERLEAVE", "1") != "0"


class SlotPool:
    """
    One allocation that holds the ``n`` (T, S) cubes a conversion reads slot-interleaved: cube ``v`` of time step ``t``
    starts ``(t * n + v) * Sp`` cells in, so the variables of one time step lie side by side (``Sp``: cells of a slot
    rounded up to a 128-byte line, ``pitch_for``).  The kernels address cube v, slot t as ``ptr_v + t * ld`` and never
    see the difference (``ld = n * Sp``, the ``ld_cells`` argument of the ``atl_*_ld`` calls); the memory system does: with seven separate
    allocations the fused pv kernel streams at 6.2-6.3 TB/s, with this layout at 6.6-6.8.
    """

    def __init__(self, ctx, T, S, names, Sp=None):
        self.ctx, self.T, self.S = ctx, int(T), int(S)
        self.names = list(names)
        self.Sp = int(Sp or S)
        assert self.Sp >= self.S and self.names
        self.ld = len(self.names) * self.Sp
        self.base = ctx.empty((max(self.T * self.ld, 1),))
        if self.Sp > self.S:  # the padding is never read as data; keep it free of stray NaN patterns
            check(ctx.lib.atl_memset(ctx.handle, self.base.ptr, 0, self.base.nbytes))
            ctx.copy_after_compute()  # the cubes are filled on the copy stream: behind the memset

    def view(self, name):
        v = self.names.index(name)
        a = DeviceArray(self.ctx, self.base.ptr + v * self.Sp * 8, (self.T, self.S), owner=self.base, ld=self.ld)
        a._pool = self
        return a


class AggPlan:
    """Indicator matrix (scipy CSR, N x S) preprocessed into the segment-local device layout."""

    def __init__(self, ctx, matrix, row_len=None, ld=None, aligned=False):
        import scipy.sparse as sp

        m = sp.csr_matrix(matrix)
        self.ctx = ctx
        self.shape = m.shape
        self.aligned = bool(aligned)
        # the tile shape depends on whether the slots of the cubes the plan will meet start on 128-byte lines
        indptr = np.ascontiguousarray(m.indptr, dtype=np.int64)
        indices = np.ascontiguousarray(m.indices, dtype=np.int32)
        data = np.ascontiguousarray(m.data, dtype=np.float64)
        h = C.c_void_p()
        self.handle = None
        if aligned:  # line-aligned plan for contiguous cubes whose slots do not start on 128-byte lines (atl_agg_create_aligned)
            check(ctx.lib.atl_agg_create_aligned(ctx.handle, m.shape[0], m.shape[1],
                                                 int(row_len) if row_len and m.shape[1] % int(row_len) == 0 else 0, indptr.ctypes.data,
                                                 indices.ctypes.data if indices.size else None,
                                                 data.ctypes.data if data.size else None, C.byref(h)))
            self.handle = h
            return
        # the slot stride of the cubes the plan will meet travels with the call (atl_agg_create_ld)
        check(ctx.lib.atl_agg_create_ld(ctx.handle, int(ld) if ld and int(ld) != m.shape[1] else 0, m.shape[0], m.shape[1],
                                        int(row_len) if row_len and m.shape[1] % int(row_len) == 0 else 0, indptr.ctypes.data,
                                        indices.ctypes.data if indices.size else None,
                                        data.ctypes.data if data.size else None, C.byref(h)))
        self.handle = h

    def info(self):
        v = [C.c_int64() for _ in range(4)] + [C.c_int32(), C.c_int32()]
        check(self.ctx.lib.atl_agg_info(self.handle, *[C.byref(x) for x in v]))
        return dict(zip(("n_rows", "n_cells", "n_segments", "n_partial_rows", "tile_w", "tile_h"),
                        (x.value for x in v)))

    def close(self):
        if self.handle:
            self.ctx.lib.atl_agg_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One HIP device + stream.  Not thread-safe; use one per thread / per rank."""

    def __init__(self, device=0, stream=None):
        self.lib = _lib.load()
        n = C.c_int()
        check(self.lib.atl_device_count(C.byref(n)))
        if n.value == 0:
            raise _lib.AtliteHipError(
                "no HIP device visible: atlite_amd runs on MI355X (gfx950) only; there is no CPU fallback"
            )
        h = C.c_void_p()
        check(self.lib.atl_create(int(device), stream, C.byref(h)))
        self.handle = h
        self.device = int(device)

    # -- memory ---------------------------------------------------------------------------
    # small device blocks (results, per-call tables: <= 64 MiB each, 512 MiB kept) are recycled by size: hipMalloc + hipFree
    # cost ~0.3 ms a pair - hipFree synchronises the device - which was a tenth of a warm Cutout.pv() call.
    #
    # Stream safety by construction (round 5): a block enters the pool together with an event pair recorded at that moment
    # on the context's compute stream AND its copy stream - every stream of the context that can have touched the block
    # (blocks handed to a communicator's own stream are never pooled: ``DeviceArray.no_recycle()``) - and ``empty()`` waits on
    # the pair on the HOST before it hands the block out again, so the next user may run on any stream.  A block released by
    # a thread other than the context's owner (a ``__del__`` run by the garbage collector) is only queued; the owner records
    # its events when it next allocates.  ``ATLITE_HIP_RECYCLE=0`` switches recycling off (every free is ``atl_free``, which
    # synchronises), as does the fenced allocator ``ATLITE_HIP_FENCE=1``.  An allocation that fails for lack of memory
    # drains the pool and is retried once.
    _POOL_MAX_BLOCK, _POOL_CAP = 64 << 20, 512 << 20

    def _pool_state(self):
        st = self.__dict__.get("_pool_st")
        if st is None:
            on = os.environ.get("ATLITE_HIP_RECYCLE", "1") != "0" and os.environ.get("ATLITE_HIP_FENCE", "0") in ("", "0")
            st = self.__dict__.setdefault("_pool_st", {"on": on, "lock": threading.Lock(), "free": {}, "held": 0, "deferred": [],
                                                       "events": [], "owner": threading.get_ident()})
        return st

    def _fence_events(self):
        """An event pair marking 'now' on the compute and the copy stream (events are reused)."""
        st = self._pool_state()
        pair = st["events"].pop() if st["events"] else None
        if pair is None:
            pair = (C.c_void_p(), C.c_void_p())
            check(self.lib.atl_event_create(self.handle, C.byref(pair[0])))
            check(self.lib.atl_event_create(self.handle, C.byref(pair[1])))
        check(self.lib.atl_event_record(self.handle, pair[0], 0))
        # (2: the copy stream as a fence - ordered behind the device-inflate reads in flight, WITHOUT settling them: a failed
        #  read's verdict must reach whoever observes the copy stream, not be swallowed by a block being recycled)
        check(self.lib.atl_event_record(self.handle, pair[1], 2))
        return pair

    def _recycle(self, ptr, nbytes):
        if not (0 < nbytes <= self._POOL_MAX_BLOCK) or getattr(self, "handle", None) is None:
            return False
        st = self._pool_state()
        if not st["on"]:
            return False
        with st["lock"]:
            if st["held"] + nbytes > self._POOL_CAP:
                return False
            st["held"] += nbytes
            if threading.get_ident() != st["owner"]:
                st["deferred"].append((ptr, nbytes))  # the owner records the events (``_drain_deferred``)
                return True
        try:
            pair = self._fence_events()
        except Exception:
            with st["lock"]:
                st["held"] -= nbytes
            return False
        with st["lock"]:
            st["free"].setdefault(nbytes, []).append((ptr, pair))
        return True

    def _drain_deferred(self, st):
        with st["lock"]:
            todo, st["deferred"] = st["deferred"], []
        if not todo:
            return
        pair = None
        for ptr, nbytes in todo:
            try:
                pair = self._fence_events()
            except Exception:
                pair = None
            if pair is None:
                with st["lock"]:
                    st["held"] -= nbytes
                self.lib.atl_free(self.handle, ptr)
                continue
            with st["lock"]:
                st["free"].setdefault(nbytes, []).append((ptr, pair))

    def _pool_take(self, nbytes):
        st = self._pool_state()
        if not st["on"]:
            return None
        if st["deferred"] and threading.get_ident() == st["owner"]:
            self._drain_deferred(st)
        with st["lock"]:
            free = st["free"].get(nbytes)
            if not free:
                return None
            ptr, pair = free.pop()
            st["held"] -= nbytes
        # everything enqueued on the context's streams before the block was released has finished
        check(self.lib.atl_event_synchronize(pair[0]))
        check(self.lib.atl_event_synchronize(pair[1]))
        with st["lock"]:
            st["events"].append(pair)
        return ptr

    def trim(self):
        """Give every pooled device block back to the driver."""
        st = self.__dict__.get("_pool_st")
        if st is None or getattr(self, "handle", None) is None:
            return
        with st["lock"]:
            blocks = [b for v in st["free"].values() for b in v]
            todo, st["deferred"] = st["deferred"], []
            st["free"].clear()
            st["held"] = 0
        for ptr, pair in blocks:
            self.lib.atl_free(self.handle, ptr)  # synchronises the compute stream, then hipFree (which waits for the device)
            st["events"].append(pair)
        for ptr, _ in todo:
            self.lib.atl_free(self.handle, ptr)

    def empty(self, shape, dtype=np.float64):
        shape = (shape,) if np.isscalar(shape) else tuple(shape)
        nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
        ptr = self._pool_take(nbytes) if 0 < nbytes <= self._POOL_MAX_BLOCK else None
        if ptr is not None:
            return DeviceArray(self, ptr, shape, dtype)
        p = C.c_void_p()
        rc = self.lib.atl_alloc(self.handle, nbytes, C.byref(p))
        if rc == _lib.ATL_E_NOMEM and (self.__dict__.get("_pool_st", {}).get("held") or self.__dict__.get("_slab_cache")):
            self.__dict__.pop("_slab_cache", None)  # what the context only keeps for the NEXT call's sake goes first
            self.trim()
            rc = self.lib.atl_alloc(self.handle, nbytes, C.byref(p))
        check(rc)
        return DeviceArray(self, p.value, shape, dtype)

    def zeros(self, shape, dtype=np.float64):
        a = self.empty(shape, dtype)
        check(self.lib.atl_memset(self.handle, a.ptr, 0, a.nbytes))
        return a

    def upload(self, host, dtype=np.float64, ld=None, out=None):
        """Host array -> DeviceArray; ``ld`` (2-d arrays): rows ``ld`` elements apart (padded slots, ``pitch_for``);
        ``out``: an existing (T, S) block of the same shape to fill instead (a ``SlotPool`` view)."""
        host = np.ascontiguousarray(host, dtype=dtype)
        if out is not None:
            assert out.ndim == 2 and tuple(host.shape) == out.shape and out.dtype == host.dtype, (host.shape, out.shape)
            es = out.dtype.itemsize
            if host.size:
                check(self.lib.atl_copy_2d(self.handle, out.ptr, (out.ld or out.shape[1]) * es, host.ctypes.data,
                                           host.shape[1] * es, host.shape[1] * es, host.shape[0], 0, 0))
            return out
        if ld is not None and host.ndim == 2 and int(ld) > host.shape[1]:
            a = self.empty_pitched(host.shape, int(ld), dtype)
            es = a.dtype.itemsize
            check(self.lib.atl_copy_2d(self.handle, a.ptr, a.ld * es, host.ctypes.data, host.shape[1] * es,
                                       host.shape[1] * es, host.shape[0], 0, 0))
            return a
        a = self.empty(host.shape, dtype)
        if host.size:
            check(self.lib.atl_upload(self.handle, a.ptr, host.ctypes.data, host.nbytes))
        return a

    def empty_pitched(self, shape, ld, dtype=np.float64):
        """(T, S) block with rows ``ld`` elements apart; the padding is zeroed (never read as data, but a stray NaN
        pattern in it would look alarming in a debugger)."""
        T, S = (int(v) for v in shape)
        ld = int(ld)
        assert ld >= S
        base = self.empty((max(T * ld, 1),), dtype)
        if ld > S:
            check(self.lib.atl_memset(self.handle, base.ptr, 0, base.nbytes))
            self.copy_after_compute()
        return DeviceArray(self, base.ptr, (T, S), dtype, owner=base, ld=ld) if ld > S else base.reshape(T, S)

    def _stride(self, S, *arrays):
        """How far apart the slots of this call's (T, S) input cubes are - the ``ld_cells`` argument of the C-ABI's ``*_ld``
        entry points (0 = contiguous): every 2-d input must have the same layout (all contiguous, or all padded to the same
        ``ld``)."""
        lds = {(a.ld if a.ld is not None else int(S)) for a in arrays
               if isinstance(a, DeviceArray) and a.ndim == 2 and a.shape[1] == int(S)}
        if len(lds) > 1:
            raise ValueError(f"the input cubes of one call mix slot strides {sorted(lds)}: upload them the same way "
                             "(Dataset.device pads every cube of a dataset alike; ATLITE_HIP_PITCH=0 switches padding off)")
        ld = lds.pop() if lds else int(S)
        return 0 if ld == int(S) else ld  # what the *_ld entry points take: 0 = contiguous

    def _relayout(self, a, ld):
        """Device copy of the (T, S) block ``a`` with slots ``ld`` elements apart (None: contiguous)."""
        T, S = a.shape
        out = self.empty_pitched((T, S), ld) if ld else self.empty((T, S))
        es = a.dtype.itemsize
        check(self.lib.atl_copy_2d(self.handle, out.ptr, (ld or S) * es, a.ptr, (a.ld or S) * es, S * es, T, 2, 0))
        return out

    def asdevice(self, x, dtype=np.float64):
        """DeviceArray as is; torch CUDA tensor zero-copy; anything else is uploaded."""
        if isinstance(x, DeviceArray):
            return x
        if getattr(x, "is_file_array", False):  # atlite_amd.io.FileArray: inflate + decode on the way in
            return x.to_device(self)
        if type(x).__module__.startswith("torch") and hasattr(x, "data_ptr"):
            if x.is_cuda:
                assert x.is_contiguous() and x.element_size() == np.dtype(dtype).itemsize
                return DeviceArray(self, x.data_ptr(), tuple(x.shape), dtype, owner=x)
            x = x.numpy()
        return self.upload(np.asarray(x), dtype)

    def sync(self):
        check(self.lib.atl_sync(self.handle))

    def copy_barrier(self):
        """Make the compute stream wait for everything enqueued on the copy stream so far."""
        ev = self.__dict__.get("_copy_ev")
        if ev is None:
            ev = C.c_void_p()
            check(self.lib.atl_event_create(self.handle, C.byref(ev)))
            self._copy_ev = ev
        check(self.lib.atl_event_record(self.handle, ev, 1))
        check(self.lib.atl_stream_wait_event(self.handle, 0, ev))

    def copy_after_compute(self):
        """Make the copy stream wait for everything enqueued on the compute stream so far (a block zeroed on the compute
        stream that the copy stream is about to fill)."""
        ev = self.__dict__.get("_compute_ev")
        if ev is None:
            ev = C.c_void_p()
            check(self.lib.atl_event_create(self.handle, C.byref(ev)))
            self._compute_ev = ev
        check(self.lib.atl_event_record(self.handle, ev, 0))
        check(self.lib.atl_stream_wait_event(self.handle, 1, ev))

    def name(self):
        buf = C.create_string_buffer(256)
        check(self.lib.atl_device_name(self.handle, buf, 256))
        return buf.value.decode()

    # -- timing ---------------------------------------------------------------------------
    def timer_start(self):
        check(self.lib.atl_timer_start(self.handle))

    def timer_stop(self):
        ms = C.c_float()
        check(self.lib.atl_timer_stop(self.handle, C.byref(ms)))
        return ms.value

    def set_profiling(self, on=True):
        """on: False / True, or an int n > 1 = keep the brackets of the n most recent launches."""
        check(self.lib.atl_set_profiling(self.handle, int(on)))
        self._profiling = int(on)

    def kernel_times(self, cap=1 << 16):
        """Durations (ms) of the most recent profiled launches, oldest first (synchronises)."""
        buf = (C.c_float * int(cap))()
        n = C.c_int64()
        check(self.lib.atl_kernel_times(self.handle, buf, int(cap), C.byref(n)))
        return np.asarray(buf[: n.value], dtype=np.float64)

    def last_kernel_ms(self):
        ms = C.c_float()
        check(self.lib.atl_last_kernel_ms(self.handle, C.byref(ms)))
        return ms.value

    # -- plans ----------------------------------------------------------------------------
    def plan(self, matrix, row_len=None, cache=True, ld=None, aligned=False):
        """
        Aggregation plan of a (N x S) matrix; row_len = X of the (Y, X) grid lets the plan use compact
        2-d cell tiles.  Plans are cached per context by matrix content (8 most recent), so repeated
        conversions over the same shapes skip the host-side preprocessing; cached plans are owned by
        the context (do not close them).  ``aligned=True``: the line-aligned plan for CONTIGUOUS cubes with
        S % 16 != 0 (``atl_agg_create_aligned``: pv with stored solar angles, wind, runoff, temperatures, spmm).
        """
        if not cache:
            return AggPlan(self, matrix, row_len=row_len, ld=ld, aligned=aligned)
        import scipy.sparse as sp

        m = matrix if sp.isspmatrix_csr(matrix) else sp.csr_matrix(matrix)
        digest = getattr(m, "_atl_digest", None)  # set on the matrices the indicator cache hands to the gateway (never copied out)
        if digest is None:
            try:
                import xxhash

                h = xxhash.xxh3_128()
            except Exception:  # pragma: no cover
                import hashlib

                h = hashlib.blake2b(digest_size=16)
            for a in (m.indptr, m.indices, m.data):
                h.update(np.ascontiguousarray(a).view(np.uint8))
            digest = h.hexdigest()
        key = (m.shape, int(row_len or 0), m.indptr.dtype.str, m.indices.dtype.str, digest,
               os.environ.get("ATLITE_HIP_TILE", ""), int(ld or 0), bool(aligned))
        cache_ = self.__dict__.setdefault("_plan_cache", {})
        if key in cache_:
            cache_[key] = cache_.pop(key)  # most recently used last
            return cache_[key]
        plan = AggPlan(self, m, row_len=row_len, ld=ld, aligned=aligned)
        cache_[key] = plan
        while len(cache_) > 8:
            cache_.pop(next(iter(cache_))).close()
        return plan

    # -- conversions (device in, device out) ------------------------------------------------
    def _out(self, plan, n_slots, S, time_agg, out=None):
        """-> (result array or None, pointer, row stride).  ``out=(ptr, ld)`` makes the call write
        into caller memory - a view of a larger result, used by the slab pipeline."""
        if out is not None:
            return None, int(out[0]), int(out[1])
        k = 2 if time_agg == "sum_count" else 1  # [sum | count]
        if plan is None:
            a = self.empty((n_slots, S)) if time_agg is None else self.empty((k * S,))
        else:
            N = plan.shape[0]
            a = self.empty((N, n_slots)) if time_agg is None else self.empty((k * N,))
        return a, a.ptr, max(n_slots, 1)

    def spmm(self, plan, dense, time_agg=None, out=None):
        T, S = dense.shape
        ldc = self._stride(S, dense)
        res, optr, ld = self._out(plan, T, S, time_agg, out)
        check(self.lib.atl_spmm_csr_ld(self.handle, ldc, plan.handle, dense.ptr, T, S, _TIME_CODES[time_agg], optr, ld))
        return res

    def pv(self, inputs: dict, params: dict, T, S, plan=None, time_agg=None, solar_tables=None, options=None,
           out=None):
        """
        inputs: name -> DeviceArray (T,S): influx_toa plus either influx_direct + influx_diffuse or
        influx; albedo or outflux; temperature; humidity (enhanced clearsky); solar_altitude +
        solar_azimuth unless ``solar_tables`` (host arrays ``sin_dec, cos_dec`` (T), ``h, cos_h``
        (T,X), ``sin_lat, cos_lat`` (Y)) are given.  params: panel constants + slope/azimuth in
        radians (scalars or per-cell).  options: tracking / trigon_model / clearsky_model /
        irradiation / panel_model (names as in atlite), solar thermal c0, c1, t_store_K.
        """
        keep = []
        options = dict(options or {})
        pin = _lib.PvInputs()
        for name in ("influx_direct", "influx_diffuse", "influx_toa", "albedo", "temperature", "solar_altitude",
                     "solar_azimuth", "influx", "outflux", "humidity"):
            if name in inputs and inputs[name] is not None:
                setattr(pin, "d_" + name, inputs[name].ptr)
        if solar_tables is not None:
            pin.d_solar_altitude = pin.d_solar_azimuth = None
            for field, key in (("d_sin_dec", "sin_dec"), ("d_cos_dec", "cos_dec"), ("d_hour_angle", "h"),
                               ("d_cos_hour_angle", "cos_h"), ("d_sin_lat", "sin_lat"), ("d_cos_lat", "cos_lat")):
                v = solar_tables[key]
                t = v if isinstance(v, DeviceArray) else self.upload(np.ascontiguousarray(v, dtype=np.float64))
                if t is not v:
                    keep.append(t)
                setattr(pin, field, t.ptr)
            pin.X = int(solar_tables["h"].shape[1])
        elif options.get("row_len"):  # X of the (Y, X) grid: lets the per-cell early-out kernels walk compact tiles
            pin.X = int(options["row_len"]) if S % int(options["row_len"]) == 0 else 0
        pp = _lib.PvParams()
        model = options.get("panel_model", params.get("model", "huld"))
        pp.panel_model = _lib.PANEL[model]
        if model == "huld":
            for k in ("c_temp_amb", "c_temp_irrad", "r_tmod", "r_irradiance", "k_1", "k_2", "k_3", "k_4", "k_5", "k_6"):
                setattr(pp, k, float(params[k]))
        else:
            pp.r_irradiance = 1.0
        if model == "bofinger":
            for k in ("A", "B", "C", "D", "NOCT", "Tstd", "Tamb", "Intc", "ta", "threshold"):
                setattr(pp, "bof_" + k, float(params[k]))
        if model == "solar_thermal":
            pp.st_c0, pp.st_c1, pp.st_t_store_K = (float(options[k]) for k in ("c0", "c1", "t_store_K"))
        pp.inverter_efficiency = float(params.get("inverter_efficiency", 1.0))
        pp.altitude_threshold = float(params.get("altitude_threshold", np.radians(1.0)))
        pp.tracking = _lib.TRACKING[options.get("tracking")]
        pp.trigon_model = _lib.TRIGON[options.get("trigon_model", "simple")]
        pp.clearsky_model = _lib.CLEARSKY[options.get("clearsky_model") or "simple"]
        pp.irradiation = _lib.IRRADIATION[options.get("irradiation", "total")]
        pp.night_skip = 1 if options.get("night_skip", os.environ.get("ATLITE_HIP_NIGHT_SKIP", "1") == "1") else 0
        slope, azimuth = params["slope"], params["azimuth"]
        if not isinstance(slope, DeviceArray) and np.ndim(slope) == 0 and np.ndim(azimuth) == 0:
            pp.slope, pp.azimuth = float(slope), float(azimuth)
            pp.d_cell_slope = pp.d_cell_azimuth = None
        else:
            # one pair per cell (S,) or - an orientation that follows the sun - per cell and time step (T, S)
            per_time = any(len(getattr(v, "shape", ())) == 2 for v in (slope, azimuth))
            shape = (T, S) if per_time else (S,)
            # an orientation cube is an input cube like the others: the same slot padding
            in_ld = next((v.ld for v in inputs.values() if isinstance(v, DeviceArray) and v.ndim == 2 and v.ld), None)
            if isinstance(slope, DeviceArray) and isinstance(azimuth, DeviceArray):
                ds, da = slope, azimuth
                if per_time and (ds.ld != in_ld or da.ld != in_ld):  # uploaded before the inputs' layout was known
                    ds, da = self._relayout(ds, in_ld), self._relayout(da, in_ld)
                    keep += [ds, da]
            else:
                ld_o = in_ld if per_time else None
                ds = self.upload(np.ascontiguousarray(np.broadcast_to(np.asarray(slope, dtype=np.float64), shape)), ld=ld_o)
                da = self.upload(np.ascontiguousarray(np.broadcast_to(np.asarray(azimuth, dtype=np.float64), shape)), ld=ld_o)
                keep += [ds, da]
            if tuple(ds.shape) != shape or tuple(da.shape) != shape:
                raise ValueError(f"orientation arrays must have shape {shape}, got {tuple(ds.shape)} / {tuple(da.shape)}")
            pp.d_cell_slope, pp.d_cell_azimuth = ds.ptr, da.ptr
            pp.orientation_per_time = 1 if per_time else 0
        ldc = self._stride(S, *[v for v in inputs.values() if v is not None], *([ds, da] if pp.orientation_per_time else []))
        if plan is not None and pp.night_skip and solar_tables is None:
            dm = self._day_map(inputs.get("solar_altitude"), pin, pp, T, S, plan, options, ldc)
            if dm is not None:
                pin.d_day_map, pin.day_map_ld = dm[0].ptr, dm[1]
        res, optr, ld = self._out(plan, T, S, time_agg, out)
        if plan is None:
            check(self.lib.atl_pv_convert_ld(self.handle, ldc, C.byref(pin), C.byref(pp), T, S, _TIME_CODES[time_agg], optr))
        else:
            check(self.lib.atl_pv_convert_aggregate_ld(self.handle, ldc, C.byref(pin), C.byref(pp), T, S, plan.handle,
                                                       _TIME_CODES[time_agg], optr, ld))
        if keep:
            self.sync()  # temporaries uploaded for this call must outlive the kernels
        return res

    def _day_map(self, alt, pin, pp, T, S, plan, options, ldc):
        """(map, ld) of the early-out's day bits for (plan, altitude cube, cut-off) - ``atl_pv_day_map`` - or None.  Built on
        first use and kept WITH THE CUBE'S ALLOCATION, so it lives exactly as long as the device copy it describes; only for
        cubes the library filled itself (``mark_static``: a caller's own device array may be rewritten between calls) unless
        ``options["day_map"]`` says the caller vouches for it.  ``ATLITE_HIP_DAY_MAP=0`` switches the maps off."""
        want = options.get("day_map")
        if alt is None or want is False or plan.aligned or T == 0 or os.environ.get("ATLITE_HIP_DAY_MAP", "1") == "0":
            return None
        root = root_block(alt)
        if not (want or getattr(root, "_static", False)):
            return None
        maps = root.__dict__.setdefault("_day_maps", {})
        key = (alt.ptr, alt.ld, T, S, id(plan), float(pp.altitude_threshold))
        hit = maps.get(key)
        if hit is not None and hit[0] is plan:
            return hit[1], hit[2]
        ld = (T + 7) // 8 * 8  # one byte per (tile, time step): a bit per 128-byte line of the tile (round 6)
        n_tiles = plan.info()["n_segments"]
        dmap = self.empty((max(n_tiles, 1) * ld,), np.uint8)
        check(self.lib.atl_pv_day_map_ld(self.handle, ldc, C.byref(pin), C.byref(pp), T, S, plan.handle, dmap.ptr, ld))
        while len(maps) >= 4:  # a handful of plans per cutout
            maps.pop(next(iter(maps)))
        maps[key] = (plan, dmap, ld)
        return dmap, ld

    def wind(self, wnd, aux, V, POWn, to_height, from_height, method, T, S, plan=None, time_agg=None, out=None):
        """V / POWn: the power curve (POW / P); V = None: no power curve - the extrapolated wind speed itself."""
        V = np.ascontiguousarray(V if V is not None else [], dtype=np.float64)
        POWn = np.ascontiguousarray(POWn if POWn is not None else [], dtype=np.float64)
        win = _lib.WindInputs(
            wnd.ptr, aux.ptr if aux is not None else None, 1 if (aux is not None and aux.ndim == 1) else 0
        )
        wp = _lib.WindParams(
            {None: _lib.WIND_NONE, "logarithmic": _lib.WIND_LOG, "power": _lib.WIND_POWER}[method],
            float(to_height),
            float(from_height),
            len(V),
            V.ctypes.data_as(_lib.c_double_p),
            POWn.ctypes.data_as(_lib.c_double_p),
        )
        ldc = self._stride(S, wnd, aux)
        res, optr, ld = self._out(plan, T, S, time_agg, out)
        if plan is None:
            check(self.lib.atl_wind_convert_ld(self.handle, ldc, C.byref(win), C.byref(wp), T, S, _TIME_CODES[time_agg], optr))
        else:
            check(self.lib.atl_wind_convert_aggregate_ld(self.handle, ldc, C.byref(win), C.byref(wp), T, S, plan.handle,
                                                         _TIME_CODES[time_agg], optr, ld))
        return res

    def thermo(self, var, T, S, offset=-273.15, fillna0=False, cop=None, plan=None, time_agg=None, out=None):
        """temperature family (var + offset [, fillna 0]) and, with cop=(sink_T, c0, c1, c2), the COP."""
        tp = _lib.ThermoParams(float(offset), 1 if fillna0 else 0, 0 if cop is None else 1,
                               *(map(float, cop) if cop is not None else (0.0, 0.0, 0.0, 0.0)))
        ldc = self._stride(S, var)
        res, optr, ld = self._out(plan, T, S, time_agg, out)
        if plan is None:
            check(self.lib.atl_thermo_convert_ld(self.handle, ldc, var.ptr, C.byref(tp), T, S, _TIME_CODES[time_agg], optr))
        else:
            check(self.lib.atl_thermo_convert_aggregate_ld(self.handle, ldc, var.ptr, C.byref(tp), T, S, plan.handle,
                                                           _TIME_CODES[time_agg], optr, ld))
        return res

    def heat_demand(self, temperature, day_ptr, threshold_K, a, constant, T, S, plan=None, time_agg=None,
                    cooling=False, out=None):
        """day_ptr: host offsets (uploaded, call synchronises) or an int64 DeviceArray (D+1,)."""
        if isinstance(day_ptr, DeviceArray):
            d_ptr, D, fresh = day_ptr, day_ptr.shape[0] - 1, False
        else:
            day_ptr = np.ascontiguousarray(day_ptr, dtype=np.int64)
            D = len(day_ptr) - 1
            assert D >= 0 and day_ptr[0] >= 0 and day_ptr[-1] <= T and np.all(np.diff(day_ptr) >= 0)
            d_ptr, fresh = self.upload(day_ptr, np.int64), True
        hp = _lib.HeatParams(float(threshold_K), float(a), float(constant), D, d_ptr.ptr, 1 if cooling else 0)
        ldc = self._stride(S, temperature)
        res, optr, ld = self._out(plan, D, S, time_agg, out)
        if plan is None:
            check(self.lib.atl_heat_demand_convert_ld(self.handle, ldc, temperature.ptr, C.byref(hp), T, S,
                                                      _TIME_CODES[time_agg], optr))
        else:
            check(self.lib.atl_heat_demand_convert_aggregate_ld(self.handle, ldc, temperature.ptr, C.byref(hp), T, S,
                                                                plan.handle, _TIME_CODES[time_agg], optr, ld))
        if fresh:
            self.sync()  # d_ptr must outlive the kernels
        return res

    def runoff(self, runoff, height, T, S, plan=None, time_agg=None, out=None):
        ldc = self._stride(S, runoff)
        res, optr, ld = self._out(plan, T, S, time_agg, out)
        hptr = height.ptr if height is not None else None
        if plan is None:
            check(self.lib.atl_runoff_convert_ld(self.handle, ldc, runoff.ptr, hptr, T, S, _TIME_CODES[time_agg], optr))
        else:
            check(self.lib.atl_runoff_convert_aggregate_ld(self.handle, ldc, runoff.ptr, hptr, T, S, plan.handle,
                                                           _TIME_CODES[time_agg], optr, ld))
        return res

    # -- post-processing of a small (rows x time) result on the device (runoff: convert.py:1046-1082) ------------
    def rolling_mean(self, series, window, min_periods=1):
        """``rolling(time=window, min_periods=min_periods).mean()`` along the last axis of a (rows, T) DeviceArray."""
        rows, T = series.shape
        out = self.empty((rows, T))
        check(self.lib.atl_rolling_mean(self.handle, series.ptr, rows, T, series.ld or T, int(window), int(min_periods), out.ptr, T))
        return out

    def quantile(self, series, q):
        """``pd.Series(values.ravel()).quantile(q)`` of a (rows, T) DeviceArray: the two bracketing order statistics come
        from a radix select on the device; the virtual index and the interpolation between them are numpy's "linear"
        method as pandas calls it (``np.percentile(values, q * 100)``: the quantile is (q * 100) / 100)."""
        rows, T = series.shape
        q = float(np.true_divide(np.asarray(q, dtype=np.float64) * 100.0, 100))
        n, rank = C.c_int64(), C.c_int64()
        pair = (C.c_double * 2)()
        check(self.lib.atl_order_statistic(self.handle, series.ptr, rows, T, series.ld or T, q, C.byref(n), pair, C.byref(rank), None))
        if n.value == 0:
            return float("nan")
        a, b = pair[0], pair[1]
        t = q * (n.value - 1) - rank.value  # gamma = virtual index - floor(virtual index)
        if b != b:  # the virtual index is the last element: numpy clips the upper neighbour to it
            b = a
        diff = b - a  # numpy/lib/_function_base_impl.py: _lerp (inf - inf = NaN there as here)
        return b - diff * (1.0 - t) if t >= 0.5 else a + diff * t

    def zero_below(self, series, threshold):
        """``where(series >= threshold, 0.0)`` in place."""
        rows, T = series.shape
        check(self.lib.atl_zero_below(self.handle, series.ptr, rows, T, series.ld or T, float(threshold)))
        return series

    def normalize_rows(self, series, time_mask, ref):
        """Row r of the (rows, T) series times ``ref[r]`` / its nan-skipping sum over the steps with ``time_mask[t]``, in place."""
        rows, T = series.shape
        m = self.upload(np.ascontiguousarray(time_mask, dtype=np.uint8), np.uint8)
        r = self.upload(np.ascontiguousarray(ref, dtype=np.float64))
        assert m.size == T and r.size == rows
        check(self.lib.atl_normalize_rows(self.handle, series.ptr, rows, T, series.ld or T, m.ptr, r.ptr))
        self.sync()  # the two small uploads must outlive the kernel
        return series

    # -- synthetic fields -------------------------------------------------------------------
    def synth_field(self, kind, seed, var_id, p0, p1, T, S, per_cell_static=False):
        out = self.empty((S,) if per_cell_static else (T, S))
        check(self.lib.atl_synth_field(self.handle, kind, seed, var_id, p0, p1, 1 if per_cell_static else 0,
                                       1 if per_cell_static else T, S, out.ptr))
        return out

    def close(self):
        for p in self.__dict__.pop("_plan_cache", {}).values():
            p.close()
        self.__dict__.pop("_slab_cache", None)  # (streaming.run's slab buffers)
        if getattr(self, "handle", None):
            self.trim()
            for pair in self.__dict__.get("_pool_st", {}).get("events", []):
                for ev in pair:
                    self.lib.atl_event_destroy(ev)
            self.__dict__.pop("_pool_st", None)
        for key in ("_copy_ev", "_compute_ev"):
            ev = self.__dict__.pop(key, None)
            if ev is not None:
                self.lib.atl_event_destroy(ev)
        if getattr(self, "handle", None):
            self.lib.atl_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default = {}
_default_lock = threading.Lock()


def default_context(device=None):
    """Process-wide context per device (LOCAL_RANK selects the device under torchrun)."""
    import os

    if device is None:
        device = int(os.environ.get("ATLITE_HIP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    with _default_lock:
        if device not in _default:
            _default[device] = Context(device)
        return _default[device]
