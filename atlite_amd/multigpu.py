"""
Single-process multi-GPU execution of the hot path behind the public API: ``Cutout.pv()`` (and every
other conversion) shards the TIME axis across the GPUs of one node by itself.

The reference's only parallel axis is its dask time chunking (atlite/cutout.py:143; per-chunk
aggregation in atlite/aggregate.py:21-32): every time step (every calendar day for heat / cooling
demand) converts and aggregates independently.  Here one host thread + one ``Context`` (device,
stream, plan cache, scratch) per GPU each take a contiguous time shard of the dataset
(``Dataset.isel_time`` - views, nothing is copied; host / file data is uploaded by its own device's
thread, so the PCIe links work in parallel), run the same fused kernels on it, and the small
(shapes x time) result is reassembled with ONE collective of the library's own RCCL communicators
(C ABI ``atl_comm_*``, created in-process, one rank per device):

* ``aggregate_time=None`` with a matrix -> ragged all-gather along time (``atl_allgather_time_v``)
* per-cell ``"sum"`` / ``"mean"``       -> all-reduce of per-shard (sum, count) (``atl_allreduce_sum``)
* per-cell series (time, y, x)         -> no exchange at all: shard r's rows are a contiguous block
                                          of the host result, each device downloads its own

Select the devices with ``ATLITE_HIP_DEVICES=0,1,...,7``, ``atlite_amd.set_devices([...])`` or
``Cutout(..., devices=[...])``.  Transport of the collective (``ATLITE_HIP_GATHER``): ``rccl`` (default for
distinct devices), ``p2p`` - the library's in-process transport, every rank pulling its peers' blocks with peer
copies on its own stream (``atl_comm_init_local``; the default for a device list with repeats such as
``[0, 0, 0]``, which RCCL cannot form a communicator for - how a one-GPU box runs the N-rank collective code,
ragged shards and all) - or ``host`` (blocks downloaded and placed by the host).  Whatever the transport, the
ranks call the same ``atl_allgather_time_v`` / ``atl_allreduce_sum``.  A rank that fails aborts the group's
communicators on its way out, so its peers raise instead of waiting inside a collective.
"""

from __future__ import annotations

import atexit
import copy
import ctypes as C
import os
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _lib
from ._lib import check
from .device import Context, DeviceArray

_groups = {}
_groups_lock = threading.Lock()
_default_devices = None


def set_devices(devices):
    """Process-wide default device list for the conversions (None = single device)."""
    global _default_devices
    _default_devices = None if devices is None else [int(d) for d in devices]


def devices_for(cutout=None):
    """Device list in effect: the cutout's own, else ``set_devices``, else ``ATLITE_HIP_DEVICES``."""
    d = getattr(cutout, "devices", None)
    if d is None:
        d = _default_devices
    if d is None and os.environ.get("ATLITE_HIP_DEVICES"):
        d = [int(v) for v in os.environ["ATLITE_HIP_DEVICES"].split(",") if v.strip() != ""]
    return None if d is None else [int(v) for v in d]


def _close_groups():
    """At interpreter exit: communicators first, then their group, then the contexts they enqueue on - left to the
    garbage collector the order is arbitrary, and a communicator that outlives its context crashes in its destructor."""
    with _groups_lock:
        gs = list(_groups.values())
        _groups.clear()
    for g in gs:
        try:
            g.close()
        except Exception:
            pass


atexit.register(_close_groups)


def group(devices):
    """The (cached) DeviceGroup of a device list."""
    key = tuple(int(d) for d in devices)
    with _groups_lock:
        if key not in _groups:
            _groups[key] = DeviceGroup(key)
        return _groups[key]


class DeviceGroup:
    """One Context + one host thread per entry of ``devices`` (a single process, one node).
    ``ctxs`` / ``comm_factory`` are injection points for tests: ``comm_factory(group, rank)`` must return an object
    with ``gather_time_v`` / ``allreduce_sum`` / ``abort`` / ``close`` (``distributed.RcclComm``'s interface)."""

    def __init__(self, devices, ctxs=None, comm_factory=None):
        self.devices = tuple(int(d) for d in devices)
        self.n = len(self.devices)
        assert self.n >= 1
        self.ctxs = list(ctxs) if ctxs is not None else [Context(d) for d in self.devices]
        self.pool = ThreadPoolExecutor(self.n, thread_name_prefix="atlite-hip-dev")
        self.distinct = len(set(self.devices)) == self.n
        self._comm_factory = comm_factory
        self._comms = None
        self._comms_transport = None
        self._local_group = None

    def map(self, fn):
        """fn(rank) on every rank's thread, concurrently (ctypes calls release the GIL).  The first failure aborts
        the group's communicators - peers waiting inside a collective for the failed rank raise instead of hanging -
        and is re-raised once every rank has returned."""
        failures = []  # in the order they happened: the first one is the cause, the rest are its victims

        def guarded(r):
            try:
                return fn(r)
            except BaseException as e:  # noqa: BLE001 - re-raised by map()
                failures.append(e)
                for c in self._comms or ():
                    try:
                        c.abort()
                    except Exception:
                        pass
                raise

        futs = [self.pool.submit(guarded, r) for r in range(self.n)]
        res = []
        for f in futs:
            try:
                res.append(f.result())
            except BaseException:  # noqa: BLE001
                if not failures:  # not a worker's exception: KeyboardInterrupt / SystemExit of the waiting thread itself
                    for c in self._comms or ():
                        try:
                            c.abort()
                        except Exception:
                            pass
                    raise
                res.append(None)
        if failures:
            self._drop_comms()  # an aborted group stays aborted: the next call builds a fresh one
            raise failures[0]
        return res

    @property
    def transport(self):
        """'rccl' | 'p2p' | 'host' (see the module docstring)."""
        if self.n == 1:
            return "host"
        if self._comm_factory is not None:
            return "custom"
        t = os.environ.get("ATLITE_HIP_GATHER", "").strip().lower() or ("rccl" if self.distinct else "p2p")
        if t not in ("rccl", "p2p", "host"):
            raise ValueError(f"ATLITE_HIP_GATHER={t!r}: expected 'rccl', 'p2p' or 'host'")
        if t == "rccl" and not self.distinct:
            t = "p2p"  # RCCL cannot put one GPU into a communicator twice
        return t

    @property
    def use_collective(self):
        return self.transport != "host"

    def comms(self):
        """The group's communicators, rank r on device r's context (the rendezvous of either transport needs all
        ranks inside the call at once, hence one thread each)."""
        t = self.transport
        if self._comms is not None and self._comms_transport != t:  # $ATLITE_HIP_GATHER changed between calls
            self._drop_comms()
        if self._comms is None:
            from .distributed import LocalComm, LocalGroup, RcclComm

            self._comms_transport = t
            if t == "custom":
                self._comms = self.map(lambda r: self._comm_factory(self, r))
            elif t == "rccl":
                # ONE thread, one ncclGroupStart / ncclGroupEnd bracket for the N devices (atl_comm_init_all): N threads
                # each inside ncclCommInitRank on their own can wait for each other for good
                self._comms = RcclComm.init_all(self.ctxs)
            else:
                self._local_group = LocalGroup(self.n)
                try:
                    self._comms = self.map(lambda r: LocalComm(self.ctxs[r], self._local_group, r))
                except BaseException:
                    # a rank failed inside the constructor: nobody holds the half-built group, its peers were woken by
                    # the rendezvous' abort - drop it so that the next call starts afresh
                    self._comms = None
                    try:
                        self._local_group.close()
                    except Exception:
                        pass
                    self._local_group = None
                    raise
        return self._comms

    def _drop_comms(self):
        for c in self._comms or ():
            try:
                c.close()
            except Exception:
                pass
        self._comms = None
        if self._local_group is not None:
            try:
                self._local_group.close()
            except Exception:  # a communicator could not be detached: the group object is leaked, not reused
                pass
            self._local_group = None

    # -- data placement ------------------------------------------------------------------------
    def _shards(self, ds, edges):
        """Per-rank ``isel_time`` views of ``ds``, cached on the dataset so that a second conversion over
        the same cutout finds its shard already resident on its device."""
        cache = ds.__dict__.setdefault("_shard_cache", {})
        key = (self.devices, tuple(edges))
        if key not in cache:
            cache.clear()  # one partition at a time: do not pin several copies of a cutout in HBM
            cache[key] = [ds.isel_time(edges[r], edges[r + 1]) for r in range(self.n)]
        return cache[key]

    @staticmethod
    def _localize(sub_ds, ctx, names):
        """Device arrays that live on ANOTHER device are staged through the host (the executor is meant
        for host / file datasets and for per-device data; this keeps a foreign array correct)."""
        for name in names:
            if name not in sub_ds:
                continue
            la = sub_ds[name]
            d = la.data
            if isinstance(d, DeviceArray) and d.ctx.device != ctx.device:
                sub_ds[name] = type(la)(d.numpy(), la.dims, attrs=la.attrs, name=la.name)

    # -- execution -----------------------------------------------------------------------------
    def run(self, spec, ds, matrix, row_len, time_agg):
        """
        Execute ``spec`` over ``ds`` sharded along time.  Returns a HOST array:
        with a matrix the (N, slots) series (time reductions of that small array are the caller's);
        without one the (slots, S) cube for ``time_agg=None`` or the (S,) nan-skipping sum / mean.
        """
        from . import streaming
        from .convert import _execute

        T = len(ds.coords["time"])
        S = len(ds.coords["y"]) * len(ds.coords["x"])
        edges = spec.shard_edges(T, self.n)
        shards = self._shards(ds, edges)
        slots = [spec.out_slots(edges[r], edges[r + 1]) for r in range(self.n)]
        n_slots = spec.n_slots(ds)
        lens = [b - a for a, b in slots]
        assert slots[0][0] == 0 and slots[-1][1] == n_slots, (slots, n_slots)
        # an EMPTY calendar day sitting exactly on a shard boundary (gaps in the time axis) is claimed by both
        # neighbours (same NaN column twice): harmless for host placement, not expressible as an all-gather
        contiguous = all(slots[r][1] == slots[r + 1][0] for r in range(self.n - 1))
        names = tuple(getattr(spec, "time_vars", ())) + tuple(getattr(spec, "static_vars", ()))
        per_cell_reduce = matrix is None and time_agg in ("sum", "mean")

        def work(r):
            ctx, sub_ds = self.ctxs[r], shards[r]
            self._localize(sub_ds, ctx, names)
            sub = copy.copy(spec.for_slab(edges[r], edges[r + 1]))
            plan = ctx.plan(matrix, row_len=row_len, ld=getattr(sub_ds, "_slot_stride", lambda: None)()) if matrix is not None else None
            if edges[r + 1] == edges[r]:
                return None
            if per_cell_reduce:
                if time_agg == "sum" and streaming.wanted(sub_ds, sub):
                    out = streaming.run(ctx, sub, sub_ds, None, "sum")  # (S,) nan-skipping sum
                    cnt = None
                else:
                    sc = sub.run(ctx, sub_ds, None, "sum_count")  # [sum | count]
                    out, cnt = sc, True
                ctx.sync()
                return out, cnt
            out = _execute(ctx, sub, sub_ds, plan, None)
            ctx.sync()
            return out

        outs = self.map(work)

        if per_cell_reduce:
            return self._reduce_cells(outs, S, time_agg)
        if matrix is None:  # (slots, S) cube: shard r's rows are one contiguous block of the result
            res = np.empty((n_slots, S))

            def pull(r):
                if outs[r] is not None and lens[r]:
                    a, b = slots[r]
                    check(self.ctxs[r].lib.atl_download(self.ctxs[r].handle, res[a:b].ctypes.data, outs[r].ptr,
                                                        (b - a) * S * 8))

            self.map(pull)
            return res
        N = matrix.shape[0]
        if self.use_collective and contiguous:
            return self._gather_series(outs, N, lens)
        res = np.empty((N, n_slots))
        for r in range(self.n):
            if outs[r] is not None and lens[r]:
                res[:, slots[r][0]:slots[r][1]] = outs[r].numpy().reshape(N, lens[r])
        return res

    def _gather_series(self, outs, N, lens):
        """Ragged all-gather of the per-rank (N, lens[r]) blocks along time: every rank ends up with the whole
        (N, sum lens) series on its device (``atl_allgather_time_v``), rank 0 downloads it."""
        comms = self.comms()

        def gather(r):
            # comms[r].gather_time_v allocates the result BEFORE it enters the collective: an allocation failure
            # raises here (and map() aborts the group) instead of leaving the peers inside the all-gather
            full = comms[r].gather_time_v(outs[r], N, lens)
            return full.numpy() if r == 0 else self.ctxs[r].sync()

        return self.map(gather)[0]

    def _reduce_cells(self, outs, S, time_agg):
        """Global nan-skipping sum / mean over time from the per-shard (sum, count) (convert.py:51-56)."""
        have_counts = all(o is None or o[1] for o in outs)
        if time_agg == "mean":
            assert have_counts
        k = 2 if have_counts else 1
        if self.use_collective and have_counts and all(o is not None for o in outs):
            comms = self.comms()

            def red(r):
                buf = comms[r].allreduce_sum(outs[r][0])  # [sum | count], 2 S doubles, in place
                return buf.numpy() if r == 0 else self.ctxs[r].sync()

            tot = self.map(red)[0]
        else:
            tot = np.zeros(k * S)
            for o in outs:
                if o is not None:
                    tot[: o[0].size] += o[0].numpy().reshape(-1)[: k * S]
        if time_agg == "sum":
            return tot[:S]
        with np.errstate(invalid="ignore", divide="ignore"):
            return tot[:S] / tot[S:]

    def close(self):
        self._drop_comms()
        self.pool.shutdown(wait=True)
        for c in self.ctxs:
            c.close()
