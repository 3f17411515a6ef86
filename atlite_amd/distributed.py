"""
Multi-GPU execution of the hot path: one process per GPU, the TIME axis sharded.

Every time step (every calendar day for heat demand) converts and aggregates independently
(SURVEY.md 8e), so rank r owns a contiguous slab of time steps of every input cube plus the
whole (small) indicator matrix, runs the fused kernel on its slab, and the (shapes x time)
result is reassembled with ONE collective over RCCL/xGMI:

* ``aggregate_time=None``  -> all-gather of the (N x T_r) blocks      (``gather_time``)
* ``"sum"`` / ``"mean"``   -> all-reduce of per-rank (sum, count)      (``reduce_time``)

A rank takes its shard of any dataset - host arrays, device arrays or a cutout FILE - without
copying or reading the rest: ``ds.isel_time(*time_partition(T, world)[rank:rank + 2])``.

``torch.distributed`` is the transport ("nccl" = RCCL on ROCm; "gloo" in the CPU tests).
Stream ordering: create the ``Context`` on a NON-default torch stream and make it current
(``s = torch.cuda.Stream(); torch.cuda.set_stream(s); Context(dev, stream=s.cuda_stream)``), then
the collectives issued here are ordered after the kernels without a host sync (the default
stream's handle is 0, which ``atl_create`` reads as "create a private stream").
The reference has no distributed path (its parallelism is dask threads over time chunks,
atlite/cutout.py:143); this module is the MI355X-native counterpart of that chunking.
"""

from __future__ import annotations

import numpy as np


def time_partition(n_steps, world_size, align=1, first=0):
    """
    Edges ``e[0..world]`` of contiguous time shards, balanced to within ``align`` steps.
    ``align`` = 24 with ``first`` = number of steps before the first full day keeps calendar
    days intact for heat demand (shard boundaries fall on day boundaries of the shifted axis).
    """
    n_steps, world_size, align = int(n_steps), int(world_size), max(1, int(align))
    edges = [0]
    for r in range(1, world_size):
        e = round(n_steps * r / world_size)
        if align > 1:
            e = first + round((e - first) / align) * align
        edges.append(int(min(max(e, edges[-1]), n_steps)))
    edges.append(n_steps)
    return edges


def _dist():
    import torch.distributed as dist

    return dist


def gather_time(local, group=None, lens=None):
    """
    All-gather along the LAST axis: ``local`` is this rank's (..., T_r) block (torch tensor on
    the rank's device, or a NumPy array -> CPU tensor); returns the (..., sum T_r) result on
    every rank, same kind as the input.  Shards may have different lengths; pass the per-rank
    lengths as ``lens`` when they are known to skip the size exchange.
    """
    import torch

    dist = _dist()
    world = dist.get_world_size(group)
    is_np = isinstance(local, np.ndarray)
    t = torch.from_numpy(np.ascontiguousarray(local)) if is_np else local.contiguous()
    if world == 1:
        return local
    if lens is None:
        lens = torch.zeros(world, dtype=torch.int64, device=t.device)
        lens[dist.get_rank(group)] = t.shape[-1]
        dist.all_reduce(lens, group=group)
        lens = [int(v) for v in lens.tolist()]
    lead = t.shape[:-1]
    if len(set(lens)) == 1:
        # one collective into a (world, ..., T_r) buffer, then a local transpose-concat
        buf = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(buf.view(-1), t.view(-1), group=group)
        out = buf.movedim(0, -2).reshape(*lead, world * lens[0])
    else:
        # ragged shards: pad to the longest, gather once, trim
        tmax = max(lens)
        pad = torch.zeros(tuple(lead) + (tmax,), dtype=t.dtype, device=t.device)
        pad[..., : t.shape[-1]] = t
        buf = torch.empty((world,) + tuple(pad.shape), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(buf.view(-1), pad.view(-1), group=group)
        out = torch.cat([buf[r][..., : lens[r]] for r in range(world)], dim=-1)
    return out.numpy() if is_np else out


def reduce_time(local_sum, local_count, mean, group=None):
    """
    Combine per-rank nan-skipping time sums (and valid counts) of shape (N,) or (S,):
    returns the global sum, or sum / count for ``mean`` (convert.py:51-56 over the full axis).
    """
    import torch

    dist = _dist()
    is_np = isinstance(local_sum, np.ndarray)
    s = torch.from_numpy(np.ascontiguousarray(local_sum, dtype=np.float64)) if is_np else local_sum.contiguous()
    c = torch.from_numpy(np.ascontiguousarray(local_count, dtype=np.float64)) if is_np else local_count.contiguous()
    both = torch.stack([s, c])
    if dist.get_world_size(group) > 1:
        dist.all_reduce(both, group=group)
    out = both[0] / both[1] if mean else both[0]
    return out.numpy() if is_np else out


class RcclComm:
    """
    The library's own RCCL communicator (C ABI ``atl_comm_*``) for hosts that do not use
    torch.distributed: rank 0 calls ``RcclComm.unique_id()`` and ships the 128 bytes to the other
    ranks by any means; every rank then constructs ``RcclComm(ctx, n_ranks, rank, uid)``.
    """

    def __init__(self, ctx, n_ranks, rank, uid):
        import ctypes as C

        from ._lib import check

        assert len(uid) == 128
        self.ctx, self.n_ranks, self.rank = ctx, int(n_ranks), int(rank)
        buf = C.create_string_buffer(bytes(uid), 128)
        h = C.c_void_p()
        check(ctx.lib.atl_comm_init(ctx.handle, self.n_ranks, self.rank, buf, C.byref(h)))
        self.handle = h

    @classmethod
    def init_all(cls, ctxs):
        """The communicators of ``ctxs`` (distinct devices of THIS process), rank r on ``ctxs[r]``, created by one thread
        inside one ncclGroupStart / ncclGroupEnd bracket (``atl_comm_init_all``) - N threads each calling
        ``ncclCommInitRank`` on its own is the pattern that can wait for itself for good."""
        import ctypes as C

        from ._lib import check

        n = len(ctxs)
        hs = (C.c_void_p * n)(*[c.handle.value if hasattr(c.handle, "value") else c.handle for c in ctxs])
        out = (C.c_void_p * n)()
        check(ctxs[0].lib.atl_comm_init_all(hs, n, out))
        comms = []
        for r, ctx in enumerate(ctxs):
            c = cls.__new__(cls)
            c.ctx, c.n_ranks, c.rank, c.handle = ctx, n, r, C.c_void_p(out[r])
            comms.append(c)
        return comms

    def info(self):
        """What the communicator itself reports: ranks in it (ncclCommCount), this rank, its device ordinal, transport."""
        import ctypes as C

        from ._lib import check

        v = [C.c_int() for _ in range(4)]
        check(self.ctx.lib.atl_comm_info(self.handle, *[C.byref(x) for x in v]))
        return dict(n_ranks=v[0].value, rank=v[1].value, device=v[2].value, transport="rccl" if v[3].value == 0 else "local")

    def gather_time_v_async(self, local_ptr, N, lens, out_ptr, ld_out):
        """``gather_time_v`` on the communicator's own stream, ordered after the context's stream (the next step's kernel
        overlaps it); device pointers in, a ticket for ``wait`` out (``atl_allgather_time_v_async``)."""
        import ctypes as C

        from ._lib import check

        h_lens = (C.c_int64 * self.n_ranks)(*[int(v) for v in lens])
        t = C.c_int64()
        check(self.ctx.lib.atl_allgather_time_v_async(self.handle, local_ptr, N, h_lens, out_ptr, int(ld_out), C.byref(t)))
        return t.value

    def wait(self, ticket):
        """The context's stream waits for the asynchronous collective ``ticket`` (and all before it)."""
        from ._lib import check

        check(self.ctx.lib.atl_comm_wait(self.handle, int(ticket)))

    def sync(self):
        from ._lib import check

        check(self.ctx.lib.atl_comm_sync(self.handle))

    @staticmethod
    def unique_id():
        import ctypes as C

        from . import _lib

        buf = C.create_string_buffer(128)
        _lib.check(_lib.load().atl_comm_unique_id(buf))
        return bytes(buf.raw)

    def abort(self):
        if getattr(self, "handle", None):
            self.ctx.lib.atl_comm_abort(self.handle)

    def gather_time_v(self, local, N, lens, out=None):
        """Ragged all-gather along time: ``local`` = this rank's DeviceArray (N, lens[rank]) (None for an empty
        shard) -> DeviceArray (N, sum lens), the blocks in rank order (``atl_allgather_time_v``)."""
        import ctypes as C

        from ._lib import check

        total = int(sum(lens))
        if out is None:
            out = self.ctx.empty((N, total)).no_recycle()  # written on the communicator's stream
        h_lens = (C.c_int64 * self.n_ranks)(*[int(v) for v in lens])
        if local is not None and hasattr(local, "no_recycle"):
            local.no_recycle()  # read on the communicator's stream
        check(self.ctx.lib.atl_allgather_time_v(self.handle, local.ptr if local is not None else None, N, h_lens,
                                                out.ptr, total))
        return out

    def gather_time(self, local):
        """local: DeviceArray (N, T_r), same T_r on every rank -> DeviceArray (N, n_ranks * T_r)."""
        from ._lib import check

        N, T_r = local.shape
        out = self.ctx.empty((N, self.n_ranks * T_r)).no_recycle()
        check(self.ctx.lib.atl_allgather_time(self.handle, local.ptr, N, T_r, out.ptr, self.n_ranks * T_r))
        return out

    def allreduce_sum(self, buf):
        from ._lib import check

        buf.no_recycle()  # reduced in place on the communicator's stream
        check(self.ctx.lib.atl_allreduce_sum(self.handle, buf.ptr, buf.size))
        return buf

    def close(self):
        if getattr(self, "handle", None):
            self.ctx.lib.atl_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LocalGroup:
    """Rendezvous object of the in-process transport (C ABI ``atl_comm_group_*``): create one, hand it to the
    ``LocalComm`` of every rank (one host thread each - all ranks must be inside the constructor at once)."""

    def __init__(self, n_ranks):
        import ctypes as C

        from . import _lib

        self.n_ranks = int(n_ranks)
        self.lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self.lib.atl_comm_group_create(self.n_ranks, C.byref(h)))
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            from . import _lib

            h, self.handle = self.handle, None
            _lib.check(self.lib.atl_comm_group_destroy(h))  # communicators still attached: the group leaks - say so


class LocalComm(RcclComm):
    """
    Communicator of the in-process transport: the ranks are host threads of this process (distinct devices, or one
    device shared by several ranks), every rank pulls its peers' blocks with peer copies on its own stream.  Same
    collectives as ``RcclComm``; the all-reduce adds in rank order, so every rank holds the same bits.
    """

    def __init__(self, ctx, group, rank):
        import ctypes as C

        from ._lib import check

        self.ctx, self.n_ranks, self.rank, self.group = ctx, group.n_ranks, int(rank), group
        h = C.c_void_p()
        check(ctx.lib.atl_comm_init_local(ctx.handle, group.handle, self.rank, C.byref(h)))
        self.handle = h
