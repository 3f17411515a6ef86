"""
CPU, world_size=2 and 8, gloo: the time-sharded N>1 path - partitioning, all-gather reassembly of the
(shapes x time) result and the all-reduce form of aggregate_time sum/mean.  The per-rank compute
is injected (the oracle here; the HIP path on a GPU box), the collectives are the product's.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from atlite_amd import distributed as D


def test_time_partition():
    assert D.time_partition(8760, 8) == [0, 1095, 2190, 3285, 4380, 5475, 6570, 7665, 8760]
    e = D.time_partition(8760, 7)
    assert e[0] == 0 and e[-1] == 8760 and all(b > a for a, b in zip(e, e[1:]))
    assert max(np.diff(e)) - min(np.diff(e)) <= 1
    # day-aligned shards for heat demand with a +4 h shift: first (partial) day has 20 steps
    e = D.time_partition(35040, 8, align=24, first=20)
    assert all((x - 20) % 24 == 0 for x in e[1:-1]) and max(np.diff(e)) - min(np.diff(e)) <= 24
    assert D.time_partition(5, 8)[-1] == 5 and D.time_partition(0, 2) == [0, 0, 0]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, T, uneven, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import atlite_oracle as orc
        from tests import helpers as H

        Y, X, N = 6, 10, 4
        ds = H.pv_dataset(T, Y, X, seed=5)
        ds["temperature"][7, 3] = np.nan
        M = H.blob_matrix(N, Y, X, seed=6)
        ori = dict(slope=np.radians(30.0), azimuth=np.radians(180.0))
        edges = [0, T // 3, T] if uneven else D.time_partition(T, world)
        a, b = edges[rank], edges[rank + 1]
        local = orc.aggregate_matrix(orc.convert_pv({k: v[a:b] for k, v in ds.items()}, H.CSI, ori), M)
        full = D.gather_time(local)
        s = np.nansum(local, axis=1)
        c = np.sum(~np.isnan(local), axis=1).astype(float)
        tot = D.reduce_time(s, c, mean=False)
        avg = D.reduce_time(s, c, mean=True)
        # torch-tensor flavour (what the GPU path passes)
        full_t = D.gather_time(torch.from_numpy(local))
        if rank == 0:
            ref = orc.aggregate_matrix(orc.convert_pv(ds, H.CSI, ori), M)
            q.put((np.array_equal(full, ref), np.array_equal(full_t.numpy(), ref),
                   np.allclose(tot, orc.aggregate_time(ref, "sum", 1), rtol=1e-13, atol=0),
                   np.allclose(avg, orc.aggregate_time(ref, "mean", 1), rtol=1e-13, atol=0)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run_world(world, T, uneven):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, T, uneven, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res == (True, True, True, True)


@pytest.mark.parametrize("uneven", [False, True])
def test_gather_and_reduce_world2(uneven):
    _run_world(2, 50, uneven)


@pytest.mark.parametrize("T", [48, 50])
def test_gather_and_reduce_world8(T):
    """The node size north_star names: 8 ranks, equal (48 = 8 x 6) and ragged (50 -> 6/7-step) shards."""
    _run_world(8, T, False)
