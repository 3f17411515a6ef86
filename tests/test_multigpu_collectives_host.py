"""
CPU: the N > 1 reassembly code of the single-process multi-device executor (atlite_amd/multigpu.py), driven
with a host-memory communicator for >= 3 RAGGED ranks - no GPU, no RCCL:

* the placement step of ``atl_allgather_time_v`` through ``atl_gather_place_v_host``, the HOST INSTANTIATION of
  the index function the device kernel k_gather_place_v runs (atl_comm.hip: gather_placement), walked over the
  kernel's own grid: a placement bug for rank >= 1 shows here;
* ``DeviceGroup._gather_series`` / ``_reduce_cells`` with an injected communicator: every rank thread enters the
  collective, rank 0's result is what comes back, a failing rank aborts its peers instead of hanging them;
* the shard cache of a Dataset is dropped when a variable is replaced, ``bench.py --gpus N`` launches
  its own ranks.

The reference's counterpart is the concatenation dask performs over its time chunks (atlite/aggregate.py:21-32).
"""
import ctypes as C
import sys
import threading
from pathlib import Path

import numpy as np
import pandas as pd
import pytest

from atlite_amd import _lib, multigpu
from atlite_amd.labeled import Dataset

ROOT = Path(__file__).resolve().parent.parent


def place_v_host(gathered, lens, N, ld_out=None):
    lib = _lib.load()
    n_ranks = len(lens)
    total = int(sum(lens))
    ld = total if ld_out is None else ld_out
    out = np.full((N, ld), -7.0)
    h_lens = (C.c_int64 * n_ranks)(*lens)
    g = np.ascontiguousarray(gathered, dtype=np.float64)
    _lib.check(lib.atl_gather_place_v_host(g.ctypes.data, n_ranks, N, h_lens, out.ctypes.data, ld))
    return out


@pytest.mark.parametrize("lens", [[3, 5, 4], [1095, 1096, 1095, 1094], [7, 0, 9], [0, 0, 5], [300], [2] * 8,
                                  [257, 256, 255, 1, 513]])
def test_gather_placement_index_math_of_ragged_ranks(lens):
    rng = np.random.default_rng(len(lens) * 1000 + sum(lens))
    N, Tmax = 6, max(lens)
    blocks = [rng.normal(size=(N, n)) for n in lens]
    gathered = np.full((len(lens), N, Tmax), np.nan)  # the padding must never be placed
    for r, b in enumerate(blocks):
        gathered[r, :, : lens[r]] = b
    want = np.concatenate(blocks, axis=1)
    np.testing.assert_array_equal(place_v_host(gathered, lens, N), want)
    # a wider output row (ld_out > sum lens): the tail of every row is left alone
    out = place_v_host(gathered, lens, N, ld_out=sum(lens) + 5)
    np.testing.assert_array_equal(out[:, : sum(lens)], want)
    assert (out[:, sum(lens):] == -7.0).all()


def test_gather_placement_refuses_bad_arguments():
    lib = _lib.load()
    g = np.zeros((2, 3, 4))
    out = np.zeros((3, 8))
    bad = (C.c_int64 * 2)(4, -1)
    assert lib.atl_gather_place_v_host(g.ctypes.data, 2, 3, bad, out.ctypes.data, 8) == -1
    ok = (C.c_int64 * 2)(4, 4)
    assert lib.atl_gather_place_v_host(g.ctypes.data, 2, 3, ok, out.ctypes.data, 7) == -1  # ld_out too small
    assert lib.atl_gather_place_v_host(g.ctypes.data, 65, 3, ok, out.ctypes.data, 8) == -1
    assert lib.atl_gather_place_v_host(None, 2, 3, ok, out.ctypes.data, 8) == -1


# ---- a host-memory communicator with RcclComm's interface ------------------------------------------------------
class HostArray:
    def __init__(self, a):
        self.a = np.ascontiguousarray(a, dtype=np.float64)
        self.size = self.a.size
        self.shape = self.a.shape

    def numpy(self):
        return self.a


class FakeCtx:
    def __init__(self, device):
        self.device = device
        self.synced = 0

    def sync(self):
        self.synced += 1

    def close(self):
        pass


class HostWorld:
    """Shared state of the fake communicators: a barrier every rank must reach (a rank that never calls hangs its
    peers - which is what the real collectives do - unless the group is aborted)."""

    def __init__(self, n):
        self.n = n
        self.barrier = threading.Barrier(n, timeout=20.0)
        self.slots = [None] * n
        self.calls = []


class HostComm:
    def __init__(self, world, rank, fail_in_gather=False):
        self.w, self.rank, self.n_ranks = world, rank, world.n
        self.fail = fail_in_gather
        self.closed = False

    def _exchange(self, mine):
        self.w.slots[self.rank] = mine
        self.w.barrier.wait()
        got = list(self.w.slots)
        self.w.barrier.wait()
        return got

    def gather_time_v(self, local, N, lens, out=None):
        if self.fail:
            raise MemoryError("rank %d could not allocate its result" % self.rank)  # BEFORE entering the collective
        Tmax = max(lens)
        send = np.zeros((N, Tmax))  # pack: full-width block, zero padding (atl_allgather_time_v)
        if local is not None and lens[self.rank]:
            send[:, : lens[self.rank]] = local.numpy().reshape(N, lens[self.rank])
        recv = np.stack(self._exchange(send))  # [rank][N][Tmax]
        self.w.calls.append(("gather", self.rank))
        return HostArray(place_v_host(recv, list(lens), N))

    def allreduce_sum(self, buf):
        tot = np.zeros_like(buf.numpy())
        for v in self._exchange(buf.numpy().copy()):  # rank order: the local transport's rule
            tot += v
        buf.a[...] = tot
        self.w.calls.append(("reduce", self.rank))
        return buf

    def abort(self):
        self.w.barrier.abort()

    def close(self):
        self.closed = True


def make_group(n, fail_rank=None):
    world = HostWorld(n)
    grp = multigpu.DeviceGroup(list(range(n)), ctxs=[FakeCtx(d) for d in range(n)],
                               comm_factory=lambda g, r: HostComm(world, r, fail_in_gather=(r == fail_rank)))
    return grp, world


@pytest.mark.parametrize("lens", [[5, 7, 6], [4, 0, 9, 3], [11] * 8, [1, 2, 3, 4, 5]])
def test_device_group_gathers_ragged_shards_through_the_injected_communicator(lens):
    n, N = len(lens), 5
    rng = np.random.default_rng(sum(lens))
    blocks = [rng.normal(size=(N, m)) for m in lens]
    outs = [HostArray(b) if m else None for b, m in zip(blocks, lens)]
    grp, world = make_group(n)
    try:
        assert grp.transport == "custom" and grp.use_collective
        got = grp._gather_series(outs, N, lens)
        np.testing.assert_array_equal(got, np.concatenate(blocks, axis=1))
        assert sorted(world.calls) == [("gather", r) for r in range(n)]  # every rank entered the collective
        assert all(c.synced == 1 for c in grp.ctxs[1:]) and grp.ctxs[0].synced == 0
    finally:
        grp.close()
    assert all(c.closed for c in (grp._comms or [])) or grp._comms is None


def test_device_group_reduces_per_cell_sums_and_counts():
    n, S = 4, 37
    rng = np.random.default_rng(3)
    sums = [rng.normal(size=S) for _ in range(n)]
    cnts = [rng.integers(0, 9, size=S).astype(float) for _ in range(n)]
    for r in range(n):  # a cell that is NaN in every shard: nan-skipping sum 0 over 0 values, mean = NaN
        sums[r][5] = cnts[r][5] = 0.0
    outs = [(HostArray(np.concatenate([s, c])), True) for s, c in zip(sums, cnts)]
    grp, world = make_group(n)
    try:
        tot = grp._reduce_cells([(HostArray(o[0].a.copy()), True) for o in outs], S, "sum")
        np.testing.assert_array_equal(tot, sums[0] + sums[1] + sums[2] + sums[3])
        mean = grp._reduce_cells(outs, S, "mean")
        with np.errstate(invalid="ignore"):
            want = (sums[0] + sums[1] + sums[2] + sums[3]) / (cnts[0] + cnts[1] + cnts[2] + cnts[3])
        np.testing.assert_array_equal(mean[np.arange(S) != 5], want[np.arange(S) != 5])
        assert np.isnan(mean[5])
        assert sorted(c for c in world.calls if c[0] == "reduce") == sorted([("reduce", r) for r in range(n)] * 2)
    finally:
        grp.close()


def test_a_failing_rank_aborts_its_peers_instead_of_hanging_them():
    lens, N = [3, 4, 5], 2
    outs = [HostArray(np.ones((N, m))) for m in lens]
    grp, world = make_group(3, fail_rank=1)
    try:
        with pytest.raises(MemoryError, match="rank 1"):
            grp._gather_series(outs, N, lens)  # ranks 0 and 2 are inside the barrier when rank 1 raises
        assert grp._comms is None  # the aborted communicators are gone; the next call builds fresh ones
    finally:
        grp.close()


def test_transport_selection(monkeypatch):
    mk = lambda devs: multigpu.DeviceGroup(devs, ctxs=[FakeCtx(d) for d in devs])  # noqa: E731
    monkeypatch.delenv("ATLITE_HIP_GATHER", raising=False)
    g = mk([0, 1, 2])
    assert g.transport == "rccl"
    g2 = mk([0, 0, 0])
    assert g2.transport == "p2p"  # RCCL cannot hold one GPU twice: the in-process transport runs the N-rank code
    monkeypatch.setenv("ATLITE_HIP_GATHER", "p2p")
    assert g.transport == "p2p"
    monkeypatch.setenv("ATLITE_HIP_GATHER", "rccl")
    assert g2.transport == "p2p"
    monkeypatch.setenv("ATLITE_HIP_GATHER", "host")
    assert g.transport == "host" and not g.use_collective
    monkeypatch.setenv("ATLITE_HIP_GATHER", "carrier-pigeon")
    with pytest.raises(ValueError):
        g.transport
    assert mk([3]).transport == "host"
    for x in (g, g2):
        x.close()


def test_replacing_a_variable_drops_the_cached_time_shards():
    T, Y, X = 12, 2, 3
    t = pd.date_range("2013-01-01", periods=T, freq="h")
    ds = Dataset({"runoff": np.zeros((T, Y, X))}, dict(time=t, y=np.arange(Y, dtype=float), x=np.arange(X, dtype=float)))
    grp = multigpu.DeviceGroup([0, 0, 0], ctxs=[FakeCtx(0)] * 3)
    try:
        edges = [0, 4, 8, 12]
        first = grp._shards(ds, edges)
        assert grp._shards(ds, edges) is first  # cached
        ds["runoff"] = np.ones((T, Y, X))  # replaced: the old shards (and their device copies) are stale
        second = grp._shards(ds, edges)
        assert second is not first and float(np.asarray(second[1]["runoff"].data).min()) == 1.0
        ds["height"] = np.ones((Y, X))  # added
        assert grp._shards(ds, edges) is not second and "height" in grp._shards(ds, edges)[2]
        third = grp._shards(ds, edges)
        del ds["height"]
        assert grp._shards(ds, edges) is not third and "height" not in grp._shards(ds, edges)[0]
    finally:
        grp.close()


def test_bench_launches_its_own_ranks(monkeypatch):
    """``python bench.py --gpus 8`` must not silently run on one GPU: without WORLD_SIZE it re-runs its
    own command line under torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1."""
    import subprocess

    sys.path.insert(0, str(ROOT))
    import bench

    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env

        class R:
            returncode = 0

        return R()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5", "--debug-gloo-one-gpu"])
    assert bench.self_launch(bench.parse()) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-7:] == ["--gpus", "8", "--steps", "20", "--warmup", "5", "--debug-gloo-one-gpu"]
    assert Path(cmd[-8]).name == "bench.py" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # already a rank of a launched job, or one GPU: run in this process
    monkeypatch.setenv("WORLD_SIZE", "8")
    assert bench.self_launch(bench.parse()) is None
    monkeypatch.delenv("WORLD_SIZE")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "1"])
    assert bench.self_launch(bench.parse()) is None
    # more GPUs asked for than the node has (no GPU here): a clear message, not a 1-GPU run
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    with pytest.raises(SystemExit, match="--gpus 8 but this node exposes 0 GPU"):
        bench.self_launch(bench.parse())
