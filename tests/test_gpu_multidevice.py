"""
GPU: the single-process multi-device executor (atlite_amd.multigpu) behind the public API.
A one-GPU box cannot form an RCCL communicator with more than one rank, so the device list repeats
device 0 ([0, 0], [0, 0, 0]): every rank has its own Context / stream / plan / host thread and its own
time shard, and the reassembly runs the N-rank collective code - atl_allgather_time_v (pack, gather, k_gather_place_v
for ragged ranks) and atl_allreduce_sum - over the library's in-process transport (atl_comm_init_local: every rank
pulls its peers' blocks with device copies on its own stream).  On a real node the same calls run over RCCL (the
n_ranks = 1 test below, the driver's multi-GPU bench).  Results must equal the single-device results and the
reference-generated golden vectors.
"""
import warnings
from pathlib import Path

import numpy as np
import pandas as pd
import pytest
import scipy.sparse as sp

import atlite_amd
from atlite_amd import Cutout, Dataset, multigpu
from tests import helpers as H

pytestmark = pytest.mark.gpu
G = Path(__file__).parent / "golden"
PV_VARS = ("influx_direct", "influx_diffuse", "influx_toa", "albedo", "temperature", "solar_altitude",
           "solar_azimuth")


def load(name):
    return dict(np.load(G / f"{name}.npz"))


def close(a, b, atol_scale=1e-12):
    a, b = np.asarray(a), np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=1e-10, atol=atol_scale * float(np.nanmax(np.abs(b))), equal_nan=True)


def cutout_from(g, names, devices=None, chunked=False):
    t = pd.DatetimeIndex(g["time"].astype("datetime64[ns]"))
    return Cutout(Dataset({k: g[k] for k in names}, dict(time=t, y=g["y"], x=g["x"]), chunked=chunked), devices=devices)


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
def test_pv_sharded_equals_single_device_and_golden(devices):
    g, gw = load("pv"), load("gateway_pv")
    S = len(g["y"]) * len(g["x"])
    M = sp.csr_matrix((gw["matrix_data"], gw["matrix_indices"], gw["matrix_indptr"]), shape=(5, S))
    kw = dict(panel="CSi", orientation={"slope": 30.0, "azimuth": 180.0})
    one, many = cutout_from(g, PV_VARS), cutout_from(g, PV_VARS, devices=devices)
    r = many.pv(matrix=M, aggregate_time=None, **kw)
    assert r.dims == ("dim_0", "time") and r.attrs["units"] == "MW"
    np.testing.assert_array_equal(r.values, one.pv(matrix=M, aggregate_time=None, **kw).values)  # same kernels, same bits
    close(r.values, gw["series_matrix"])
    close(many.pv(matrix=M, aggregate_time="mean", **kw).values, gw["mean_matrix"])
    close(many.pv(matrix=M, aggregate_time="sum", **kw).values, gw["sum_matrix"])
    r, cap = many.pv(matrix=M, per_unit=True, return_capacity=True, aggregate_time="mean", **kw)
    close(r.values, gw["pu_mean_matrix"])
    # per cell: series (host cube), mean and sum maps (per-shard [sum | count] combined)
    cells = many.pv(aggregate_time=None, **kw)
    assert cells.dims == ("time", "y", "x")
    close(cells.values, g["out_CSi_const30_180"])
    close(many.pv(aggregate_time="mean", **kw).values, gw["cells_mean"])
    close(many.pv(aggregate_time="sum", **kw).values, gw["cells_sum"])
    # chunked (dask-like) datasets keep their (time, dim) result layout
    cc = cutout_from(g, PV_VARS, devices=devices, chunked=True)
    r = cc.pv(matrix=M, aggregate_time=None, index=pd.Index(list("abcde"), name="bus"), **kw)
    assert r.dims == ("time", "bus")
    close(r.values.T, gw["series_matrix"])


def test_in_kernel_solar_position_tables_are_sharded_with_the_time_axis():
    p = load("pv")
    names = ("influx_direct", "influx_diffuse", "influx_toa", "albedo", "temperature")
    M = H.blob_matrix(4, len(p["y"]), len(p["x"]), seed=2)
    kw = dict(panel="CSi", orientation={"slope": 25.0, "azimuth": 90.0}, matrix=M, aggregate_time=None)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", DeprecationWarning)
        ref = cutout_from(p, names).pv(**kw).values
        got = cutout_from(p, names, devices=[0, 0, 0]).pv(**kw).values
    np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("shift", [0.0, 4.0, -5.0])
def test_heat_demand_shards_on_calendar_days(shift):
    g = load("heat_demand")
    c = cutout_from(g, ("temperature",), devices=[0, 0, 0])
    r = c.heat_demand(threshold=15.0, a=1.3, constant=0.2, hour_shift=shift, aggregate_time=None)
    close(r.values, g[f"out_shift{shift:+.0f}"], atol_scale=1e-9)
    S = len(g["y"]) * len(g["x"])
    M = sp.csr_matrix(np.ones((2, S)))
    ra = c.heat_demand(threshold=15.0, a=1.3, constant=0.2, hour_shift=shift, matrix=M, aggregate_time=None)
    ref = cutout_from(g, ("temperature",)).heat_demand(threshold=15.0, a=1.3, constant=0.2, hour_shift=shift, matrix=M,
                                                        aggregate_time=None)
    np.testing.assert_array_equal(ra.values, ref.values)


def test_wind_runoff_and_env_selection(monkeypatch):
    g = load("wind")
    c = cutout_from(g, ("wnd100m", "roughness", "wnd_shear_exp"))
    monkeypatch.setenv("ATLITE_HIP_DEVICES", "0,0")
    assert multigpu.devices_for(c) == [0, 0]
    close(c.wind(turbine="Vestas_V112_3MW", aggregate_time=None).values, g["out_Vestas_V112_3MW_logarithmic"])
    monkeypatch.delenv("ATLITE_HIP_DEVICES")
    atlite_amd.set_devices([0, 0, 0])
    try:
        close(c.wind(turbine="Vestas_V112_3MW", interpolation_method="power", aggregate_time=None).values,
              g["out_Vestas_V112_3MW_power"])
        r = load("runoff")
        cr = cutout_from(r, ("runoff", "height"))
        close(cr.runoff(aggregate_time=None).values, r["out_weighted"])
        M = sp.csr_matrix(np.ones((1, r["height"].size)))
        close(cr.runoff(matrix=M, aggregate_time=None).values[0], r["out_weighted"].reshape(len(r["time"]), -1).sum(1))
    finally:
        atlite_amd.set_devices(None)
    assert multigpu.devices_for(c) is None


def test_more_devices_than_time_steps_and_host_streaming(monkeypatch):
    g = load("runoff")
    T = len(g["time"])
    sub = {k: (v[:3] if k in ("runoff", "time") else v) for k, v in g.items()}
    c = cutout_from(sub, ("runoff", "height"), devices=[0] * 5)  # two ranks get an empty shard
    close(c.runoff(aggregate_time=None).values, g["out_weighted"][:3])
    M = sp.csr_matrix(np.ones((1, g["height"].size)))
    close(c.runoff(matrix=M, aggregate_time=None).values[0], g["out_weighted"][:3].reshape(3, -1).sum(1))
    # every rank runs the slab pipeline on its own shard of a HOST dataset
    monkeypatch.setenv("ATLITE_HIP_STREAM", "1")
    monkeypatch.setenv("ATLITE_HIP_SLAB_STEPS", "8")
    c = cutout_from(g, ("runoff", "height"), devices=[0, 0])
    close(c.runoff(matrix=M, aggregate_time=None).values[0], g["out_weighted"].reshape(T, -1).sum(1))
    close(c.runoff(aggregate_time="sum").values, np.nansum(g["out_weighted"], axis=0))
    close(c.runoff(aggregate_time="mean").values, np.nanmean(g["out_weighted"], axis=0))


def test_rccl_ragged_allgather_single_rank(ctx):
    """atl_allgather_time_v through the C ABI (one rank: RCCL's all-gather degenerates to a copy, the
    padding and placement code still runs)."""
    import ctypes as C

    from atlite_amd._lib import check
    from atlite_amd.distributed import RcclComm

    comm = RcclComm(ctx, 1, 0, RcclComm.unique_id())
    a = np.random.default_rng(0).random((7, 13))
    d = ctx.upload(a)
    out = ctx.zeros((7, 20))
    lens = (C.c_int64 * 1)(13)
    check(ctx.lib.atl_allgather_time_v(comm.handle, d.ptr, 7, lens, out.ptr, 20))
    ctx.sync()
    np.testing.assert_array_equal(out.numpy()[:, :13], a)
    comm.close()


@pytest.mark.parametrize("mode", [["--collective", "lib"], ["--collective", "lib", "--no-step-overlap"],
                                  ["--collective", "lib", "--graph"], ["--collective", "torch", "--pipeline", "2"]])
def test_bench_collective_branch_over_rccl_on_one_rank(mode):
    """bench.py's N > 1 step on a 1-rank RCCL communicator - what can be exercised of it with a single GPU: the library's
    own collective (atl_comm_init + atl_allgather_time_v_async on the communicator's stream behind the next step's
    kernel; the blocking form; the step's launches replayed as a hipGraph) and torch.distributed's (async all-gather on
    the process group's stream + strided placement copy).  The parity leg checks the reassembled result against the
    oracle."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--debug-rccl-self", *mode, "--steps", "3", "--warmup", "1",
                        "--T", "960", "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    j = json.loads(line)
    assert j["parity"]["ok"] and j["parity"]["max_rel_err"] < 1e-10, j["parity"]
    if "lib" in mode:
        assert j["multi_gpu"]["collective"].startswith("library") and j["multi_gpu"]["ranks_seen"] == [1], j["multi_gpu"]



def test_bench_prints_its_line_when_no_collective_works():
    """bench.py --gpus N decides on its collective BEFORE a cube is generated: a preflight forms the communicator and sends
    16 doubles per rank through it, and a ladder - the library's communicator, torch.distributed, no collective - is walked
    by all ranks together.  Here every rung is told to fail (two ranks on this one GPU, control plane over gloo): the run
    still ends with ONE line for n_gpus = 2, flagged collective_failed, with the ladder in it."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--debug-gloo-one-gpu", "--preflight-fail", "all",
                        "--steps", "2", "--warmup", "1", "--T", "960", "--no-cpu-baseline", "--no-extras"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    mg = j["multi_gpu"]
    assert j["n_gpus"] == 2 and mg["collective_failed"] is True and mg["collective"].startswith("NONE")
    assert [x["ok"] for x in mg["preflight"]] == [False, False] and len(mg["per_rank_kernel_ms"]) == 2
    assert j["value"] > 0 and j["ms_per_step"] > 0


def _bench_line(args, scale="0.02", timeout=1200):
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, ATL_BENCH_WORKLOAD_SCALE=scale)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(root / "bench.py"), *args], capture_output=True, text=True, env=env, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_multi_gpu_line_carries_the_three_workloads():
    """bench.py --gpus N (N > 1) prints ONE line that covers what BASELINE.json asks of 8 GPUs: C2 strong scaling (the line's
    own value), configs[3] (pv 8760x800x800, in-kernel solar position) and configs[4] (heat demand + runoff 35040x400x400,
    shards on calendar days) - each with per-rank kernel ms, the gather alone, the step with the gather inside it, the own block
    in place, all ranks agreeing on the gathered result and an ORACLE parity sample of every rank's own shard.  Two ranks on
    this one GPU, the collective over gloo on host copies, a fiftieth of the time axes."""
    j = _bench_line(["--gpus", "2", "--debug-gloo-one-gpu", "--steps", "2", "--warmup", "1", "--T", "960", "--no-cpu-baseline",
                     "--no-extras", "--workload-steps", "2"])
    assert j["n_gpus"] == 2 and j["value"] > 0
    assert j["multi_gpu"]["parity"]["ok_on_every_rank"] and j["multi_gpu"]["parity"]["max_rel_err_rank0"] < 1e-10  # C2 itself
    w = j["workloads"]
    assert set(w) == {"c4", "c5"}, w
    for name in ("c4", "c5"):
        e = w[name]
        assert "error" not in e, e
        assert e["n_gpus"] == 2 and e["shards"] == 2 and e["collective"] == "gloo-debug"
        assert e["own_block_in_place"] and e["ranks_agree_on_the_result"], e
        assert e["parity"]["ok"] and e["parity"]["ok_on_every_rank"], e["parity"]
        assert len(e["per_rank_kernel_ms"]) == 2 and all(v > 0 for v in e["per_rank_kernel_ms"])
        assert e["gather_ms"] > 0 and e["ms_per_step"] > 0 and e["value"] > 0
    assert all(v % 24 == 0 for v in w["c5"]["time_steps_per_gpu"]), w["c5"]["time_steps_per_gpu"]  # calendar days stay whole
    assert w["c5"]["parity"]["heat_demand"]["ok"] and w["c5"]["parity"]["runoff"]["ok"]


def test_bench_workloads_on_an_emulated_shard():
    """--emulate-shard 8 --workloads c4,c5 on one GPU: rank 0's shard of the 8-way split of both configurations (no collective),
    the figures the scaling rehearsal (profiles/r06_scale_rehearsal.json) predicts the 8-GPU speed-up from."""
    j = _bench_line(["--emulate-shard", "8", "--workloads", "c4,c5", "--steps", "2", "--warmup", "1", "--T", "960", "--no-cpu-baseline",
                     "--no-extras", "--workload-steps", "2"], scale="0.05")
    for name in ("c4", "c5"):
        e = j["workloads"][name]
        assert "error" not in e, e
        assert e["shards"] == 8 and e["n_gpus"] == 1 and e["collective"] == "none"
        assert e["parity"]["ok"] and e["own_block_in_place"], e


# ---- the N-rank collective code on ONE GPU: the in-process transport (atl_comm_init_local) ----------------------
def _local_ranks(n):
    """n Contexts on device 0 + their communicators of one local group (each built on its own thread: the rendezvous
    needs all ranks inside the constructor at once)."""
    from concurrent.futures import ThreadPoolExecutor

    from atlite_amd.device import Context
    from atlite_amd.distributed import LocalComm, LocalGroup

    ctxs = [Context(0) for _ in range(n)]
    grp = LocalGroup(n)
    pool = ThreadPoolExecutor(n)
    comms = list(pool.map(lambda r: LocalComm(ctxs[r], grp, r), range(n)))
    return ctxs, grp, comms, pool


@pytest.mark.parametrize("lens", [[5, 0, 9], [1095, 1096, 1095, 1094], [300, 300, 300], [1, 700, 2, 3, 257]])
def test_local_transport_ragged_allgather_and_ordered_allreduce(lens):
    """atl_allgather_time_v / atl_allreduce_sum with MORE THAN ONE rank on the device: pack, gather (every rank pulls
    its peers' blocks on its own stream), k_gather_place_v for rank >= 1, on every rank."""
    n, N = len(lens), 7
    rng = np.random.default_rng(sum(lens))
    blocks = [rng.normal(size=(N, m)) for m in lens]
    want = np.concatenate(blocks, axis=1)
    ctxs, grp, comms, pool = _local_ranks(n)
    try:
        def rank(r):
            local = ctxs[r].upload(blocks[r]) if lens[r] else None
            full = comms[r].gather_time_v(local, N, lens)
            again = comms[r].gather_time_v(local, N, lens)  # a second collective reuses events and scratch
            vec = ctxs[r].upload(np.arange(1000, dtype=np.float64) * (r + 1) + 0.1 * r)
            comms[r].allreduce_sum(vec)
            return full.numpy(), again.numpy(), vec.numpy()

        res = list(pool.map(rank, range(n)))
        tot = np.zeros(1000)
        for r in range(n):  # rank order: the same bits on every rank
            tot += np.arange(1000, dtype=np.float64) * (r + 1) + 0.1 * r
        for full, again, vec in res:
            np.testing.assert_array_equal(full, want)
            np.testing.assert_array_equal(again, want)
            np.testing.assert_array_equal(vec, tot)
        if len(set(lens)) == 1:  # equal shards: atl_allgather_time is the same code without padding
            eq = list(pool.map(lambda r: comms[r].gather_time(ctxs[r].upload(blocks[r])).numpy(), range(n)))
            for e in eq:
                np.testing.assert_array_equal(e, want)
    finally:
        for c in comms:
            c.close()
        grp.close()
        pool.shutdown()
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("lens", [[40, 40, 40], [5, 0, 9, 33]])
def test_async_allgather_on_the_communicators_own_stream(lens):
    """atl_allgather_time_v_async: the collective and its placement run on the communicator's own stream, ordered after
    the context's stream; atl_comm_wait orders the context's stream behind a ticket.  Several steps in flight with two
    send buffers by parity (what bench.py's N > 1 step does), N ranks on one GPU over the in-process transport; the
    communicator reports its size and transport (atl_comm_info)."""
    n, N, steps = len(lens), 6, 5
    rng = np.random.default_rng(11)
    blocks = [[rng.normal(size=(N, m)) for m in lens] for _ in range(steps)]
    ctxs, grp, comms, pool = _local_ranks(n)
    try:
        def rank(r):
            ctx, comm = ctxs[r], comms[r]
            assert comm.info() == dict(n_ranks=n, rank=r, device=0, transport="local")
            send = [ctx.empty((N, max(lens[r], 1))) for _ in range(2)]
            outs = [ctx.zeros((N, sum(lens))) for _ in range(steps)]
            tickets = [None, None]
            for s in range(steps):
                par = s & 1
                if tickets[par] is not None:
                    comm.wait(tickets[par])  # the gather that read send[par] two steps ago
                if lens[r]:
                    check(ctx.lib.atl_upload(ctx.handle, send[par].ptr, np.ascontiguousarray(blocks[s][r]).ctypes.data, blocks[s][r].nbytes))
                tickets[par] = comm.gather_time_v_async(send[par].ptr if lens[r] else None, N, lens, outs[s].ptr, sum(lens))
            comm.wait(tickets[(steps - 1) & 1])
            comm.sync()
            ctx.sync()
            return [o.numpy() for o in outs]

        from atlite_amd._lib import check

        res = list(pool.map(rank, range(n)))
        for got in res:
            for s in range(steps):
                np.testing.assert_array_equal(got[s], np.concatenate(blocks[s], axis=1))
    finally:
        for c in comms:
            c.close()
        grp.close()
        pool.shutdown()
        for c in ctxs:
            c.close()


def test_rccl_init_all_and_info(ctx):
    """atl_comm_init_all: the communicators of a device list from ONE thread inside ncclGroupStart / ncclGroupEnd (what
    multigpu.DeviceGroup uses for distinct devices); with the single device of this box: one rank, which RCCL itself
    confirms (ncclCommCount / ncclCommCuDevice through atl_comm_info).  A repeated device is refused before RCCL sees it."""
    from atlite_amd.device import Context
    from atlite_amd.distributed import RcclComm

    (comm,) = RcclComm.init_all([ctx])
    assert comm.info() == dict(n_ranks=1, rank=0, device=ctx.device, transport="rccl")
    a = np.random.default_rng(0).random((5, 37))
    np.testing.assert_array_equal(comm.gather_time(ctx.upload(a)).numpy(), a)
    out = ctx.zeros((5, 37))
    t = comm.gather_time_v_async(ctx.upload(a).ptr, 5, [37], out.ptr, 37)
    comm.wait(t)
    ctx.sync()
    np.testing.assert_array_equal(out.numpy(), a)
    comm.close()
    other = Context(ctx.device)
    with pytest.raises(ValueError, match="appears twice"):
        RcclComm.init_all([ctx, other])
    other.close()


def test_hipgraph_capture_replays_a_conversion(ctx):
    """atl_capture_begin / atl_capture_end / atl_graph_launch: a fused pv convert + aggregate call recorded once replays
    with the same bits; growing the scratch arena inside a capture is refused (run the sequence once first)."""
    import ctypes as C

    from atlite_amd._lib import check
    from oracle import atlite_oracle as orc

    T, Y, X, N = 96, 12, 20, 5
    ds = H.pv_dataset(T, Y, X, seed=4)
    M = H.blob_matrix(N, Y, X, seed=5)
    params = dict(H.CSI, slope=np.radians(30.0), azimuth=np.radians(180.0))
    dev = {k: ctx.upload(v) for k, v in ds.items()}
    plan = ctx.plan(M, row_len=X)
    out = ctx.zeros((N, T))
    ref = ctx.pv(dev, params, T, Y * X, plan=plan).numpy()  # (also settles the scratch arena)
    check(ctx.lib.atl_capture_begin(ctx.handle))
    ctx.pv(dev, params, T, Y * X, plan=plan, out=(out.ptr, T))
    g = C.c_void_p()
    check(ctx.lib.atl_capture_end(ctx.handle, C.byref(g)))
    ctx.sync()
    assert not out.numpy().any()  # recorded, not executed
    for _ in range(3):
        check(ctx.lib.atl_memset(ctx.handle, out.ptr, 0, out.nbytes))
        check(ctx.lib.atl_graph_launch(ctx.handle, g))
        ctx.sync()
        np.testing.assert_array_equal(out.numpy(), ref)
    check(ctx.lib.atl_graph_destroy(g))
    want = orc.aggregate_matrix(orc.convert_pv(ds, H.CSI, dict(slope=params["slope"], azimuth=params["azimuth"])), M)
    np.testing.assert_allclose(ref, want, rtol=1e-10, atol=1e-12 * np.abs(want).max())
    with pytest.raises(ValueError, match="no capture is open"):
        check(ctx.lib.atl_capture_end(ctx.handle, C.byref(C.c_void_p())))


def test_local_transport_times_out_instead_of_hanging(monkeypatch):
    """A rank that never reaches the collective: its peers get an error after $ATLITE_HIP_COMM_TIMEOUT_S."""
    from atlite_amd._lib import AtliteHipError

    monkeypatch.setenv("ATLITE_HIP_COMM_TIMEOUT_S", "1.5")
    ctxs, grp, comms, pool = _local_ranks(3)
    try:
        def rank(r):
            if r == 2:
                return "absent"
            try:
                comms[r].gather_time_v(ctxs[r].upload(np.ones((2, 4))), 2, [4, 4, 4])
            except (AtliteHipError, RuntimeError) as e:
                return str(e)
            return "no error"

        res = list(pool.map(rank, range(3)))
        assert res[2] == "absent"
        assert all(("did not reach the collective" in m) or ("aborted" in m) for m in res[:2]), res
    finally:
        for c in comms:
            c.close()
        grp.close()
        pool.shutdown()
        for c in ctxs:
            c.close()


def test_executor_uses_the_collective_for_repeated_devices_and_host_placement_on_request(monkeypatch):
    g, gw = load("pv"), load("gateway_pv")
    S = len(g["y"]) * len(g["x"])
    M = sp.csr_matrix((gw["matrix_data"], gw["matrix_indices"], gw["matrix_indptr"]), shape=(5, S))
    kw = dict(panel="CSi", orientation={"slope": 30.0, "azimuth": 180.0}, matrix=M, aggregate_time=None)
    monkeypatch.delenv("ATLITE_HIP_GATHER", raising=False)
    many = cutout_from(g, PV_VARS, devices=[0, 0, 0])
    grp = multigpu.group([0, 0, 0])
    assert grp.transport == "p2p"
    a = many.pv(**kw).values
    assert grp._comms is not None and len(grp._comms) == 3  # the ragged all-gather ran over three ranks
    cm = many.pv(aggregate_time="mean", panel="CSi", orientation={"slope": 30.0, "azimuth": 180.0}).values  # all-reduce
    monkeypatch.setenv("ATLITE_HIP_GATHER", "host")
    assert grp.transport == "host"
    b = many.pv(**kw).values
    np.testing.assert_array_equal(a, b)
    close(a, gw["series_matrix"])
    close(cm, gw["cells_mean"])
    close(many.pv(aggregate_time="mean", panel="CSi", orientation={"slope": 30.0, "azimuth": 180.0}).values, gw["cells_mean"])
