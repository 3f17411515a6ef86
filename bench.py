#!/usr/bin/env python3
"""
bench.py - atlite convert+aggregate hot path on MI355X.

Metric (BASELINE.json): grid-cell-timesteps/sec of pv convert+aggregate, plus achieved HBM GB/s of
the dominant kernel.  One "step" = one full pass of the fused convert+aggregate path over the whole
cutout (inputs resident in HBM), producing the (shapes x time) result on every rank.

  --config c2 (default)  BASELINE.json configs[1]: Cutout.pv(panel='CSi', orientation fixed) on an
                         8760 x 200 x 200 fp64 cutout, 100 polygon shapes, stored solar angles
                         (7 cubes, 56 B per cell-step).
  --config c4            BASELINE.json configs[3]: 8760 x 800 x 800, 500 shapes, in-kernel solar
                         position (5 cubes, 40 B per cell-step: 224 GB, so N = 1 fits in 288 GB and
                         every N runs the same kernel).

N > 1 (launched by torch.distributed.run, one rank per GPU): STRONG scaling by default - the fixed
workload's time axis is cut into N contiguous shards (the reference's own parallel axis: time chunks,
atlite/cutout.py:143, aggregate.py:21-32), each rank converts + aggregates its shard and an RCCL
all-gather over xGMI reassembles the (shapes x time) result on every rank, inside the timed region:
a step's all-gather and placement copy run behind the NEXT step's kernel (double-buffered, ordered by
events; the region ends with a device-wide synchronize, so the last step's result is in place when the
clock stops).  ``--no-step-overlap`` finishes every step before the next starts; ``--pipeline P`` cuts a
rank's shard into P sub-launches so that the all-gather of one piece overlaps the kernel of the next.
``--scaling weak`` gives every rank a whole 8760-step year instead.

Prints ONE JSON line on rank 0.  At N = 1 (config c2) the line also carries: the same workload with
the night early-out (the Python API's default), with BASELINE's overlapping star-convex polygons, the
end-to-end time of the public ``Cutout.pv()`` call, and the CPU baseline.
"""

from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

CSI = dict(c_temp_amb=1, c_temp_irrad=0.035, r_tmod=298, r_irradiance=1000, k_1=-0.017162, k_2=-0.040289,
           k_3=-0.004681, k_4=0.000148, k_5=0.000169, k_6=0.000005, inverter_efficiency=0.9)
ORI = dict(slope=np.radians(30.0), azimuth=np.radians(180.0))
CONFIGS = {
    # bytes per cell-step of the fused pv kernel = 8 B x cubes read
    "c2": dict(T=8760, Y=200, X=200, shapes=100, stored_angles=True, bytes_per_cell_step=7 * 8),
    "c4": dict(T=8760, Y=800, X=800, shapes=500, stored_angles=False, bytes_per_cell_step=5 * 8),
}
GEN_STEPS = 1095  # sub-shard of the synthetic generator when the solar angles are only scratch


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2")
    ap.add_argument("--T", type=int, default=None)
    ap.add_argument("--Y", type=int, default=None)
    ap.add_argument("--X", type=int, default=None)
    ap.add_argument("--shapes", type=int, default=None)
    ap.add_argument("--shape-kind", choices=["tessellation", "star"], default="tessellation")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--layout", choices=["interleaved", "separate"], default="interleaved",
                    help="where the input cubes lie in HBM: slot-interleaved in one allocation (what the library's own device "
                         "copies use, device.SlotPool) or one allocation per cube")
    ap.add_argument("--no-step-overlap", action="store_true",
                    help="N > 1: finish a step's all-gather and placement before the next step's kernel starts "
                         "(default: they run behind it, on the collective's and a side stream)")
    ap.add_argument("--pipeline", type=int, default=0,
                    help="sub-launches per step whose all-gathers overlap the next sub-launch (0 = auto: 1 at "
                         "N=1 and for shards below 1e8 cell-steps, else 2)")
    ap.add_argument("--emulate-shard", type=int, default=0, metavar="N",
                    help="single GPU: run rank 0's shard of an N-way strong-scaling run (no collective) and "
                         "report the per-step overhead budget")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip night-skip / star / API end-to-end legs")
    ap.add_argument("--cpu-steps", type=int, default=3200, help="time steps of the CPU baseline sample")
    ap.add_argument("--night-skip", action="store_true",
                    help="enable the night early-out in the MAIN measurement (it reads fewer bytes than the "
                         "56 B/cell the roofline figure assumes; always reported separately at N=1)")
    ap.add_argument("--debug-rccl-self", action="store_true",
                    help="testing only (one GPU, one process): open a 1-rank RCCL process group and route the step "
                         "through the collective branch (async all-gather on the group's stream, placement copy)")
    ap.add_argument("--debug-gloo-one-gpu", action="store_true",
                    help="testing only: all ranks share GPU 0 and the collective runs over gloo on host copies")
    return ap.parse_args()


def usable_cpus():
    """CPUs this process may use: the cgroup CPU quota if one is set, else os.cpu_count()."""
    n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(p))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, -(-q // p)))
        except Exception:
            pass
    return n


def cpu_baseline(inputs_host, M, n_threads):
    """Oracle (NumPy restatement of the reference's eager op sequence) over time chunks of 100 on a
    thread pool - mirrors chunks={'time': 100} + dask's threaded scheduler (atlite/cutout.py:143)."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import atlite_oracle as orc

    Tn = inputs_host["temperature"].shape[0]
    chunks = [(a, min(a + 100, Tn)) for a in range(0, Tn, 100)]

    def work(c):
        ds = {k: v[c[0]: c[1]] for k, v in inputs_host.items()}
        return orc.aggregate_matrix(orc.convert_pv(ds, CSI, ORI), M, dask_branch=True)

    t0 = time.perf_counter()
    with ThreadPoolExecutor(n_threads) as ex:
        res = list(ex.map(work, chunks))
    dt = time.perf_counter() - t0
    return dt, np.concatenate(res, axis=0).T  # (N, T')


def pmc_traffic(tag):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/pmc_latest.json: FETCH_SIZE x 2 [gfx950 wide-read correction] + WRITE_SIZE, separate passes)."""
    f = ROOT / "profiles" / "pmc_latest.json"
    try:
        j = json.loads(f.read_text())
        ent = j.get("workloads", {}).get(tag) or (j if j.get("workload") == tag else None)
        if ent:
            return ent.get("hbm_bytes_per_launch"), f"profiles/pmc_latest.json[{tag}] (rocprofv3 --pmc, not this run)"
    except Exception:
        pass
    return None, None


def generate_pv(ctx, synthetic, solar, _lib, T_loc, Y, X, off, stored_angles, interleaved=True):
    """This rank's (T_loc, S) input cubes generated in HBM - slot-interleaved in one allocation (the layout of the
    library's own device copies, device.SlotPool) or one allocation per cube; without stored angles the generator's two
    solar-angle outputs go to a reusable scratch and only the 5 cubes the kernel reads are kept."""
    from atlite_amd.device import SlotPool, pitch_for

    S = Y * X
    if stored_angles:
        inputs, coords = synthetic.pv_inputs(ctx, T_loc, Y, X, offset_hours=off, interleaved=interleaved)
        return inputs, coords["x"], coords["y"], None
    x, y = synthetic.grid_coords(Y, X)
    five = [k for k in synthetic.PV_VARS if not k.startswith("solar_")]
    if interleaved:
        pool = SlotPool(ctx, T_loc, S, five, pitch_for(S))
        big, ld = {k: pool.view(k) for k in five}, pool.ld
    else:
        big, ld = {k: ctx.empty((T_loc, S)) for k in five}, S
    g = max(1, min(GEN_STEPS * S // ld, T_loc))  # the scratch shares the cubes' slot stride: the same bytes either way
    alt, az = ctx.empty((g * ld,)), ctx.empty((g * ld,))
    _lib.check(ctx.lib.atl_set_slot_stride(ctx.handle, 0 if ld == S else ld))
    for a in range(0, T_loc, g):
        n = min(g, T_loc - a)
        t = synthetic.time_index(n, "2013-01-01", off + a)
        h, dec = solar.hour_angle(t, x, "-30min")
        doy, hour = np.asarray(t.dayofyear, float), np.asarray(t.hour, float)
        tseason = 283.15 + 12.0 * np.sin(2 * np.pi * (doy - 110.0) / 365.0) + 5.0 * np.sin(2 * np.pi * (hour - 9.0) / 24.0)
        tabs = [ctx.upload(v) for v in (np.sin(dec), np.cos(dec), h, np.radians(y), tseason)]
        s = _lib.SynthSolar(*[v.ptr for v in tabs], X, Y, 42 + 1000003 * (off + a))
        ptrs = [big[k].ptr + a * ld * 8 for k in five] + [alt.ptr, az.ptr]
        _lib.check(ctx.lib.atl_synth_pv_inputs(ctx.handle, C.byref(s), n, S, *ptrs))
        ctx.sync()
    _lib.check(ctx.lib.atl_set_slot_stride(ctx.handle, 0))
    del alt, az
    t = synthetic.time_index(T_loc, "2013-01-01", off)
    h, dec = solar.hour_angle(t, x, "-30min")
    lat = np.radians(y)
    tables = dict(sin_dec=np.sin(dec), cos_dec=np.cos(dec), h=h, cos_h=np.cos(h), sin_lat=np.sin(lat), cos_lat=np.cos(lat))
    tables = {k: ctx.upload(np.ascontiguousarray(v)) for k, v in tables.items()}
    return big, x, y, tables


def self_launch(a):
    """``python bench.py --gpus N`` without an external launcher: re-run this very command line under
    ``python -m torch.distributed.run`` (one rank per GPU, rendezvous on 127.0.0.1) and hand its exit code back.
    Returns None when the process is already a rank of a launched job (or N == 1)."""
    if a.gpus <= 1 or "WORLD_SIZE" in os.environ or a.emulate_shard:
        return None
    import socket
    import subprocess

    if not a.debug_gloo_one_gpu:
        import torch

        have = torch.cuda.device_count()
        if have < a.gpus:
            sys.exit(f"bench.py: --gpus {a.gpus} but this node exposes {have} GPU(s) (torch.cuda.device_count()); "
                     f"run with --gpus {max(have, 1)} or on a node with {a.gpus} GPUs")
    with socket.socket() as s:  # a free rendezvous port
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


def main():
    a = parse()
    rc = self_launch(a)
    if rc is not None:
        sys.exit(rc)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch

    if not (ROOT / "atlite_amd" / "lib" / "libatlite_hip.so").exists() and "ATLITE_HIP_LIB" not in os.environ:
        if rank == 0:
            import __graft_entry__

            __graft_entry__.build()  # fresh checkout: the library is a (git-ignored) build artefact
        while not (ROOT / "atlite_amd" / "lib" / "libatlite_hip.so").exists():
            time.sleep(1.0)
    from atlite_amd import _lib, gis, solar, synthetic
    from atlite_amd import distributed as D
    from atlite_amd.device import Context

    dist = None
    if a.debug_gloo_one_gpu:
        local = 0
    if a.debug_rccl_self and world == 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))
    if world > 1:
        import torch.distributed as dist

        torch.cuda.set_device(local)
        if a.debug_gloo_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    n_gpus = world
    assert a.gpus == n_gpus or (world == 1 and a.emulate_shard) or (world == 1 and a.gpus == 1), \
        f"--gpus {a.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    # One explicit (non-default) torch stream carries our kernels; the RCCL collectives run on the process
    # group's own stream, event-ordered against it by torch (async_op + wait(): no host sync).
    stream = torch.cuda.Stream(device=local)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    ctx = Context(local, stream=stream.cuda_stream)

    cfg = dict(CONFIGS[a.config])
    for k, v in (("T", a.T), ("Y", a.Y), ("X", a.X), ("shapes", a.shapes)):
        if v is not None:
            cfg[k] = v
    T, Y, X = cfg["T"], cfg["Y"], cfg["X"]
    S = Y * X
    bpc = cfg["bytes_per_cell_step"]
    parts = a.emulate_shard if (a.emulate_shard and world == 1) else world
    my = 0 if a.emulate_shard else rank
    if a.scaling == "weak" and not a.emulate_shard:
        edges = [T * r for r in range(world + 1)]
        T_total = T * world
    else:
        edges = D.time_partition(T, parts)
        T_total = T
    T_loc, off = edges[my + 1] - edges[my], edges[my]
    shard_lens = [edges[r + 1] - edges[r] for r in range(parts)]
    inputs, x, y, tables = generate_pv(ctx, synthetic, solar, _lib, T_loc, Y, X, off, cfg["stored_angles"],
                                       interleaved=a.layout == "interleaved")
    ld = next(iter(inputs.values())).ld or S  # cells between the slots of a cube (S: one allocation per cube)
    dx, dy = x[1] - x[0], y[1] - y[0]
    bounds = (x[0] - dx / 2, y[0] - dy / 2, x[-1] + dx / 2, y[-1] + dy / 2)

    def shapes_of(kind):
        polys = (gis.random_tessellation if kind == "tessellation" else gis.random_star_polygons)(cfg["shapes"], bounds, seed=42)
        return polys, gis.compute_indicatormatrix(x, y, polys, ctx=ctx)  # on the device, as Cutout.indicatormatrix does

    polys, M = shapes_of(a.shape_kind)
    plan = ctx.plan(M, row_len=X, ld=None if ld == S else ld)
    plan_info = plan.info()
    N = M.shape[0]

    pin = _lib.PvInputs()
    for k, v in inputs.items():
        setattr(pin, "d_" + k, v.ptr)
    if tables is not None:
        for field, key in (("d_sin_dec", "sin_dec"), ("d_cos_dec", "cos_dec"), ("d_hour_angle", "h"),
                           ("d_cos_hour_angle", "cos_h"), ("d_sin_lat", "sin_lat"), ("d_cos_lat", "cos_lat")):
            setattr(pin, field, tables[key].ptr)
        pin.X = X

    def pv_params(night_skip):
        pp = _lib.PvParams()
        for k, v in dict(CSI, **ORI).items():
            setattr(pp, k, float(v))
        pp.d_cell_slope = pp.d_cell_azimuth = None
        pp.altitude_threshold = float(np.radians(1.0))
        pp.night_skip = 1 if night_skip else 0
        return pp

    # ---- the step ---------------------------------------------------------------------------
    # auto: two sub-launches per step (the first one's all-gather overlaps the second) once a rank's shard is big
    # enough to pay for the extra launch - measured on a 1/8 shard of C2 (4.4e7 cell-steps): 0.417 ms with one
    # launch, 0.450 ms with two, against an all-gather of 7 MB that takes less than the difference
    equal = len(set(shard_lens)) == 1
    collective = parts > 1 or (a.debug_rccl_self and dist is not None)
    # RCCL path: the all-gather and the placement copy of a step run behind the NEXT step's kernel - a second set of
    # (piece, gather) buffers by step parity, the placement on a side stream, buffer reuse ordered by events; the timed
    # region ends with a device-wide synchronize, so every step's result is in place when the clock stops.  One
    # launch per step then: there is nothing left for sub-launches to hide.
    overlap = collective and equal and not (a.emulate_shard or a.debug_gloo_one_gpu or a.no_step_overlap)
    P = a.pipeline if a.pipeline > 0 else (1 if parts == 1 or overlap or T_loc * S < 1.0e8 else 2)
    P = max(1, min(P, T_loc // 8 or 1))
    pe = D.time_partition(T_loc, P)  # sub-launch edges inside this rank's shard
    assert equal or world == 1 or P == 1, "pipelined gather needs equal shards"
    full = torch.empty((N, sum(shard_lens)), dtype=torch.float64, device=dev)  # (shapes x all time steps)
    piece = [torch.empty((N, pe[i + 1] - pe[i]), dtype=torch.float64, device=dev) for i in range(P)]
    gbuf = [torch.empty((parts, N, pe[i + 1] - pe[i]), dtype=torch.float64, device=dev) for i in range(P)] if collective else None
    if overlap:
        piece2 = [piece, [torch.empty_like(t) for t in piece]]
        gbuf2 = [gbuf, [torch.empty_like(t) for t in gbuf]]
        side = torch.cuda.Stream(device=dev)
        placed = [[None] * P, [None] * P]  # event: the placement copy that last read (piece, gather)[parity][i] is done
        step_no = [0]
    cube_ptrs = {k: getattr(pin, k) for k in ("d_influx_direct", "d_influx_diffuse", "d_influx_toa", "d_albedo",
                                              "d_temperature", "d_solar_altitude", "d_solar_azimuth")}
    tab_ptrs = {k: getattr(pin, k) for k in ("d_sin_dec", "d_cos_dec", "d_hour_angle", "d_cos_hour_angle")}

    def pin_at(t0):
        """The input descriptor advanced to time step t0 of this rank's shard."""
        if t0 == 0:
            return pin
        q = _lib.PvInputs()
        C.memmove(C.byref(q), C.byref(pin), C.sizeof(pin))
        for k, p in cube_ptrs.items():
            if p:
                setattr(q, k, p + t0 * ld * 8)
        for k, p in tab_ptrs.items():
            if p:
                setattr(q, k, p + t0 * (X if "hour" in k else 1) * 8)
        return q

    pins = [pin_at(pe[i]) for i in range(P)]
    full3 = full.view(N, parts, T_loc) if equal else None

    def cabi_pv(pin_, pp, T_, out_ptr, ld_out):
        """The timed call.  The slot stride is call-scoped context state (the Python wrappers reset it after every op)."""
        _lib.check(ctx.lib.atl_set_slot_stride(ctx.handle, 0 if ld == S else ld))
        _lib.check(ctx.lib.atl_pv_convert_aggregate(ctx.handle, C.byref(pin_), C.byref(pp), T_, S, plan.handle, 0, out_ptr, ld_out))

    def launch(pp, i, out_t):
        cabi_pv(pins[i], pp, pe[i + 1] - pe[i], out_t.data_ptr(), out_t.stride(0))

    def step(pp):
        if not collective:
            for i in range(P):  # P == 1 unless asked otherwise: straight into the result
                launch(pp, i, full[:, pe[i]:pe[i + 1]] if P == 1 else piece[i])
                if P > 1:
                    full[:, pe[i]:pe[i + 1]].copy_(piece[i])
            return full
        if overlap:
            par = step_no[0] & 1
            step_no[0] += 1
            main = torch.cuda.current_stream()
            for i in range(P):
                if placed[par][i] is not None:
                    main.wait_event(placed[par][i])  # two steps back: long done, costs nothing
                launch(pp, i, piece2[par][i])
                w = dist.all_gather_into_tensor(gbuf2[par][i].view(-1), piece2[par][i].view(-1), async_op=True)
                with torch.cuda.stream(side):
                    w.wait()  # the side stream waits for the collective; the main stream goes on to the next launch
                    full3[:, :, pe[i]:pe[i + 1]].copy_(gbuf2[par][i].permute(1, 0, 2))
                    ev = torch.cuda.Event()
                    ev.record(side)
                    placed[par][i] = ev
            return full
        works = []
        for i in range(P):
            launch(pp, i, piece[i])
            if a.emulate_shard:
                works.append(None)
            elif a.debug_gloo_one_gpu:
                torch.cuda.current_stream().synchronize()
                host = torch.empty(gbuf[i].shape, dtype=torch.float64)
                dist.all_gather_into_tensor(host.view(-1), piece[i].cpu().view(-1))
                gbuf[i].copy_(host)
                works.append(None)
            else:  # RCCL on the group's stream, ordered after the kernel; the next launch overlaps it
                works.append(dist.all_gather_into_tensor(gbuf[i].view(-1), piece[i].view(-1), async_op=True))
        for i in range(P):
            if works[i] is not None:
                works[i].wait()  # stream-level wait
            if a.emulate_shard:
                full3[:, 0, pe[i]:pe[i + 1]].copy_(piece[i])
            else:  # [rank][N][Tc] blocks -> (N x rank x T_loc) in place, one strided copy
                full3[:, :, pe[i]:pe[i + 1]].copy_(gbuf[i].permute(1, 0, 2))
        return full

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(pp, steps, warmup):
        """-> (seconds over `steps` steps [max over ranks], per-launch kernel ms of the timed region)."""
        ctx.set_profiling(max(2, steps * P))
        for _ in range(warmup):
            step(pp)
        fence()
        ctx.set_profiling(max(2, steps * P))
        t0 = time.perf_counter()
        for _ in range(steps):
            step(pp)
        fence()
        dt = time.perf_counter() - t0
        k = ctx.kernel_times()
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if a.debug_gloo_one_gpu else dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, k

    pp_main = pv_params(a.night_skip)
    dt, kms = timed(pp_main, a.steps, a.warmup)
    assert len(kms) == a.steps * P, (len(kms), a.steps, P)
    ms_per_step = dt / a.steps * 1e3
    value = (T_loc if a.emulate_shard else T_total) * S / (dt / a.steps)
    k_step = kms.reshape(a.steps, P).sum(axis=1)  # fused-kernel time per step on this rank
    k_ms = float(k_step.mean())
    algo_bytes = bpc * T_loc * S
    achieved = algo_bytes / (k_ms * 1e-3) / 1e9
    kname = "k_fused_segred<PvConvT<%s>>" % ("stored angles" if cfg["stored_angles"] else "in-kernel solar position")
    tag = f"pv_{T_loc}x{Y}x{X}_{N}shapes_{a.shape_kind}" + ("" if cfg["stored_angles"] else "_sp") + \
          ("_nightskip" if a.night_skip else "")
    traffic, traffic_src = pmc_traffic(tag)

    # per-rank kernel time and the collective by itself (outside the timed region): what a rank's step is made of
    per_rank_kernel_ms = gather_ms = None

    def diagnostics():
        cdev = "cpu" if a.debug_gloo_one_gpu else dev
        mine = torch.tensor([k_ms], dtype=torch.float64, device=cdev)
        lst = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(lst, mine)
        per_rank = [float(v.item()) for v in lst]
        if a.debug_gloo_one_gpu:
            return per_rank, None
        reps = 10
        fence()
        t0 = time.perf_counter()
        for _ in range(reps):  # all-gather of every piece + the strided placement copy, nothing to hide behind
            for i in range(P):
                dist.all_gather_into_tensor(gbuf[i].view(-1), piece[i].view(-1))
                if full3 is not None:
                    full3[:, :, pe[i]:pe[i + 1]].copy_(gbuf[i].permute(1, 0, 2))
        fence()
        tg = torch.tensor([(time.perf_counter() - t0) / reps * 1e3], dtype=torch.float64, device=dev)
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        return per_rank, float(tg.item())

    if world > 1:
        try:  # diagnostics only (the same code path on every rank): nothing here may cost the run its result line
            per_rank_kernel_ms, gather_ms = diagnostics()
        except Exception as e:  # noqa: BLE001
            print(f"[bench] rank {rank}: multi-GPU diagnostics skipped: {e!r}", file=sys.stderr)

    # the reassembled result holds every rank's block in place
    if world > 1:
        res = step(pp_main)
        fence()
        own = torch.empty((N, T_loc), dtype=torch.float64, device=dev)
        cabi_pv(pin, pp_main, T_loc, own.data_ptr(), T_loc)
        torch.cuda.synchronize()
        assert torch.equal(res[:, edges[rank]:edges[rank + 1]], own), "all-gather misplaced this rank's block"
        chk = torch.stack([res.sum(), res.abs().sum()])
        chk = chk.cpu() if a.debug_gloo_one_gpu else chk
        lst = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(lst, chk)
        assert all(torch.equal(v, chk) for v in lst), "ranks disagree on the gathered result"

    result = {
        "metric": "grid-cell-timesteps/sec (pv convert+aggregate)",
        "value": value,
        "unit": "cell-timesteps/s",
        "n_gpus": n_gpus,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": a.scaling,
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"{a.config}: pv CSi slope 30 az 180, {T_total}x{Y}x{X} fp64, {N} {a.shape_kind} polygon shapes, "
                        f"aggregate_time=None, " + ("stored solar angles (7 cubes)" if cfg["stored_angles"]
                                                    else "in-kernel solar position (5 cubes)") +
                        "; timed call = atl_pv_convert_aggregate (C ABI) on a prebuilt plan, result left in HBM",
            "parallelism": f"time-sharded x{world}" + (f" + RCCL all-gather, {P} pipelined piece(s) per step" if world > 1 else "") +
                           (", gather + placement of a step behind the next step's kernel" if overlap and world > 1 else ""),
            "time_steps_per_gpu": T_loc,
            "night_skip": bool(a.night_skip),
            "layout": ("slot-interleaved: the cubes in ONE allocation, the variables of a time step side by side "
                       f"(cube v, slot t at base + (t * {len(inputs)} + v) * {ld // len(inputs)} cells) - the layout of the library's own "
                       "device copies (device.SlotPool, Dataset.device_group)") if a.layout == "interleaved"
                      else "one allocation per cube",
            "cell_tile": f"{plan_info['tile_w']}x{plan_info['tile_h']}",
            "partial_rows": plan_info["n_partial_rows"],
        },
        "roofline": {
            "bound": "hbm",
            "kernel": kname,
            "achieved": achieved,
            "peak": 8000.0,
            "unit": "GB/s",
            "frac": achieved / 8000.0,
            "traffic": traffic,
            "traffic_source": traffic_src,
            "kernel_ms": k_ms,
            "kernel_ms_median": float(np.median(k_step)),
            "kernel_ms_min": float(np.min(k_step)),
            "launches_per_step": P,
            "algorithmic_bytes": algo_bytes,
        },
    }
    if world > 1:
        result["multi_gpu"] = {
            "per_rank_kernel_ms": per_rank_kernel_ms,  # fused kernel(s) of one step on each rank's own shard
            "gather_ms": gather_ms,  # serial all-gather + placement of one step's result (max over ranks), untimed region
            "result_bytes": int(N * sum(shard_lens) * 8),
            "transport": "gloo on host copies (debug, all ranks on GPU 0)" if a.debug_gloo_one_gpu else "RCCL over xGMI",
        }
    if a.emulate_shard:
        # what one rank of an N-way strong-scaling run spends per step besides the collective
        result["emulated_shard"] = {
            "of": parts, "pieces": P, "step_ms": ms_per_step, "fused_kernel_ms": k_ms,
            "overhead_ms": ms_per_step - k_ms, "overhead_frac": (ms_per_step - k_ms) / ms_per_step,
            "note": "rank 0's shard of the strong-scaling split on one GPU: fused kernel(s) + k_combine + placement "
                    "copy + host launch path; no collective",
        }

    single = world == 1 and not a.emulate_shard
    if rank == 0 and single and not a.no_parity:
        # parity of this very run: a spread of time steps (night, sunrise, noon, sunset)
        from oracle import atlite_oracle as orc

        # 10 winter + 10 summer days (every sunrise / sunset in them) + the last step: >= 240 steps
        n_days = 240 if S <= 40000 else 24
        sel = np.unique(np.clip(np.concatenate([np.arange(0, n_days), np.arange(4000, 4000 + n_days), [T_loc - 1]]), 0, T_loc - 1))
        step(pp_main)
        torch.cuda.synchronize()
        got = full.cpu().numpy()[:, sel]
        host = {k: np.stack([v.slab(int(t), int(t) + 1).numpy()[0] for t in sel]) for k, v in inputs.items()}
        if tables is not None:
            al, az = orc.solar_position(synthetic.time_index(T_loc)[sel], x, y, "-30min")
            host["solar_altitude"], host["solar_azimuth"] = al.reshape(len(sel), S), az.reshape(len(sel), S)
        ref = np.concatenate([orc.aggregate_matrix(orc.convert_pv({k: v[i:i + 16] for k, v in host.items()}, CSI, ORI), M)
                              for i in range(0, len(sel), 16)], axis=1)
        scale = np.abs(ref).max()
        err = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-12 * scale)
        result["parity"] = {"checked_steps": int(len(sel)), "max_rel_err": float(err.max()), "rtol": 1e-10,
                            "ok": bool(err.max() <= 1e-10)}

    if rank == 0 and single and a.config == "c2" and not a.no_extras:
        ks = max(3, min(a.steps, 10))
        # warm-up launches of the side legs: after the host-side gap before each of them the shader clock needs ~30 ms
        # of load to settle, and the early-out kernel is issue-bound (profiles/r03_clock_per_launch.txt: 2.3 -> 1.97 ms
        # over the first eight launches of a burst); the headline measurement above keeps the caller's --warmup
        kw_ = 12
        # (1) the Python API's default: night early-out (bit-identical output, fewer bytes read)
        if not a.night_skip:
            ref_out = step(pp_main).clone()
            dts, kk = timed(pv_params(True), ks, kw_)
            same = bool(torch.equal(step(pv_params(True)), ref_out))
            tr, src = pmc_traffic(tag + "_nightskip")
            result["night_skip"] = {"ms_per_step": dts / ks * 1e3, "kernel_ms": float(kk.mean()),
                                    "value": T_total * S / (dts / ks), "bit_identical": same,
                                    "traffic": tr, "traffic_source": src,
                                    "achieved_on_traffic_GBps": (tr / (float(kk.mean()) * 1e-3) / 1e9) if tr else None}
        # (2) BASELINE's "random-polygon" shapes: overlapping star-convex polygons, cells may be uncovered
        if a.shape_kind == "tessellation":
            polys_s, M_s = shapes_of("star")
            plan_main, plan = plan, ctx.plan(M_s, row_len=X, ld=None if ld == S else ld)
            dts, kk = timed(pp_main, ks, kw_)
            info_s = plan.info()
            covered = int((np.asarray((M_s != 0).sum(0)).ravel() > 0).sum())
            tr, src = pmc_traffic(f"pv_{T_loc}x{Y}x{X}_{N}shapes_star")
            result["star_polygons"] = {
                "ms_per_step": dts / ks * 1e3, "kernel_ms": float(kk.mean()), "value": T_total * S / (dts / ks),
                "partial_rows": info_s["n_partial_rows"], "cell_tile": f"{info_s['tile_w']}x{info_s['tile_h']}",
                "covered_cells": covered, "max_shapes_per_cell": int(np.asarray((M_s != 0).sum(0)).max()),
                "algorithmic_bytes": bpc * T_loc * covered, "traffic": tr, "traffic_source": src,
                "achieved_GBps_on_covered_cells": bpc * T_loc * covered / (float(kk.mean()) * 1e-3) / 1e9,
                "achieved_on_traffic_GBps": (tr / (float(kk.mean()) * 1e-3) / 1e9) if tr else None,
            }
            plan = plan_main
        # (3) what a user of the drop-in API waits for: cutout.pv(...) on a device-resident Dataset
        from atlite_amd import Cutout, Dataset

        cut = Cutout(Dataset(dict(inputs), dict(time=synthetic.time_index(T_loc), y=y, x=x)))
        kw = dict(panel="CSi", orientation={"slope": 30.0, "azimuth": 180.0}, aggregate_time=None)

        def call(**k):
            t0 = time.perf_counter()
            r = cut.pv(**kw, **k)
            return (time.perf_counter() - t0) * 1e3, r

        cold, r0 = call(shapes=polys)  # indicator matrix + plan build + kernel + D2H + labelled result
        warm = min(call(shapes=polys)[0] for _ in range(3))  # plan cached (matrix content hash)
        warm_m = min(call(matrix=M)[0] for _ in range(3))
        same = bool(np.array_equal(np.asarray(r0.values), step(pp_main).cpu().numpy()))
        result["api_e2e_ms"] = {"call": "cutout.pv(panel='CSi', orientation={slope:30,azimuth:180}, shapes=polys, "
                                        "aggregate_time=None) -> host (shapes x time) labelled array; night early-out on",
                                "cold": cold, "warm": warm, "warm_matrix_given": warm_m,
                                "equals_timed_result": same}

        # (4) the same cubes in an allocation each (the layout a caller's own device arrays have, and the library's before
        # round 3): same kernel, same bytes, bit-identical result - the memory system alone makes the difference
        if a.layout == "interleaved" and not a.night_skip:
            sep = generate_pv(ctx, synthetic, solar, _lib, T_loc, Y, X, off, cfg["stored_angles"], interleaved=False)[0]
            plan_sep = ctx.plan(M, row_len=X)
            ctx.set_profiling(True)
            run_sep = lambda: ctx.pv(sep, dict(CSI, **ORI), T_loc, S, plan=plan_sep, options=dict(night_skip=False))  # noqa: E731
            ksep = []
            for i in range(kw_ + ks):
                r_sep = run_sep()
                if i >= kw_:
                    ksep.append(ctx.last_kernel_ms())
            ksep_ms = float(np.mean(ksep))
            result["separate_cubes"] = {"kernel_ms": ksep_ms, "achieved_GBps": algo_bytes / (ksep_ms * 1e-3) / 1e9,
                                        "frac": algo_bytes / (ksep_ms * 1e-3) / 1e9 / 8000.0,
                                        "bit_identical": bool(np.array_equal(r_sep.numpy(), step(pp_main).cpu().numpy())),
                                        "note": "one allocation per cube instead of the slot-interleaved one; same kernel and bytes"}
            del sep, plan_sep, r_sep

    if rank == 0 and single and not a.no_cpu_baseline and cfg["stored_angles"]:
        Tc = min(a.cpu_steps, T_loc)
        t_a = 4000 if T_loc >= 4000 + Tc else 0  # daytime-rich slab in summer
        host = {k: inputs[k].slab(t_a, t_a + Tc).numpy() for k in synthetic.PV_VARS}
        # one thread per 100-step chunk (dask's granularity), bounded by the host's cores and by RAM
        # (each worker holds ~25 chunk-sized temporaries)
        n_chunks = (Tc + 99) // 100
        try:
            avail = int([l for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0].split()[1]) * 1024
        except Exception:
            avail = 16 << 30
        by_mem = max(1, int(0.4 * avail / (25 * 100 * S * 8)))
        cores = max(1, min(usable_cpus(), n_chunks, by_mem))
        cdt = min(cpu_baseline(host, M, cores)[0] for _ in range(2))
        T1 = min(400, Tc)
        cdt1 = cpu_baseline({k: v[:T1] for k, v in host.items()}, M, 1)[0]
        result["cpu_baseline"] = {
            "value": Tc * S / cdt,
            "unit": "cell-timesteps/s",
            "cores": cores,
            "kind": "port",
            "sample": f"{Tc} of {T_loc} time steps (t={t_a}..{t_a + Tc}) of the same cutout and shapes; NumPy "
                      f"oracle over time chunks of 100 on a {cores}-thread pool ({cdt:.2f} s; the box exposes "
                      f"{os.cpu_count()} hardware threads, its cgroup grants {usable_cpus()} CPUs)",
            "single_thread_value": T1 * S / cdt1,
            "single_thread_sample": f"{T1} time steps, 1 thread ({cdt1:.2f} s)",
        }

    if rank == 0 and single and not a.no_cpu_baseline and cfg["stored_angles"] and "cpu_baseline" in result:
        # BASELINE.md variant C: the same chain on dask.array 2021.10.0 (the reference's minimum pin), chunks {"time": 100},
        # threaded scheduler - under the image's second interpreter, which is the one that has dask
        conda = "/opt/conda/bin/python3.9"
        if os.path.exists(conda):
            import subprocess

            try:
                r = subprocess.run([conda, str(ROOT / "tools" / "cpu_baseline_dask.py"), "800", str(usable_cpus())],
                                   capture_output=True, text=True, timeout=240)
                j = json.loads(r.stdout[r.stdout.index("{"):])
                c = j["C_dask_array_threads"]
                result["cpu_baseline"]["dask_array"] = {
                    "value": c["cell_steps_per_s"], "unit": "cell-timesteps/s", "cores": j["host_threads_used"],
                    "sample": f"800 x 200 x 200 slab, 100 shapes, dask.array {j['versions']['dask']} / numpy {j['versions']['numpy']} "
                              f"under {conda}, chunks time=100, threaded scheduler ({c['seconds']:.2f} s, {c['graph_tasks']} tasks); "
                              f"same interpreter, eager NumPy on 1 thread: {j['A_numpy_1_thread']['cell_steps_per_s']:.3g}",
                    "equals_numpy_chain_max_rel": j["C_equals_B_max_rel"]}
            except Exception as e:  # noqa: BLE001 - a second opinion, never a reason to lose the line
                result["cpu_baseline"]["dask_array"] = {"skipped": repr(e)[:200]}

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
