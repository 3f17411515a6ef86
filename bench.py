#!/usr/bin/env python3
"""
bench.py - atlite convert+aggregate hot path on MI355X.

Metric (BASELINE.json): grid-cell-timesteps/sec of pv convert+aggregate, plus achieved HBM
GB/s of the dominant kernel.  Workload at N=1 = BASELINE.json configs[1]:
``Cutout.pv(panel='CSi', orientation fixed)`` on an 8760 x 200 x 200 synthetic ERA5-shaped
fp64 cutout, 100 random-polygon shapes.  One "step" = one full pass of the fused
convert+aggregate path over the whole cutout (inputs resident in HBM), producing the
(shapes x time) result.

N > 1 (launched by torch.distributed.run, one rank per GPU): the time axis is sharded - rank r
holds year r of an N-year cutout (8760 steps each, same grid and shapes) - and each step ends
with an RCCL all-gather that reassembles the (shapes x N*8760) result on every rank:
``scaling = "weak"``.  ``--scaling strong`` instead splits the single 8760-step year.

Prints ONE JSON line on rank 0.
"""

from __future__ import annotations

import argparse
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
BYTES_PER_CELL_STEP = 7 * 8  # pv fused convert+aggregate, ERA5 getter variant


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--T", type=int, default=8760)
    ap.add_argument("--Y", type=int, default=200)
    ap.add_argument("--X", type=int, default=200)
    ap.add_argument("--shapes", type=int, default=100)
    ap.add_argument("--shape-kind", choices=["tessellation", "star"], default="tessellation")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=3200, help="time steps of the CPU baseline sample")
    ap.add_argument("--night-skip", action="store_true",
                    help="enable the night early-out (not the default measurement: it reads fewer bytes than "
                         "the 56 B/cell the roofline figure assumes)")
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

    ori = dict(slope=np.radians(30.0), azimuth=np.radians(180.0))
    Tn = inputs_host["temperature"].shape[0]
    chunks = [(a, min(a + 100, Tn)) for a in range(0, Tn, 100)]

    def work(c):
        ds = {k: v[c[0] : c[1]] for k, v in inputs_host.items()}
        return orc.aggregate_matrix(orc.convert_pv(ds, CSI, ori), M, dask_branch=True)

    t0 = time.perf_counter()
    with ThreadPoolExecutor(n_threads) as ex:
        res = list(ex.map(work, chunks))
    dt = time.perf_counter() - t0
    return dt, np.concatenate(res, axis=0).T  # (N, T')


def main():
    a = parse()
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
    from atlite_amd import gis, synthetic
    from atlite_amd.device import Context

    dist = None
    if a.debug_gloo_one_gpu:
        local = 0
    if world > 1:
        import torch.distributed as dist

        torch.cuda.set_device(local)
        if a.debug_gloo_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    n_gpus = world
    assert a.gpus == n_gpus or world == 1, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    # One explicit (non-default) torch stream carries both our kernels and the RCCL collective, so
    # they are stream-ordered.  (The default stream's handle is 0 = "create your own" for atl_create.)
    stream = torch.cuda.Stream(device=local)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    ctx = Context(local, stream=stream.cuda_stream)

    T, Y, X, S = a.T, a.Y, a.X, a.Y * a.X
    if a.scaling == "weak":
        T_loc, off = T, rank * T
        T_total = T * world
    else:
        edges = [(T * r) // world for r in range(world + 1)]
        T_loc, off = edges[rank + 1] - edges[rank], edges[rank]
        T_total = T
        assert all(edges[r + 1] - edges[r] == T_loc for r in range(world)), "strong scaling needs world | T"
    inputs, coords = synthetic.pv_inputs(ctx, T_loc, Y, X, offset_hours=off)
    x, y = coords["x"], coords["y"]
    dx, dy = x[1] - x[0], y[1] - y[0]
    bounds = (x[0] - dx / 2, y[0] - dy / 2, x[-1] + dx / 2, y[-1] + dy / 2)
    polys = (gis.random_tessellation if a.shape_kind == "tessellation" else gis.random_star_polygons)(
        a.shapes, bounds, seed=42)
    M = gis.compute_indicatormatrix(x, y, polys)
    plan = ctx.plan(M, row_len=X)
    plan_info = plan.info()
    N = M.shape[0]
    params = dict(CSI, slope=np.radians(30.0), azimuth=np.radians(180.0))

    out_local = torch.empty((N, T_loc), dtype=torch.float64, device=f"cuda:{local}")
    from atlite_amd import distributed as D

    shard_lens = [T_loc] * world
    from atlite_amd import _lib
    import ctypes as C

    pin = _lib.PvInputs(*[inputs[k].ptr for k in synthetic.PV_VARS])
    pp = _lib.PvParams()
    for k, v in params.items():
        setattr(pp, k if k not in ("slope", "azimuth") else k, float(v))
    pp.d_cell_slope = pp.d_cell_azimuth = None
    pp.altitude_threshold = float(np.radians(1.0))
    pp.night_skip = 1 if a.night_skip else 0

    def step():
        _lib.check(ctx.lib.atl_pv_convert_aggregate(ctx.handle, C.byref(pin), C.byref(pp), T_loc, S,
                                                    plan.handle, 0, out_local.data_ptr(), T_loc))
        if world > 1 and a.debug_gloo_one_gpu:
            torch.cuda.current_stream().synchronize()
            return D.gather_time(out_local.cpu(), lens=shard_lens)
        if world > 1:
            return D.gather_time(out_local, lens=shard_lens)  # (N, world * T_loc) on every rank
        return out_local

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ctx.set_profiling(True)
    for _ in range(a.warmup):
        step()
    fence()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
        kernel_ms.append(ctx.last_kernel_ms())  # waits for this step's fused kernel only
    fence()
    dt = time.perf_counter() - t0
    if world > 1:  # the reassembled (shapes x all time steps) result holds this rank's block in place
        full = step()
        fence()
        assert tuple(full.shape) == (N, world * T_loc), full.shape
        mine = full[:, rank * T_loc:(rank + 1) * T_loc]
        assert torch.equal(mine.to(out_local.device), out_local), "all-gather misplaced this rank's block"
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if a.debug_gloo_one_gpu else f"cuda:{local}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / a.steps * 1e3
    cells = T_total * S
    value = cells / (dt / a.steps)

    k_ms = float(np.mean(kernel_ms))
    algo_bytes = BYTES_PER_CELL_STEP * T_loc * S
    achieved = algo_bytes / (k_ms * 1e-3) / 1e9
    traffic = None
    pmc = ROOT / "profiles" / "pmc_latest.json"
    if pmc.exists():
        try:
            j = json.loads(pmc.read_text())
            if j.get("workload") == f"pv_{T_loc}x{Y}x{X}_{N}shapes_{a.shape_kind}":
                traffic = j.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

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
            "workload": f"Cutout.pv(panel='CSi', orientation={{slope:30,azimuth:180}}) {T_total}x{Y}x{X} fp64, "
                        f"{N} {a.shape_kind} polygon shapes, aggregate_time=None",
            "parallelism": f"time-sharded x{world}" + (" + RCCL all-gather" if world > 1 else ""),
            "time_steps_per_gpu": T_loc,
            "night_skip": bool(a.night_skip),
            "cell_tile": f"{plan_info['tile_w']}x{plan_info['tile_h']}",
            "partial_rows": plan_info["n_partial_rows"],
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "k_fused_segred<PvConv>",
            "achieved": achieved,
            "peak": 8000.0,
            "unit": "GB/s",
            "frac": achieved / 8000.0,
            "traffic": traffic,
            "kernel_ms": k_ms,
            "kernel_ms_median": float(np.median(kernel_ms)),
            "kernel_ms_min": float(np.min(kernel_ms)),
            "algorithmic_bytes": algo_bytes,
        },
    }

    if rank == 0 and not a.no_parity:
        # parity of this very run: a spread of time steps (night, sunrise, noon, sunset)
        from oracle import atlite_oracle as orc

        # 10 winter + 10 summer days (every sunrise / sunset in them) + the last step: >= 240 steps
        sel = np.unique(np.clip(np.concatenate([np.arange(0, 240), np.arange(4000, 4240), [T_loc - 1]]), 0, T_loc - 1))
        got = out_local.cpu().numpy()[:, sel]
        host = {}
        for k in synthetic.PV_VARS:
            full = inputs[k]
            host[k] = np.stack([full.slab(int(t), int(t) + 1).numpy()[0] for t in sel])
        ref = orc.aggregate_matrix(
            orc.convert_pv(host, CSI, dict(slope=np.radians(30.0), azimuth=np.radians(180.0))), M)
        scale = np.abs(ref).max()
        err = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-12 * scale)
        result["parity"] = {"checked_steps": int(len(sel)), "max_rel_err": float(err.max()), "rtol": 1e-10,
                            "ok": bool(err.max() <= 1e-10)}

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
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

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
