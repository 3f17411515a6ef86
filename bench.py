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

The collective of N > 1 is the LIBRARY'S OWN (``--collective lib``, the default): ``atl_comm_init`` +
``atl_allgather_time_v_async`` of the C ABI (RCCL on the communicator's own stream; torch.distributed only ships the
128-byte unique id and takes the max of the ranks' clocks); ``--collective torch`` gathers with
``torch.distributed.all_gather_into_tensor`` instead.

Prints ONE JSON line on rank 0.  At N = 1 (config c2) the line also carries: the same workload with
the night early-out (the Python API's default), with BASELINE's overlapping star-convex polygons, the
end-to-end time of the public ``Cutout.pv()`` call, the CPU baseline, and - ``configs`` - the N = 1 legs of the other
BASELINE.json configurations at their own sizes (C3 wind per cell / aggregated, the C5 shard's heat demand and
runoff, C4 in full with the in-kernel solar position), each with kernel ms, algorithmic bytes, roofline fraction
and a parity check against the oracle.  ``--legs a,b`` restricts the run to some legs (what tools/profile_bench.sh
runs under rocprofv3; its summaries land in profiles/bench_profile_latest.json, from which ``frac_from_profile`` and
the counter ``traffic`` of every leg are quoted - an entry whose kernel name is not the leg's is refused).
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
    ap.add_argument("--collective", choices=["lib", "torch"], default="lib",
                    help="N > 1: who gathers - the library's own RCCL communicator (atl_comm_init + atl_allgather_time_v_async, "
                         "C ABI) or torch.distributed")
    ap.add_argument("--graph", action="store_true",
                    help="replay a rank's launches of one step (slot stride, fused kernel, k_combine) as ONE hipGraph "
                         "(atl_capture_begin / atl_graph_launch)")
    ap.add_argument("--workloads", default=None,
                    help="N > 1 (or --debug-gloo-one-gpu): comma-separated subset of the extra multi-GPU workloads c4,c5 run after the "
                         "C2 line (default: both when N > 1; 'none' to skip) - BASELINE configs[3] (pv 8760x800x800, 500 shapes, in-kernel "
                         "solar position, time-sharded) and configs[4] (heat demand + runoff 35040x400x400, 50 shapes, day-aligned shards)")
    ap.add_argument("--workload-steps", type=int, default=5)
    ap.add_argument("--legs", default="all",
                    help="comma-separated subset of the N = 1 legs: headline,night_skip,star_polygons,api,separate_cubes,"
                         "from_file,cpu,c3_series,c3_cf_map,c3_aggregated,c5_heat,c5_runoff,odd_caller,c2_sp,c4_full_sp (default: all)")
    ap.add_argument("--debug-rccl-self", action="store_true",
                    help="testing only (one GPU, one process): open a 1-rank RCCL process group and route the step "
                         "through the collective branch (async all-gather on the group's stream, placement copy)")
    ap.add_argument("--preflight-fail", default="", choices=["", "lib", "torch", "all"],
                    help="testing only: pretend this rung of the collective ladder (lib -> torch.distributed -> none) failed "
                         "its preflight")
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


PEAK_GBPS = 8000.0  # HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md

# the dominant kernel of every leg, as rocprofv3 names it (tools/profile_bench.py strips the anonymous namespace): a
# profile entry is only quoted for a leg when it was measured on THIS kernel
KERNELS = {
    "headline": "k_fused_segred<PvConvT<false, false, false, 0, 0, 0>, true, false>",
    "night_skip": "k_fused_segred_night<PvConvT<false, false, true, 0, 0, 0>, true, false, true>",
    "star_polygons": "k_fused_segred<PvConvT<false, false, false, 0, 0, 0>, true, false>",
    "star_night_skip": "k_fused_segred_night<PvConvT<false, false, true, 0, 0, 0>, true, false>",
    "c3_series": "k_cells_series_flat<WindConvT<1, -1>>",
    "c3_cf_map": "k_cells_timered<WindConvT<1, -1>, true>",
    "c3_aggregated": "k_fused_segred<WindConvT<1, -1>, true, false>",
    "c5_heat": "k_fused_segred<HeatConv, true, false>",
    "c5_runoff": "k_fused_segred<RunoffConv, true, false>",
    "odd_caller": "k_fused_segred<PvConvT<false, false, false, 0, 0, 0>, true, false>",
    "c4_full_sp": "k_fused_segred<PvConvT<true, false, false, 0, 0, 0>, true, false>",
    "c2_sp": "k_fused_segred<PvConvT<true, false, false, 0, 0, 0>, true, false>",
    "c4_headline": "k_fused_segred<PvConvT<true, false, false, 0, 0, 0>, true, false>",
}


def box_id():
    """Which machine this is: the host name alone is the same on every box of the pool (a container's), so the kernel's boot id
    goes with it - two runs share it exactly when they ran on the same boot of the same box."""
    import socket

    try:
        boot = open("/proc/sys/kernel/random/boot_id").read().strip()[:8]
    except OSError:
        boot = "?"
    return f"{socket.gethostname()}-{boot}"


def profile_entry(leg):
    """The committed rocprofv3 record of a leg (profiles/bench_profile_latest.json, written by tools/profile_bench.sh
    from --kernel-trace --stats and separate --pmc FETCH_SIZE / WRITE_SIZE passes over THIS file's ``--legs <leg>``
    run): {kernel, avg_us, median_us, calls, hbm_bytes_per_launch, source}.  None unless the record's kernel is the
    leg's dominant kernel by name."""
    f = ROOT / "profiles" / "bench_profile_latest.json"
    try:
        ent = json.loads(f.read_text()).get("legs", {}).get(leg)
    except Exception:
        return None
    if not ent or KERNELS.get(leg) is None or ent.get("kernel") != KERNELS[leg]:
        return None
    return ent


def lines_touched(covered, slot_cells, line_cells=16):
    """How many 128-byte lines of one time slot of one cube hold at least one cell some shape covers - what the memory system
    fetches for a plan whose shapes leave cells out (the kernel's lanes load only weighted cells, the hardware whole lines).
    `covered`: bool per cell; `slot_cells`: cells between the slots of a cube.  None when the slots do not start on a line
    (then the count differs from slot to slot)."""
    covered = np.asarray(covered, dtype=bool).ravel()
    if slot_cells % line_cells:
        return None
    pad = (-covered.size) % line_cells
    if pad:
        covered = np.concatenate([covered, np.zeros(pad, bool)])
    return int(covered.reshape(-1, line_cells).any(axis=1).sum())


def roofline_of(leg, algo_bytes, k_ms, extra=None):
    """The roofline object of a leg: achieved = algorithmic bytes / HIP-event kernel time of this run; from the committed
    profile of the same leg: the counter traffic per launch and the fraction its average duration gives.  Nothing above
    the peak is reported as evidence: such a figure is dropped with a note."""
    k_ms = np.atleast_1d(np.asarray(k_ms, dtype=np.float64))
    mean = float(k_ms.mean())
    achieved = algo_bytes / (mean * 1e-3) / 1e9
    r = {"bound": "hbm", "kernel": KERNELS.get(leg), "achieved": achieved, "peak": PEAK_GBPS, "unit": "GB/s",
         "frac": achieved / PEAK_GBPS, "traffic": None, "traffic_source": None, "kernel_ms": mean,
         "kernel_ms_median": float(np.median(k_ms)), "kernel_ms_min": float(k_ms.min()), "algorithmic_bytes": int(algo_bytes)}
    ent = profile_entry(leg)
    if ent:
        src = ent.get("source", "profiles/bench_profile_latest.json")
        if ent.get("avg_us"):
            us = ent.get("avg_us_timed") or ent["avg_us"]  # the launches after the leg's warm-up, when the record says which
            r["profile_kernel_ms"] = us * 1e-3
            r["frac_from_profile"] = algo_bytes / (us * 1e-6) / 1e9 / PEAK_GBPS
            import socket

            # the same build runs +-4 % from box to box: say whether the record and this line come from the same one
            r["profile_box"], r["this_box"] = ent.get("box"), box_id()
            r["profile_is_this_box"] = ent.get("box") == box_id()
            r["frac_vs_profile"] = r["frac"] / r["frac_from_profile"]
            r["profile_source"] = (f"{src} (rocprofv3 --kernel-trace --stats, avg of the {ent.get('timed_launches', ent.get('calls'))} "
                                   f"launches of this leg after its warm-up, not this run; all {ent.get('calls')} launches: {ent['avg_us'] * 1e-3:.3f} ms)")
        tr = ent.get("hbm_bytes_per_launch")
        if tr:
            on_traffic = tr / (mean * 1e-3) / 1e9
            if on_traffic <= PEAK_GBPS and tr >= 0.9 * algo_bytes:
                r["traffic"] = tr
                r["traffic_source"] = f"{src} (rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE, separate passes, not this run)"
                r["traffic_over_algorithmic"] = tr / algo_bytes
                r["achieved_on_traffic"] = on_traffic
            else:
                r["traffic_rejected"] = f"profile says {tr:.4g} B per launch = {on_traffic:.0f} GB/s at this run's kernel time: not credible, dropped"
    if r["achieved"] > PEAK_GBPS:
        r["invalid"] = "achieved exceeds the HBM peak: the kernel cannot have read its algorithmic bytes"
    if extra:
        r.update(extra)
    return r


def device_view(a, torch):
    """A torch view (no copy) of a DeviceArray - also of a pitched / slot-interleaved one - through __cuda_array_interface__."""

    class _V:
        pass

    v = _V()
    es = a.dtype.itemsize
    inner = int(np.prod(a.shape[1:], dtype=np.int64)) if a.ndim > 1 else 1
    strides = None
    if a.ld is not None:
        assert a.ndim == 2
        strides = (a.ld * es, es)
    v.__cuda_array_interface__ = {"shape": tuple(a.shape), "typestr": a.dtype.str, "data": (int(a.ptr), False), "version": 2,
                                  "strides": strides}
    return torch.as_tensor(v, device=f"cuda:{a.ctx.device}")


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
    for a in range(0, T_loc, g):
        n = min(g, T_loc - a)
        t = synthetic.time_index(n, "2013-01-01", off + a)
        h, dec = solar.hour_angle(t, x, "-30min")
        doy, hour = np.asarray(t.dayofyear, float), np.asarray(t.hour, float)
        tseason = 283.15 + 12.0 * np.sin(2 * np.pi * (doy - 110.0) / 365.0) + 5.0 * np.sin(2 * np.pi * (hour - 9.0) / 24.0)
        tabs = [ctx.upload(v) for v in (np.sin(dec), np.cos(dec), h, np.radians(y), tseason)]
        s = _lib.SynthSolar(*[v.ptr for v in tabs], X, Y, 42 + 1000003 * (off + a), 0 if ld == S else ld)
        ptrs = [big[k].ptr + a * ld * 8 for k in five] + [alt.ptr, az.ptr]
        _lib.check(ctx.lib.atl_synth_pv_inputs(ctx.handle, C.byref(s), n, S, *ptrs))
        ctx.sync()
    del alt, az
    t = synthetic.time_index(T_loc, "2013-01-01", off)
    h, dec = solar.hour_angle(t, x, "-30min")
    lat = np.radians(y)
    tables = dict(sin_dec=np.sin(dec), cos_dec=np.cos(dec), h=h, cos_h=np.cos(h), sin_lat=np.sin(lat), cos_lat=np.cos(lat))
    tables = {k: ctx.upload(np.ascontiguousarray(v)) for k, v in tables.items()}
    return big, x, y, tables


def config_legs(ctx, legs, reps, check=True):
    """N = 1 legs of the other BASELINE.json configurations at their own sizes, through the same C ABI calls the Python API
    makes: kernel ms (HIP events of the dominant kernel), algorithmic bytes, roofline fraction, parity of a
    sample against the oracle (rtol 1e-10, atol 1e-12 max; heat demand atol 1e-9 a).
      c3_*   configs[2]: Cutout.wind('Vestas_V112_3MW') on 8760 x 400 x 400 - per-cell series (24 B/cell-step), capacity-
             factor map (16 B), aggregated to 100 shapes (16 B)
      c5_*   configs[4]: one GPU's 1/8 shard (4380 steps) of the 35040 x 400 x 400 heat demand / runoff run, 50 shapes (8 B each)
      c4_full_sp  configs[3] on ONE GPU: pv 8760 x 800 x 800, 500 shapes, in-kernel solar position (5 cubes, 40 B, 224 GB)"""
    import gc

    import torch

    from atlite_amd import Cutout, Dataset, _lib, gis, solar, synthetic
    from atlite_amd.device import SlotPool, pitch_for
    from atlite_amd.resource import get_windturbineconfig
    from oracle import atlite_oracle as orc

    out = {}
    WARM = 10

    def run(fn):
        ctx.set_profiling(True)
        ms, r = [], None
        for i in range(WARM + reps):
            r = None  # free the previous result first (a per-cell series is 11 GB)
            r = fn()
            if i >= WARM:
                ms.append(ctx.last_kernel_ms())
        return np.asarray(ms), r

    def rows(dev, sel):
        return np.stack([dev.slab(int(t), int(t) + 1).numpy()[0] for t in sel])

    def matrix_of(Y, X, n):
        x, y = synthetic.grid_coords(Y, X)
        dx, dy = x[1] - x[0], y[1] - y[0]
        polys = gis.random_tessellation(n, (x[0] - dx / 2, y[0] - dy / 2, x[-1] + dx / 2, y[-1] + dy / 2), seed=42)
        return gis.compute_indicatormatrix(x, y, polys, ctx=ctx)

    def close(got, ref, atol=None):
        scale = float(np.nanmax(np.abs(ref))) if np.size(ref) else 0.0
        atol = 1e-12 * scale if atol is None else atol
        err = np.abs(got - ref) - atol
        rel = float(np.nanmax(err / np.maximum(np.abs(ref), 1e-300))) if np.size(ref) else 0.0
        ok = bool(np.allclose(got, ref, rtol=1e-10, atol=atol, equal_nan=True))
        return {"ok": ok, "checked_values": int(np.size(ref)), "max_rel_err_beyond_atol": max(rel, 0.0), "rtol": 1e-10}

    def api_times(call, what):
        t0 = time.perf_counter()
        call()
        cold = (time.perf_counter() - t0) * 1e3
        warm = []
        for _ in range(5):
            t0 = time.perf_counter()
            call()
            warm.append((time.perf_counter() - t0) * 1e3)
        return {"call": what, "cold": cold, "warm": min(warm)}

    def record(leg, workload, bytes_, cells, ms, parity):
        r = roofline_of(leg, bytes_, ms)
        e = {"workload": workload, "kernel": r["kernel"], "ms": r["kernel_ms"], "ms_median": r["kernel_ms_median"], "ms_min": r["kernel_ms_min"],
             "algorithmic_bytes": r["algorithmic_bytes"], "achieved_GBps": r["achieved"], "frac": r["frac"],
             "value": cells / (r["kernel_ms"] * 1e-3), "unit": "cell-timesteps/s (kernel time)"}
        for k in ("frac_from_profile", "profile_kernel_ms", "traffic", "traffic_over_algorithmic", "traffic_rejected", "invalid"):
            if k in r:
                e[k] = r[k]
        if parity is not None:
            e["parity"] = parity
        out[leg] = e

    # ---- configs[2]: wind per cell -------------------------------------------------------------------------------------
    c3 = [l for l in legs if l.startswith("c3_")]
    if c3:
        T, Y, X = 8760, 400, 400
        S = Y * X
        turb = get_windturbineconfig("Vestas_V112_3MW")
        V, POW, P, hub = turb["V"], turb["POW"], turb["P"], turb["hub_height"]
        inter = os.environ.get("ATL_BENCH_WIND_LAYOUT", "interleaved") == "interleaved"
        if inter:  # the layout of the library's own device copies: the two cubes slot-interleaved in one allocation
            pool = SlotPool(ctx, T, S, ["wnd100m", "roughness"], pitch_for(S))
            wnd, z0 = pool.view("wnd100m"), pool.view("roughness")
            tmp = ctx.empty((T, S))
            for var, kind, p0, p1, dst in ((5, _lib.SYN_RAYLEIGH, 8.0, 0.0, wnd), (6, _lib.SYN_EXPLOG, 1e-3, 1.5e3, z0)):
                _lib.check(ctx.lib.atl_synth_field(ctx.handle, kind, 42, var, p0, p1, 0, T, S, tmp.ptr))
                _lib.check(ctx.lib.atl_copy_2d(ctx.handle, dst.ptr, dst.ld * 8, tmp.ptr, S * 8, S * 8, T, 2, 0))
            ctx.sync()
            del tmp
        else:
            d = synthetic.wind_inputs(ctx, T, Y, X)
            wnd, z0 = d["wnd100m"], d["roughness"]
        args = (wnd, z0, V, POW / P, hub, 100.0, "logarithmic", T, S)
        lay = "the two cubes slot-interleaved in one allocation" if inter else "one allocation per cube"
        if "c3_series" in c3:
            # the 11 GB result is allocated ONCE and written by every launch (as the headline's result is): a fresh allocation per
            # call made this leg bimodal from process to process - 5.3 / 5.6 / 6.0-6.2 ms with one library on one box, by where the
            # driver happened to place the new block (tools/probes/series_placement.py: 5.25-5.30 ms on a buffer that stays)
            ser = ctx.empty((T, S))
            ms, _none = run(lambda: ctx.wind(*args, out=(ser.ptr, S)))
            par = None
            if check:
                sel = np.unique(np.concatenate([np.arange(0, 6), [T // 2, T - 1]]))
                par = close(rows(ser, sel), orc.convert_wind(rows(wnd, sel), rows(z0, sel), V, POW, P, hub, 100.0))
            record("c3_series", f"configs[2]: wind V112 per-cell series, {T}x{Y}x{X}, 16 B read + 8 B written per cell-step; {lay}",
                   24 * T * S, T * S, ms, par)
            del ser
        if "c3_cf_map" in c3:
            ms, cf = run(lambda: ctx.wind(*args, time_agg="mean"))
            par = None
            if check:
                cells = np.unique(np.concatenate([[0, X - 1, S - 1], np.random.default_rng(1).integers(0, S, 61)]))
                idx = torch.as_tensor(cells, device=f"cuda:{ctx.device}")
                w_c, z_c = device_view(wnd, torch)[:, idx].cpu().numpy(), device_view(z0, torch)[:, idx].cpu().numpy()
                par = close(cf.numpy()[cells], orc.convert_wind(w_c, z_c, V, POW, P, hub, 100.0).mean(axis=0))
            record("c3_cf_map", f"configs[2]: wind V112 capacity-factor map (aggregate_time='mean'), {T}x{Y}x{X}, 16 B per cell-step; {lay}",
                   16 * T * S, T * S, ms, par)
            del cf
        if "c3_aggregated" in c3:
            M = matrix_of(Y, X, 100)
            plan = ctx.plan(M, row_len=X, ld=wnd.ld)
            ms, agg = run(lambda: ctx.wind(*args, plan=plan))
            par = None
            if check:
                sel = np.unique(np.concatenate([np.arange(0, 20), [T // 2, T - 1]]))
                par = close(agg.numpy()[:, sel], orc.aggregate_matrix(orc.convert_wind(rows(wnd, sel), rows(z0, sel), V, POW, P, hub, 100.0), M))
            info = plan.info()
            record("c3_aggregated", f"configs[2] aggregated: wind V112, {T}x{Y}x{X}, 100 tessellation shapes ({info['tile_w']}x{info['tile_h']} "
                                    f"tiles, {info['n_partial_rows']} partial rows), 16 B per cell-step; {lay}", 16 * T * S, T * S, ms, par)
            # what a user of the drop-in API waits for: cutout.wind(...) on the device-resident dataset -> host result
            xg, yg = synthetic.grid_coords(Y, X)
            cut = Cutout(Dataset({"wnd100m": wnd, "roughness": z0}, dict(time=synthetic.time_index(T), y=yg, x=xg)))
            out["c3_aggregated"]["api_e2e_ms"] = api_times(lambda: cut.wind(turbine="Vestas_V112_3MW", matrix=M, aggregate_time=None),
                                                           "cutout.wind(turbine='Vestas_V112_3MW', matrix=M, aggregate_time=None) -> host (shapes x time)")
            del agg, plan, cut
        del wnd, z0, args
        if inter:
            del pool
        gc.collect()

    # ---- configs[4]: one GPU's shard of heat demand + runoff ------------------------------------------------------------
    c5 = [l for l in legs if l.startswith("c5_")]
    if c5:
        T, Y, X = 35040 // 8, 400, 400
        S = Y * X
        d = synthetic.heat_runoff_inputs(ctx, T, Y, X)
        M = matrix_of(Y, X, 50)
        plan = ctx.plan(M, row_len=X)
        info = plan.info()
        tag = f"{T} of 35040 steps x {Y}x{X}, 50 tessellation shapes ({info['tile_w']}x{info['tile_h']} tiles, {info['n_partial_rows']} partial rows), 8 B per cell-step"
        if "c5_heat" in c5:
            day_ptr = np.arange(0, T + 1, 24)
            if day_ptr[-1] != T:
                day_ptr = np.append(day_ptr, T)
            ms, hd = run(lambda: ctx.heat_demand(d["temperature"], day_ptr, 288.15, 1.0, 0.0, T, S, plan=plan))
            par = None
            if check:
                days = np.unique(np.concatenate([[0, 1, len(day_ptr) - 2], np.random.default_rng(2).integers(0, len(day_ptr) - 1, 9)]))
                got, ref = hd.numpy()[:, days], []
                for dd in days:
                    blk = d["temperature"].slab(int(day_ptr[dd]), int(day_ptr[dd + 1])).numpy()
                    ref.append(orc.aggregate_matrix(orc.convert_heat_demand(blk, np.array([0, blk.shape[0]]), threshold=15.0, a=1.0, constant=0.0), M)[:, 0])
                par = close(got, np.stack(ref, axis=1), atol=1e-9)
            record("c5_heat", "configs[4], one GPU's shard: heat demand, " + tag, 8 * T * S, T * S, ms, par)
            del hd
        if "c5_runoff" in c5:
            ms, ro = run(lambda: ctx.runoff(d["runoff"], d["height"], T, S, plan=plan))
            par = None
            if check:
                sel = np.unique(np.concatenate([np.arange(0, 20), [T // 2, T - 1]]))
                par = close(ro.numpy()[:, sel], orc.aggregate_matrix(orc.convert_runoff(rows(d["runoff"], sel), d["height"].numpy()[None, :]), M))
            record("c5_runoff", "configs[4], one GPU's shard: runoff x height, " + tag, 8 * T * S, T * S, ms, par)
            xg, yg = synthetic.grid_coords(Y, X)
            cut = Cutout(Dataset({"runoff": d["runoff"], "height": d["height"]}, dict(time=synthetic.time_index(T, "2011-01-01"), y=yg, x=xg)))
            out["c5_runoff"]["api_e2e_ms"] = api_times(lambda: cut.runoff(matrix=M, aggregate_time=None, smooth=True),
                                                       "cutout.runoff(matrix=M, aggregate_time=None, smooth=True): aggregation + 168-step rolling mean "
                                                       "on the device -> host (shapes x time)")
            del ro, cut
        del d, plan
        gc.collect()

    # ---- a real-world (odd) grid held by the CALLER as contiguous cubes: ordinary plan vs line-aligned plan ----------------
    if "odd_caller" in legs:
        T, Y, X = 8760, 201, 201  # 50 x 50 degrees at 0.25: S % 16 = 1, every slot starts somewhere else in its 128-byte line
        S = Y * X
        cubes, _ = synthetic.pv_inputs(ctx, T, Y, X)  # one contiguous (T, S) allocation per variable: the caller's own layout
        M = matrix_of(Y, X, 100)
        params = dict(CSI, **ORI)
        plain, aligned = ctx.plan(M, row_len=X), ctx.plan(M, row_len=X, aligned=True)
        ms0, _r = run(lambda: ctx.pv(cubes, params, T, S, plan=plain, options=dict(night_skip=False)))
        del _r
        ms, res = run(lambda: ctx.pv(cubes, params, T, S, plan=aligned, options=dict(night_skip=False)))
        par = None
        if check:
            sel = np.unique(np.concatenate([np.arange(0, 24), [T // 2, T // 2 + 1, T - 1]]))
            host = {k: rows(v, sel) for k, v in cubes.items()}
            par = close(res.numpy()[:, sel], orc.aggregate_matrix(orc.convert_pv(host, CSI, ORI), M))
        info = aligned.info()
        record("odd_caller", f"pv {T}x{Y}x{X} (S % 16 = 1), 100 tessellation shapes, the seven cubes CONTIGUOUS as a caller holds them (no padded "
                             f"copy): line-aligned plan (atl_agg_create_aligned: 16 tilings of {info['tile_w']}x{info['tile_h']} tiles, "
                             f"{info['n_partial_rows']} partial rows), 56 B per cell-step", 56 * T * S, T * S, ms, par)
        out["odd_caller"]["ordinary_plan_ms"] = float(np.mean(ms0))
        out["odd_caller"]["ordinary_plan_frac"] = 56 * T * S / (float(np.mean(ms0)) * 1e-3) / 1e9 / PEAK_GBPS
        del res, cubes, plain, aligned
        gc.collect()

    # ---- configs[3] on one GPU: the whole 8760 x 800 x 800 cutout, in-kernel solar position ------------------------------
    # (c2_sp: the same kernel on the C2 grid - 8760 x 200 x 200, 100 shapes - where it is bound by the vector ALU, not by HBM)
    for leg_sp, (T, Y, X, N) in (("c2_sp", (8760, 200, 200, 100)),):
        if leg_sp not in legs:
            continue
        S = Y * X
        big, x, y, tables = generate_pv(ctx, synthetic, solar, _lib, T, Y, X, 0, False, interleaved=True)
        ld = next(iter(big.values())).ld
        M = matrix_of(Y, X, N)
        plan = ctx.plan(M, row_len=X, ld=ld)
        info = plan.info()
        tabs = dict(tables)
        params = dict(CSI, **ORI)
        ms, res = run(lambda: ctx.pv(big, params, T, S, plan=plan, solar_tables=tabs, options=dict(night_skip=False)))
        par = None
        if check:
            sel = np.unique(np.clip(np.concatenate([np.arange(0, 48), np.arange(4000, 4048), [T - 1]]), 0, T - 1))
            host = {k: rows(v, sel) for k, v in big.items()}
            al, az = orc.solar_position(synthetic.time_index(T)[sel], x, y, "-30min")
            host["solar_altitude"], host["solar_azimuth"] = al.reshape(len(sel), S), az.reshape(len(sel), S)
            par = close(res.numpy()[:, sel], orc.aggregate_matrix(orc.convert_pv(host, CSI, ORI), M))
        record(leg_sp, f"pv CSi, {T}x{Y}x{X}, {N} tessellation shapes ({info['tile_w']}x{info['tile_h']} tiles, {info['n_partial_rows']} partial rows), "
                       f"IN-KERNEL solar position, 5 cubes slot-interleaved = 40 B per cell-step", 40 * T * S, T * S, ms, par)
        del big, res, plan, tabs, tables
        gc.collect()
    if "c4_full_sp" in legs:
        T, Y, X, N = 8760, 800, 800, 500
        S = Y * X
        big, x, y, tables = generate_pv(ctx, synthetic, solar, _lib, T, Y, X, 0, False, interleaved=True)
        ld = next(iter(big.values())).ld
        M = matrix_of(Y, X, N)
        plan = ctx.plan(M, row_len=X, ld=ld)
        info = plan.info()
        tabs = dict(tables)
        params = dict(CSI, **ORI)
        ms, res = run(lambda: ctx.pv(big, params, T, S, plan=plan, solar_tables=tabs, options=dict(night_skip=False)))
        par = None
        if check:
            sel = np.unique(np.clip(np.concatenate([np.arange(0, 24), np.arange(4000, 4024), [T - 1]]), 0, T - 1))
            host = {k: rows(v, sel) for k, v in big.items()}
            al, az = orc.solar_position(synthetic.time_index(T)[sel], x, y, "-30min")
            host["solar_altitude"], host["solar_azimuth"] = al.reshape(len(sel), S), az.reshape(len(sel), S)
            ref = np.concatenate([orc.aggregate_matrix(orc.convert_pv({k: v[i:i + 8] for k, v in host.items()}, CSI, ORI), M)
                                  for i in range(0, len(sel), 8)], axis=1)
            par = close(res.numpy()[:, sel], ref)
        record("c4_full_sp", f"configs[3] on ONE GPU: pv CSi, {T}x{Y}x{X}, {N} tessellation shapes ({info['tile_w']}x{info['tile_h']} tiles, "
                             f"{info['n_partial_rows']} partial rows), in-kernel solar position, 5 cubes slot-interleaved = 40 B per cell-step (224 GB resident)",
               40 * T * S, T * S, ms, par)
        del big, res, plan, tabs, tables
        gc.collect()
    return out


def sharded_workload(name, ctx, a, rank, world, dev, torch, dist, comm, mode, barrier, fence, D, _lib, gis, synthetic, solar, parts=None):
    """One of BASELINE.json's 8-GPU configurations as a `world`-rank run: every rank converts + aggregates its own time shard,
    ONE ragged all-gather per result reassembles (shapes x time) on every rank.
      c4  configs[3]: pv CSi 8760 x 800 x 800, 500 shapes, in-kernel solar position (5 cubes, 40 B per cell-step)
      c5  configs[4]: heat demand (daily means; shards on calendar days) + runoff, 35040 x 400 x 400, 50 shapes (8 B + 8 B)
    mode: "lib" (atl_allgather_time_v[_async] on the library's communicator), "torch" (torch.distributed, padded blocks),
    "gloo-debug" (host copies, all ranks on one GPU) or "none" (no collective on this node: the ranks' own steps).
    Returns ms per step with the gather behind the next step's kernel and with the gather finished inside the step, the kernel
    ms of every rank, the gather alone, and the parity of THIS rank's block against the oracle on a sample."""
    from oracle import atlite_oracle as orc

    # parts: shards of the time axis (= world; --emulate-shard N on one GPU: N, of which this process runs rank 0's, no collective)
    parts = world if parts is None else parts
    assert parts == world or (world == 1 and mode == "none")
    steps, warm = max(2, a.workload_steps), 2
    if name == "c4":
        T, Y, X, N = 8760, 800, 800, 500
        bpc, align = 40, 1
    elif name == "c5":
        T, Y, X, N = 35040, 400, 400, 50
        bpc, align = 16, 24
    else:
        raise ValueError(f"unknown workload {name!r}")
    if os.environ.get("ATL_BENCH_WORKLOAD_SCALE"):  # testing: a fraction of the time axis (whole days)
        T = max(48 * parts, int(T * float(os.environ["ATL_BENCH_WORKLOAD_SCALE"])) // 24 * 24)
    S = Y * X
    edges = D.time_partition(T, parts, align=align)
    lens = [edges[r + 1] - edges[r] for r in range(parts)]
    T_loc, off = lens[rank], edges[rank]
    x, y = synthetic.grid_coords(Y, X)
    dx, dy = x[1] - x[0], y[1] - y[0]
    polys = gis.random_tessellation(N, (x[0] - dx / 2, y[0] - dy / 2, x[-1] + dx / 2, y[-1] + dy / 2), seed=42)
    M = gis.compute_indicatormatrix(x, y, polys, ctx=ctx)

    def rows(dev_arr, sel):
        return np.stack([dev_arr.slab(int(t), int(t) + 1).numpy()[0] for t in sel])

    def close(got, ref, atol=None):
        scale = float(np.nanmax(np.abs(ref))) if np.size(ref) else 0.0
        atol = 1e-12 * scale if atol is None else atol
        err = np.abs(got - ref) - atol
        rel = float(np.nanmax(err / np.maximum(np.abs(ref), 1e-300))) if np.size(ref) else 0.0
        return {"ok": bool(np.allclose(got, ref, rtol=1e-10, atol=atol, equal_nan=True)), "checked_values": int(np.size(ref)),
                "max_rel_err_beyond_atol": max(rel, 0.0), "rtol": 1e-10}

    # ---- the rank's shard: cubes, plan, and the launches of one step as (result name, slots, launch(out_ptr, ld_out)) ----------
    if name == "c4":
        big, _x, _y, tables = generate_pv(ctx, synthetic, solar, _lib, T_loc, Y, X, off, False, interleaved=True)
        ld = next(iter(big.values())).ld
        plan = ctx.plan(M, row_len=X, ld=ld)
        params, tabs = dict(CSI, **ORI), dict(tables)
        outs = [("pv", lens, lambda ptr, ldo: ctx.pv(big, params, T_loc, S, plan=plan, solar_tables=tabs, options=dict(night_skip=False),
                                                     out=(ptr, ldo)))]

        def parity(block):
            sel = np.unique(np.clip(np.concatenate([np.arange(0, 8), np.arange(T_loc // 2, T_loc // 2 + 8)]), 0, T_loc - 1))
            host = {k: rows(v, sel) for k, v in big.items()}
            al, az = orc.solar_position(synthetic.time_index(T_loc, "2013-01-01", off)[sel], x, y, "-30min")
            host["solar_altitude"], host["solar_azimuth"] = al.reshape(len(sel), S), az.reshape(len(sel), S)
            ref = np.concatenate([orc.aggregate_matrix(orc.convert_pv({k: v[i:i + 8] for k, v in host.items()}, CSI, ORI), M)
                                  for i in range(0, len(sel), 8)], axis=1)
            return close(block["pv"][:, sel], ref)
    else:
        d = synthetic.heat_runoff_inputs(ctx, T_loc, Y, X)
        plan = ctx.plan(M, row_len=X)
        day_ptr = np.arange(0, T_loc + 1, 24)
        if day_ptr[-1] != T_loc:
            day_ptr = np.append(day_ptr, T_loc)
        day_lens = [-(-l // 24) for l in lens]  # every shard starts on a day boundary (time_partition(align=24))
        outs = [("heat_demand", day_lens, lambda ptr, ldo: ctx.heat_demand(d["temperature"], day_ptr, 288.15, 1.0, 0.0, T_loc, S, plan=plan,
                                                                          out=(ptr, ldo))),
                ("runoff", lens, lambda ptr, ldo: ctx.runoff(d["runoff"], d["height"], T_loc, S, plan=plan, out=(ptr, ldo)))]

        def parity(block):
            days = np.unique(np.concatenate([[0, 1, len(day_ptr) - 2], np.random.default_rng(2).integers(0, len(day_ptr) - 1, 5)]))
            ref = []
            for dd in days:
                blk = d["temperature"].slab(int(day_ptr[dd]), int(day_ptr[dd + 1])).numpy()
                ref.append(orc.aggregate_matrix(orc.convert_heat_demand(blk, np.array([0, blk.shape[0]]), threshold=15.0, a=1.0, constant=0.0), M)[:, 0])
            p1 = close(block["heat_demand"][:, days], np.stack(ref, axis=1), atol=1e-9)
            sel = np.unique(np.concatenate([np.arange(0, 12), [T_loc // 2, T_loc - 1]]))
            p2 = close(block["runoff"][:, sel], orc.aggregate_matrix(orc.convert_runoff(rows(d["runoff"], sel), d["height"].numpy()[None, :]), M))
            return {"ok": p1["ok"] and p2["ok"], "heat_demand": p1, "runoff": p2}

    # ---- buffers: per result two pieces (step parity) and the gathered (N x sum of slots) --------------------------------------
    side = torch.cuda.Stream(device=dev) if mode == "torch" else None
    bufs = []
    for rname, slens, fn in outs:
        mine, tot, mx = slens[rank], sum(slens), max(slens)
        b = dict(name=rname, lens=slens, fn=fn, mine=mine, total=tot, start=sum(slens[:rank]),
                 piece=[torch.empty((N, max(mine, 1)), dtype=torch.float64, device=dev) for _ in range(2)],
                 full=torch.empty((N, tot), dtype=torch.float64, device=dev), ticket=[None, None])
        if mode in ("torch", "gloo-debug"):  # equal-sized blocks for all_gather_into_tensor: shards padded to the longest
            b["pad"] = [torch.zeros((N, mx), dtype=torch.float64, device=dev) for _ in range(2)]
            b["gb"] = [torch.empty((world, N, mx), dtype=torch.float64, device=dev) for _ in range(2)]
        bufs.append(b)

    def gather(b, par, overlap):
        if mode == "none":
            b["full"][:, b["start"]:b["start"] + b["mine"]].copy_(b["piece"][par][:, :b["mine"]])
            return
        if mode == "lib":
            if overlap:
                b["ticket"][par] = comm.gather_time_v_async(b["piece"][par].data_ptr(), N, b["lens"], b["full"].data_ptr(), b["total"])
            else:
                h_lens = (C.c_int64 * world)(*b["lens"])
                _lib.check(ctx.lib.atl_allgather_time_v(comm.handle, b["piece"][par].data_ptr(), N, h_lens, b["full"].data_ptr(), b["total"]))
            return
        pad, gb = b["pad"][par], b["gb"][par]
        pad[:, :b["mine"]].copy_(b["piece"][par][:, :b["mine"]])

        def place():
            o = 0
            for r in range(world):
                b["full"][:, o:o + b["lens"][r]].copy_(gb[r, :, :b["lens"][r]])
                o += b["lens"][r]

        if mode == "gloo-debug":
            torch.cuda.current_stream().synchronize()
            host = torch.empty(gb.shape, dtype=torch.float64)
            dist.all_gather_into_tensor(host.view(-1), pad.cpu().view(-1))
            gb.copy_(host)
            place()
            return
        w = dist.all_gather_into_tensor(gb.view(-1), pad.view(-1), async_op=True)
        if overlap:
            with torch.cuda.stream(side):
                w.wait()
                place()
                ev = torch.cuda.Event()
                ev.record(side)
                b["ticket"][par] = ev
        else:
            w.wait()
            place()

    step_no = [0]

    def step(overlap):
        par = step_no[0] & 1
        step_no[0] += 1
        for b in bufs:
            t = b["ticket"][par]
            if t is not None:  # the gather that last read this piece (two steps back)
                if mode == "lib":
                    comm.wait(t)
                else:
                    torch.cuda.current_stream().wait_event(t)
                b["ticket"][par] = None
            b["fn"](b["piece"][par].data_ptr(), b["piece"][par].stride(0))
            gather(b, par, overlap)

    def timed(overlap):
        n_k = len(bufs)
        ctx.set_profiling(max(2, (steps + warm) * n_k))
        for _ in range(warm):
            step(overlap)
        fence()
        ctx.set_profiling(max(2, steps * n_k))
        t0 = time.perf_counter()
        for _ in range(steps):
            step(overlap)
        fence()
        dt = time.perf_counter() - t0
        k = np.asarray(ctx.kernel_times(), float)
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        k_step = float(k.reshape(-1, n_k).sum(axis=1).mean()) if k.size and k.size % n_k == 0 else (float(k.sum()) / steps if k.size else None)
        return dt / steps * 1e3, k_step

    can_overlap = mode in ("lib", "torch")
    ms_ov, k_ov = timed(True) if can_overlap else (None, None)
    ms_sync, k_sync = timed(False)
    ms_step = ms_ov if ms_ov is not None else ms_sync
    k_ms = k_ov if k_ov is not None else k_sync
    # the gather alone (every result of one step), nothing to hide behind
    gather_ms = None
    if mode != "none":
        fence()
        t0 = time.perf_counter()
        for _ in range(5):
            for b in bufs:
                gather(b, 0, False)
        fence()
        tg = torch.tensor([(time.perf_counter() - t0) / 5 * 1e3], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        gather_ms = float(tg.item())
    per_rank = [k_ms]
    if world > 1:
        mine_t = torch.tensor([k_ms or 0.0], dtype=torch.float64)
        lst = [torch.zeros_like(mine_t) for _ in range(world)]
        dist.all_gather(lst, mine_t)
        per_rank = [float(v.item()) for v in lst]
    # ---- checks: this rank's block in place and equal to the oracle on a sample; all ranks hold the same gathered result --------
    step(False)
    fence()
    block, placed_ok, agree = {}, True, True
    for b in bufs:
        own = b["piece"][(step_no[0] - 1) & 1][:, :b["mine"]]
        placed_ok = placed_ok and bool(torch.equal(b["full"][:, b["start"]:b["start"] + b["mine"]], own))
        block[b["name"]] = own.cpu().numpy()
        if world > 1 and mode != "none":
            chk = torch.stack([b["full"].sum(), b["full"].abs().sum()]).cpu()
            lst = [torch.zeros_like(chk) for _ in range(world)]
            dist.all_gather(lst, chk)
            agree = agree and all(torch.equal(v, chk) for v in lst)
    par = parity(block) if not a.no_parity else None
    if world > 1 and par is not None:  # every rank checks its own shard; the line carries rank 0's record and the AND over ranks
        okf = torch.tensor([1 if par["ok"] else 0], dtype=torch.int32)
        dist.all_reduce(okf, op=dist.ReduceOp.MIN)
        par["ok_on_every_rank"] = bool(int(okf.item()) == 1)
    cells = (T if parts == world else T_loc) * S  # (an emulated shard: this rank's own cell-steps)
    info = plan.info()
    return {
        "workload": (f"configs[3]: pv CSi slope 30 az 180, {T}x{Y}x{X} fp64, {N} tessellation shapes, in-kernel solar position (5 cubes, 40 B per "
                     f"cell-step)" if name == "c4" else
                     f"configs[4]: heat demand (threshold 15, daily means) + runoff x height, {T}x{Y}x{X} fp64, {N} tessellation shapes "
                     f"(8 B + 8 B per cell-step), shards on calendar days") + f"; time-sharded x{parts}, {info['tile_w']}x{info['tile_h']} tiles, "
                    f"{info['n_partial_rows']} partial rows",
        "value": cells / (ms_step * 1e-3), "unit": "cell-timesteps/s", "scaling": "strong", "n_gpus": world, "shards": parts, "steps": steps,
        "ms_per_step": ms_step, "ms_per_step_gather_overlapped": ms_ov, "ms_per_step_gather_inside_the_step": ms_sync,
        "time_steps_per_gpu": lens, "per_rank_kernel_ms": per_rank, "gather_ms": gather_ms,
        "result_bytes": int(sum(N * b["total"] * 8 for b in bufs)), "collective": mode,
        "ranks_seen": (comm.info()["n_ranks"] if comm is not None else None),
        "algorithmic_bytes_per_rank": int(bpc * T_loc * S),
        "frac_of_hbm_peak_rank_kernel": (bpc * T_loc * S / (k_ms * 1e-3) / 1e9 / PEAK_GBPS) if k_ms else None,
        "own_block_in_place": placed_ok, "ranks_agree_on_the_result": agree, "parity": par,
    }



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


def from_file_leg(ctx, _lib, gis, M, Y, X, T=8760, chunks=(24, 100, 100), host_calls=1, device_calls=4):
    """pv from a cutout file, device inflate vs host inflate: seconds, cell-steps/s, GB/s of file and fp64-equivalent GB/s.
    Default: a whole year of the C2 grid (10 220 chunk streams of 960 kB: more than the 8 192 the device holds at a time)."""
    import subprocess
    import tempfile

    import atlite_amd as aa

    conda = "/opt/conda/bin/python3.9"
    if not os.path.exists(conda):
        return {"skipped": "no conda interpreter with h5py to write the file"}
    tmp = tempfile.mkdtemp(prefix="atl_bench_", dir="/tmp")
    path = os.path.join(tmp, "cutout.nc")
    t0 = time.perf_counter()
    subprocess.run([conda, str(ROOT / "tests" / "golden" / "make_nc_fixtures.py"), "--cutout", path, str(T), str(Y), str(X),
                    *[str(c) for c in chunks], "f4", "11", str(max(2, len(os.sched_getaffinity(0)))), "pv"], check=True, timeout=400)
    t_write = time.perf_counter() - t0
    names = ["influx_direct", "influx_diffuse", "influx_toa", "albedo", "temperature", "solar_altitude", "solar_azimuth"]
    out = {"file": f"NetCDF-4, {T}x{Y}x{X} float32, chunks {tuple(chunks)}, shuffle + zlib level 1, the 7 pv cubes; written in {t_write:.1f} s",
           "call": "Cutout(path).pv(panel='CSi', orientation={slope:30,azimuth:180}, matrix=M, aggregate_time=None).values"}
    try:
        cf = aa.Cutout(path)
        if M is None:
            M = gis.compute_indicatormatrix(cf.coords["x"], cf.coords["y"], gis.random_tessellation(100, cf.bounds, seed=0))
        kw = dict(panel="CSi", orientation={"slope": 30.0, "azimuth": 180.0}, matrix=M, aggregate_time=None)
        disk = sum(cf.data.file.variables[v].stored_bytes for v in names)
        out["chunk_streams"] = int(sum(cf.data.file.variables[v].n_chunks for v in names))
        cells = T * Y * X
        dctx = default_context_of()

        def times():
            ms = (C.c_double * 5)()
            cb, rb = C.c_int64(), C.c_int64()
            _lib.check(dctx.lib.atl_nc_ingest_times(dctx.handle, ms, C.byref(cb), C.byref(rb)))
            st = [C.c_int64() for _ in range(3)]
            _lib.check(dctx.lib.atl_nc_ingest_stats(dctx.handle, *[C.byref(x) for x in st]))
            return np.array(list(ms)), cb.value, rb.value, [x.value for x in st]

        def leg(mode, n):
            os.environ["ATLITE_HIP_INFLATE"] = mode
            ts, r = [], None
            for _ in range(n):
                t0 = time.perf_counter()
                r = cf.pv(**kw).values
                ts.append(time.perf_counter() - t0)
            best = min(ts)
            return r, {"first_s": ts[0], "best_s": best, "calls": n, "value": cells / best, "unit": "cell-timesteps/s",
                       "file_GBps": disk / best / 1e9, "fp64_equivalent_GBps": 7 * cells * 8 / best / 1e9}

        try:
            st0 = times()[3]
            r_dev, out["device_inflate"] = leg("", device_calls)  # the library's own choice (device from 1024 streams on)
            st1 = times()[3]
            out["device_inflate"]["chunks_inflated_on_the_device"] = st1[0] - st0[0]
            out["device_inflate"]["chunks_inflated_on_host_threads"] = st1[1] - st0[1]
            if st1[0] - st0[0] > 0:
                m0, c0, r0, s0 = times()
                cf.pv(**kw)
                m1, c1, r1, s1 = times()
                dm = m1 - m0
                out["device_inflate"]["one_warm_call"] = {
                    "host_gather_ms": dm[0], "h2d_ms": dm[1], "k_inflate_ms": dm[2], "unpack_of_unwritten_chunks_ms": dm[4],
                    "segments_decoded_side_by_side": int(dm[3]),
                    "streams": s1[0] - s0[0], "redone_on_host": s1[2] - s0[2], "compressed_bytes": c1 - c0, "inflated_bytes": r1 - r0,
                    "note": "many streams - ONE fed launch: the kernel starts first, the compressed bytes follow in DMA batches (h2d_ms: "
                            "first to last DMA), each wave waits for its stream's batch, then inflates, checks the Adler-32 and unpacks "
                            "its chunk (k_inflate_ms spans all of that, waits included); host_gather_ms is the preads' wall time inside "
                            "the same window - the three overlap.  Few long streams (segments_decoded_side_by_side > 0): their DEFLATE "
                            "blocks are found (behind the DMAs), counted, decoded and resolved by a wave per block; k_inflate_ms spans "
                            "those passes and starts when the last DMA has landed"}
            r_host, out["host_inflate"] = leg("host", host_calls)
            out["bit_identical"] = bool(np.array_equal(np.asarray(r_dev), np.asarray(r_host)))
            out["stored_bytes"] = int(disk)
        finally:
            os.environ.pop("ATLITE_HIP_INFLATE", None)
    finally:
        try:
            os.remove(path)
            os.rmdir(tmp)
        except OSError:
            pass
    return out


def default_context_of():
    from atlite_amd.device import default_context

    return default_context()


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
    if a.debug_rccl_self and world == 1 and a.collective == "torch":
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
            # control plane (unique id, clocks, barriers: CPU tensors) over gloo; torch's own RCCL communicator is only
            # created when --collective torch gathers with it (or the library's communicator cannot be formed)
            kw = {"device_id": torch.device("cuda", local)} if a.collective == "torch" else {}
            dist.init_process_group("cpu:gloo,cuda:nccl", **kw)
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
    # ---- the collective, decided BEFORE any cube is generated: a preflight that forms the communicator and pushes 128 bytes per
    # rank through it, and a ladder lib (atl_comm_* over RCCL) -> torch.distributed (RCCL) -> no collective at all, so that
    # a line with n_gpus = N is printed whatever the node's RCCL does.  Every rung is agreed on by all
    # ranks over the gloo control group; every wait has a time-out (atl_comm_init: $ATLITE_HIP_COMM_TIMEOUT_S; here: 90 s).
    equal = len(set(shard_lens)) == 1
    collective = parts > 1 or (a.debug_rccl_self and world == 1)
    simulate = bool(a.preflight_fail)  # --preflight-fail: walk the ladder with pretended failures (also on one GPU, over gloo)
    use_lib = collective and a.collective == "lib" and (not (a.emulate_shard or a.debug_gloo_one_gpu) or (simulate and not a.emulate_shard))
    comm = None
    lib_error = None
    preflight = {"ladder": []}
    collective_failed = False

    def all_ranks_ok(ok):
        if world == 1:
            return bool(ok)
        okf = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(okf, op=dist.ReduceOp.MIN)
        return int(okf.item()) == 1

    def with_timeout(fn, seconds=float(os.environ.get("ATLITE_BENCH_PREFLIGHT_TIMEOUT", 90))):
        """fn() on a helper thread; (result, None) or (None, error text) - a call that never returns is given up on."""
        import threading

        box = {}

        def run():
            try:
                box["r"] = fn()
            except Exception as e:  # noqa: BLE001
                box["e"] = repr(e)

        th = threading.Thread(target=run, daemon=True)
        th.start()
        th.join(seconds)
        if th.is_alive():
            return None, f"no answer within {seconds:.0f} s"
        return box.get("r"), box.get("e")

    def tiny_gather(how):
        """16 doubles per rank through the collective `how`; True iff every rank's block arrived where it belongs."""
        n = max(world, 1)
        src = torch.full((1, 16), float(rank + 1), dtype=torch.float64, device=dev)
        dst = torch.zeros((1, 16 * n), dtype=torch.float64, device=dev)
        if how == "lib":
            h_lens = (C.c_int64 * n)(*([16] * n))
            torch.cuda.synchronize(dev)  # (this runs on a helper thread: torch's "current device" there is not ours)
            _lib.check(ctx.lib.atl_allgather_time_v(comm.handle, src.data_ptr(), 1, h_lens, dst.data_ptr(), 16 * n))
            ctx.sync()
        else:
            dist.all_gather_into_tensor(dst.view(-1), src.view(-1))
        torch.cuda.synchronize(dev)
        want = torch.arange(1, n + 1, dtype=torch.float64, device=dev).repeat_interleave(16).view(1, -1)
        return bool(torch.equal(dst, want))

    if use_lib:
        # the library's own communicator (C ABI atl_comm_*): rank 0 draws the unique id, the control group ships it
        try:
            uid = D.RcclComm.unique_id() if rank == 0 else None
        except Exception as e:  # noqa: BLE001
            uid, lib_error = None, repr(e)
        if world > 1:
            box = [uid]
            dist.broadcast_object_list(box, src=0)
            uid = box[0]
        if uid is not None and a.preflight_fail not in ("lib", "all"):
            try:
                comm = D.RcclComm(ctx, max(world, 1), rank, uid)
            except Exception as e:  # noqa: BLE001
                lib_error = repr(e)
        elif uid is not None:
            lib_error = "--preflight-fail"
        ok = comm is not None
        if all_ranks_ok(ok):
            good, err = with_timeout(lambda: tiny_gather("lib"))
            ok = bool(good) and err is None
            lib_error = err or (None if ok else "the test all-gather returned wrong data")
        if not all_ranks_ok(ok):  # every rank or none: a rank without a communicator would leave the others inside the collective
            if comm is not None:
                try:
                    comm.abort()
                    comm.close()
                except Exception:  # noqa: BLE001
                    pass
            comm = None
        preflight["ladder"].append({"rung": "lib (atl_comm_init + atl_allgather_time_v)", "ok": comm is not None, "error": lib_error})
        if comm is None:
            use_lib = False
            if rank == 0:
                print(f"[bench] the library's RCCL communicator is not available ({lib_error}); trying torch.distributed",
                      file=sys.stderr)
    if collective and world > 1 and not use_lib and (not (a.emulate_shard or a.debug_gloo_one_gpu) or (simulate and not a.emulate_shard)):
        good, err = (None, "--preflight-fail") if a.preflight_fail in ("torch", "all") else with_timeout(lambda: tiny_gather("torch"))
        ok = all_ranks_ok(bool(good) and err is None)
        preflight["ladder"].append({"rung": "torch.distributed all_gather_into_tensor (RCCL)", "ok": ok,
                                    "error": err or (None if good else "wrong data")})
        if not ok:
            # last rung: every rank converts + aggregates its shard, nobody gathers; the line says so
            collective, collective_failed = False, True
            if rank == 0:
                print(f"[bench] no working collective on this node ({err}); timing the ranks' own steps WITHOUT a gather "
                      "(multi_gpu.collective_failed = true)", file=sys.stderr)
    if world > 1 and rank == 0:
        print(f"[bench] preflight: {json.dumps(preflight)}", file=sys.stderr)
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
    # RCCL path: the all-gather and the placement copy of a step run behind the NEXT step's kernel - a second set of
    # (piece, gather) buffers by step parity, the placement on a side stream, buffer reuse ordered by events; the timed
    # region ends with a device-wide synchronize, so every step's result is in place when the clock stops.  One
    # launch per step then: there is nothing left for sub-launches to hide.
    overlap = collective and (equal or use_lib) and not (a.emulate_shard or a.debug_gloo_one_gpu or a.no_step_overlap)
    P = a.pipeline if a.pipeline > 0 else (1 if parts == 1 or overlap or T_loc * S < 1.0e8 else 2)
    P = max(1, min(P, T_loc // 8 or 1))
    if use_lib:
        P = 1  # the library's all-gather places whole shards; with the gather behind the next kernel pieces hide nothing
    pe = D.time_partition(T_loc, P)  # sub-launch edges inside this rank's shard
    assert equal or world == 1 or P == 1, "pipelined gather needs equal shards"
    full = torch.empty((N, sum(shard_lens)), dtype=torch.float64, device=dev)  # (shapes x all time steps)
    piece = [torch.empty((N, pe[i + 1] - pe[i]), dtype=torch.float64, device=dev) for i in range(P)]
    gbuf = [torch.empty((parts, N, pe[i + 1] - pe[i]), dtype=torch.float64, device=dev) for i in range(P)] if collective and not use_lib else None
    step_no = [0]
    if overlap:
        piece2 = [piece, [torch.empty_like(t) for t in piece]]
        if not use_lib:
            gbuf2 = [gbuf, [torch.empty_like(t) for t in gbuf]]
            side = torch.cuda.Stream(device=dev)
        placed = [[None] * P, [None] * P]  # event / ticket: the gather + placement that last read piece[parity][i] is done
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
        """The timed call: ONE entry of the C ABI, the cubes' slot stride an argument of it."""
        _lib.check(ctx.lib.atl_pv_convert_aggregate_ld(ctx.handle, 0 if ld == S else ld, C.byref(pin_), C.byref(pp), T_, S, plan.handle, 0,
                                                       out_ptr, ld_out))

    graphs = {}  # --graph: (params identity, piece, output pointer) -> hipGraph of the piece's launches

    def launch(pp, i, out_t):
        if not a.graph:
            return cabi_pv(pins[i], pp, pe[i + 1] - pe[i], out_t.data_ptr(), out_t.stride(0))
        key = (id(pp), i, out_t.data_ptr(), plan.handle.value)
        g = graphs.get(key)
        if g is None:
            cabi_pv(pins[i], pp, pe[i + 1] - pe[i], out_t.data_ptr(), out_t.stride(0))  # once for real: the scratch arena settles
            _lib.check(ctx.lib.atl_capture_begin(ctx.handle))
            try:
                cabi_pv(pins[i], pp, pe[i + 1] - pe[i], out_t.data_ptr(), out_t.stride(0))
            finally:
                g = C.c_void_p()
                _lib.check(ctx.lib.atl_capture_end(ctx.handle, C.byref(g)))
            graphs[key] = g
        _lib.check(ctx.lib.atl_graph_launch(ctx.handle, g))

    def step(pp):
        if not collective:
            for i in range(P):  # P == 1 unless asked otherwise: straight into the result
                launch(pp, i, full[:, pe[i]:pe[i + 1]] if P == 1 else piece[i])
                if P > 1:
                    full[:, pe[i]:pe[i + 1]].copy_(piece[i])
            return full
        if use_lib:  # the library's own collective (atl_allgather_time_v[_async]): packs ragged shards, gathers, places
            if overlap:
                par = step_no[0] & 1
                step_no[0] += 1
                if placed[par][0] is not None:
                    comm.wait(placed[par][0])  # two steps back: long done, costs nothing
                launch(pp, 0, piece2[par][0])
                placed[par][0] = comm.gather_time_v_async(piece2[par][0].data_ptr(), N, shard_lens, full.data_ptr(), full.stride(0))
            else:
                launch(pp, 0, piece[0])
                h_lens = (C.c_int64 * len(shard_lens))(*shard_lens)
                _lib.check(ctx.lib.atl_allgather_time_v(comm.handle, piece[0].data_ptr(), N, h_lens, full.data_ptr(), full.stride(0)))
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
            if a.emulate_shard:  # rank 0's block goes straight to its place in the result: what a rank does besides the collective
                launch(pp, i, full3[:, 0, pe[i]:pe[i + 1]])
                works.append(None)
                continue
            launch(pp, i, piece[i])
            if a.debug_gloo_one_gpu:
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
            if not a.emulate_shard:  # [rank][N][Tc] blocks -> (N x rank x T_loc) in place, one strided copy
                full3[:, :, pe[i]:pe[i + 1]].copy_(gbuf[i].permute(1, 0, 2))
        return full

    def barrier():
        if world > 1:  # over the control group (a CPU tensor: gloo)
            dist.all_reduce(torch.zeros(1, dtype=torch.int32))

    def fence():
        barrier()
        torch.cuda.synchronize()  # device-wide: the communicator's own stream included
        barrier()

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
            tt = torch.tensor([dt], dtype=torch.float64)  # the control group (gloo) carries the clocks
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, k

    legs = None if a.legs == "all" else {v.strip() for v in a.legs.split(",") if v.strip()}

    def want(leg):
        return legs is None or leg in legs

    pp_main = pv_params(a.night_skip)
    leg_main = ("headline" if cfg["stored_angles"] else "c4_headline") if a.shape_kind == "tessellation" and not a.night_skip \
        else ("night_skip" if a.shape_kind == "tessellation" and cfg["stored_angles"] else None)
    dt, kms = timed(pp_main, a.steps, a.warmup)
    if a.graph:  # the kernel brackets are not part of a captured graph: the kernel's own time from plain launches afterwards
        a.graph = False
        kms = timed(pp_main, a.steps, 2)[1]
        a.graph = True
    assert len(kms) == a.steps * P, (len(kms), a.steps, P)
    ms_per_step = dt / a.steps * 1e3
    value = (T_loc if a.emulate_shard else T_total) * S / (dt / a.steps)
    k_step = kms.reshape(a.steps, P).sum(axis=1)  # fused-kernel time per step on this rank
    k_ms = float(k_step.mean())
    algo_bytes = bpc * T_loc * S

    # per-rank kernel time and the collective by itself (outside the timed region): what a rank's step is made of
    per_rank_kernel_ms = gather_ms = comm_info = None

    def diagnostics():
        mine = torch.tensor([k_ms], dtype=torch.float64)
        lst = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(lst, mine)
        per_rank = [float(v.item()) for v in lst]
        info = None
        if use_lib:  # what the communicator itself says: ranks in it (ncclCommCount), device ordinal (ncclCommCuDevice)
            ci = comm.info()
            mine_i = torch.tensor([ci["n_ranks"], ci["rank"], ci["device"], local], dtype=torch.int64)
            lst_i = [torch.zeros_like(mine_i) for _ in range(world)]
            dist.all_gather(lst_i, mine_i)
            info = {"ranks_seen": [int(v[0]) for v in lst_i], "rank_of": [int(v[1]) for v in lst_i],
                    "device_of_rank": [int(v[2]) for v in lst_i], "local_rank": [int(v[3]) for v in lst_i]}
        if a.debug_gloo_one_gpu or collective_failed:
            return per_rank, None, info
        reps = 10
        fence()
        t0 = time.perf_counter()
        for _ in range(reps):  # all-gather of every piece + the placement, nothing to hide behind
            if use_lib:
                h_lens = (C.c_int64 * len(shard_lens))(*shard_lens)
                src = piece2[0][0] if overlap else piece[0]
                _lib.check(ctx.lib.atl_allgather_time_v(comm.handle, src.data_ptr(), N, h_lens, full.data_ptr(), full.stride(0)))
                continue
            for i in range(P):
                dist.all_gather_into_tensor(gbuf[i].view(-1), piece[i].view(-1))
                if full3 is not None:
                    full3[:, :, pe[i]:pe[i + 1]].copy_(gbuf[i].permute(1, 0, 2))
        fence()
        tg = torch.tensor([(time.perf_counter() - t0) / reps * 1e3], dtype=torch.float64)
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        return per_rank, float(tg.item()), info

    if world > 1:
        try:  # diagnostics only (the same code path on every rank): nothing here may cost the run its result line
            per_rank_kernel_ms, gather_ms, comm_info = diagnostics()
        except Exception as e:  # noqa: BLE001
            print(f"[bench] rank {rank}: multi-GPU diagnostics skipped: {e!r}", file=sys.stderr)
    elif use_lib:
        comm_info = {"ranks_seen": [comm.info()["n_ranks"]], "device_of_rank": [comm.info()["device"]]}

    # the reassembled result holds every rank's block in place
    multi_parity = None
    if world > 1 and not collective_failed:
        res = step(pp_main)
        fence()
        own = torch.empty((N, T_loc), dtype=torch.float64, device=dev)
        cabi_pv(pin, pp_main, T_loc, own.data_ptr(), T_loc)
        torch.cuda.synchronize()
        assert torch.equal(res[:, edges[rank]:edges[rank + 1]], own), "all-gather misplaced this rank's block"
        chk = torch.stack([res.sum(), res.abs().sum()]).cpu()
        lst = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(lst, chk)
        assert all(torch.equal(v, chk) for v in lst), "ranks disagree on the gathered result"
        if not a.no_parity:
            # ... and an ORACLE sample of every rank's own shard (round 6: until then the multi-rank path checked placement and
            # agreement only): two dozen steps around the shard's first morning and its middle
            from oracle import atlite_oracle as orc

            sel = np.unique(np.clip(np.concatenate([np.arange(0, 12), np.arange(T_loc // 2, T_loc // 2 + 12)]), 0, T_loc - 1))
            host = {k: np.stack([v.slab(int(t), int(t) + 1).numpy()[0] for t in sel]) for k, v in inputs.items()}
            if tables is not None:
                al, az = orc.solar_position(synthetic.time_index(T_loc, "2013-01-01", off)[sel], x, y, "-30min")
                host["solar_altitude"], host["solar_azimuth"] = al.reshape(len(sel), S), az.reshape(len(sel), S)
            ref = orc.aggregate_matrix(orc.convert_pv(host, CSI, ORI), M)
            got = own.cpu().numpy()[:, sel]
            scale = np.abs(ref).max()
            err = float((np.abs(got - ref) / np.maximum(np.abs(ref), 1e-12 * scale)).max())
            okf = torch.tensor([1 if err <= 1e-10 else 0], dtype=torch.int32)
            dist.all_reduce(okf, op=dist.ReduceOp.MIN)
            multi_parity = {"checked_steps_per_rank": int(len(sel)), "max_rel_err_rank0": err, "rtol": 1e-10,
                            "ok_on_every_rank": bool(int(okf.item()) == 1)}

    shapes_word = ("a 100-cell Voronoi TESSELLATION (every cell covered: all 56 B/cell-step are read; BASELINE's overlapping "
                   "star polygons: star_polygons below)") if a.shape_kind == "tessellation" else "overlapping star-convex polygons"
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
            "workload": f"{a.config}: {N} shapes = {shapes_word}; pv CSi slope 30 az 180, {T_total}x{Y}x{X} fp64, "
                        f"aggregate_time=None, " + ("stored solar angles (7 cubes)" if cfg["stored_angles"]
                                                    else "in-kernel solar position (5 cubes)") +
                        "; timed call = atl_pv_convert_aggregate (C ABI) on a prebuilt plan, result left in HBM",
            "parallelism": f"time-sharded x{world}" + (f" + all-gather by {'the library (atl_allgather_time_v' + ('_async' if overlap else '') + ', RCCL)' if use_lib else 'torch.distributed (RCCL)'}, {P} piece(s) per step" if world > 1 else "") +
                           (", gather + placement of a step behind the next step's kernel" if overlap and world > 1 else "") +
                           (", a rank's launches replayed as one hipGraph" if a.graph else ""),
            "time_steps_per_gpu": T_loc,
            "night_skip": bool(a.night_skip),
            "layout": ("slot-interleaved: the cubes in ONE allocation, the variables of a time step side by side "
                       f"(cube v, slot t at base + (t * {len(inputs)} + v) * {ld // len(inputs)} cells) - the layout of the library's own "
                       "device copies (device.SlotPool, Dataset.device_group)") if a.layout == "interleaved"
                      else "one allocation per cube",
            "cell_tile": f"{plan_info['tile_w']}x{plan_info['tile_h']}",
            "partial_rows": plan_info["n_partial_rows"],
        },
        "roofline": roofline_of(leg_main if world == 1 and not a.emulate_shard and T_loc == CONFIGS[a.config]["T"] and
                                (Y, X, N) == (CONFIGS[a.config]["Y"], CONFIGS[a.config]["X"], CONFIGS[a.config]["shapes"]) else None,
                                algo_bytes, k_step, {"launches_per_step": P}),
    }
    if result["roofline"]["kernel"] is None:
        result["roofline"]["kernel"] = KERNELS["headline" if cfg["stored_angles"] else "c4_headline"] if not a.night_skip else KERNELS["night_skip"]
    if world > 1 or use_lib:
        result["multi_gpu"] = {
            "per_rank_kernel_ms": per_rank_kernel_ms,  # fused kernel(s) of one step on each rank's own shard
            "gather_ms": gather_ms,  # serial all-gather + placement of one step's result (max over ranks), untimed region
            "result_bytes": int(N * sum(shard_lens) * 8),
            "collective": ("library: atl_comm_init + atl_allgather_time_v" + ("_async (communicator's own stream)" if overlap else "")) if use_lib
                          else "NONE: no collective worked on this node; every rank timed its own shard, nothing was gathered" if collective_failed
                          else ("torch.distributed all_gather_into_tensor" + (f" (the library's communicator failed: {lib_error})" if lib_error else "")),
            "collective_failed": bool(collective_failed),
            "preflight": preflight["ladder"],
            "transport": "gloo on host copies (debug, all ranks on GPU 0)" if a.debug_gloo_one_gpu else "RCCL over xGMI",
        }
        if comm_info:
            result["multi_gpu"].update(comm_info)
        if multi_parity is not None:
            result["multi_gpu"]["parity"] = multi_parity
    if a.emulate_shard:
        # what one rank of an N-way strong-scaling run spends per step besides the collective
        result["emulated_shard"] = {
            "of": parts, "pieces": P, "step_ms": ms_per_step, "fused_kernel_ms": k_ms, "graph": bool(a.graph),
            "overhead_ms": ms_per_step - k_ms, "overhead_frac": (ms_per_step - k_ms) / ms_per_step,
            "note": "rank 0's shard of the strong-scaling split on one GPU: fused kernel(s) + k_combine (written straight to the "
                    "rank's place in the result) + host launch path; no collective",
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

    extras = rank == 0 and single and a.config == "c2" and not a.no_extras
    if extras:
        ks = max(3, min(a.steps, 10))
        # warm-up launches of the side legs: after the host-side gap before each of them the shader clock needs ~30 ms
        # of load to settle, and the early-out kernel is issue-bound (profiles/r03_clock_per_launch.txt: 2.3 -> 1.97 ms
        # over the first eight launches of a burst); the headline measurement above keeps the caller's --warmup
        kw_ = 12
        # (1) the Python API's default: night early-out (bit-identical output, fewer bytes read)
        if not a.night_skip and want("night_skip"):
            ref_out = step(pp_main).clone()
            dts_v, kk_v = timed(pv_params(True), ks, kw_)  # the tile's altitudes loaded and voted on (first call; a caller's own cubes)
            same_v = bool(torch.equal(step(pv_params(True)), ref_out))
            # ... and with the day map of (plan, altitude cube, cut-off), built once and kept with the cube: the API's steady state
            ld_m = (T_loc + 7) // 8 * 8  # one byte per (tile, time step)
            dmap = ctx.empty((max(plan_info["n_segments"], 1) * ld_m,), np.uint8)
            ctx.sync()
            ctx.timer_start()
            _lib.check(ctx.lib.atl_pv_day_map_ld(ctx.handle, 0 if ld == S else ld, C.byref(pin), C.byref(pv_params(True)), T_loc, S,
                                                 plan.handle, dmap.ptr, ld_m))
            map_ms = ctx.timer_stop()
            for q in pins:
                q.d_day_map, q.day_map_ld = dmap.ptr, ld_m
            dts, kk = timed(pv_params(True), ks, kw_)
            same = bool(torch.equal(step(pv_params(True)), ref_out)) and same_v
            for q in pins:
                q.d_day_map, q.day_map_ld = None, 0
            # algorithmic bytes of the early-out: the altitude cube in full + the six other cubes where a cell is up
            # (the fewest bytes any per-cell early-out could read; the kernel decides per 128-cell tile and slot)
            try:
                alt_t = device_view(inputs["solar_altitude"], torch)
                n_day = int((alt_t >= float(np.radians(1.0))).sum().item())
                night_bytes = 8 * T_loc * S + 48 * n_day
            except Exception as e:  # noqa: BLE001
                n_day, night_bytes = None, None
                print(f"[bench] day-cell count skipped: {e!r}", file=sys.stderr)
            result["night_skip"] = {"ms_per_step": dts / ks * 1e3, "value": T_total * S / (dts / ks), "bit_identical": same,
                                    "day_cell_steps": n_day, "day_map": True, "day_map_build_ms": map_ms,
                                    "day_cell_bytes": 56 * n_day if n_day else None,
                                    "voting_kernel": {"ms_per_step": dts_v / ks * 1e3, "kernel_ms": float(np.mean(kk_v)) if len(kk_v) else None,
                                                      "note": "no day map: the tile's altitudes loaded and voted on (round 4's kernel; first call, "
                                                              "or cubes the caller may rewrite)"},
                                    "roofline": roofline_of("night_skip", night_bytes or algo_bytes, kk,
                                                            {"note": "algorithmic bytes = 8 B x every cell-step (altitude) + 48 B x the "
                                                                     "cell-steps above the 1 degree cut-off; the kernel skips per tile and "
                                                                     "slot, so its traffic is a little above that"}
                                                            if night_bytes else {"note": "day-cell count unavailable: bytes of the full-read kernel"})}
        # (2) BASELINE's "random-polygon" shapes: overlapping star-convex polygons, cells may be uncovered
        if a.shape_kind == "tessellation" and want("star_polygons"):
            polys_s, M_s = shapes_of("star")
            plan_main, plan = plan, ctx.plan(M_s, row_len=X, ld=None if ld == S else ld)
            dts, kk = timed(pp_main, ks, kw_)
            info_s = plan.info()
            cov_mask = np.asarray((M_s != 0).sum(0)).ravel() > 0
            covered = int(cov_mask.sum())
            n_lines = lines_touched(cov_mask, ld)
            line_bytes = None if n_lines is None else n_lines * 16 * bpc * T_loc
            result["star_polygons"] = {
                "ms_per_step": dts / ks * 1e3, "value": T_total * S / (dts / ks),
                "partial_rows": info_s["n_partial_rows"], "cell_tile": f"{info_s['tile_w']}x{info_s['tile_h']}",
                "covered_cells": covered, "max_shapes_per_cell": int(np.asarray((M_s != 0).sum(0)).max()),
                "roofline": roofline_of("star_polygons", bpc * T_loc * covered, kk,
                                        {"note": "algorithmic bytes = 56 B x the cells some shape covers; whole "
                                                                          "128-byte lines are fetched along the ragged edges (traffic): "
                                                                          "line_bytes = the 128-byte lines that hold a covered cell, counted "
                                                                          "on the host from the indicator matrix",
                                         "lines_128B_per_slot": n_lines, "line_bytes": line_bytes,
                                         "frac_on_line_bytes": None if not line_bytes else line_bytes / (float(np.mean(kk)) * 1e-3) / 1e9 / PEAK_GBPS}),
            }
            tr_s = result["star_polygons"]["roofline"].get("traffic")
            if tr_s and line_bytes:
                result["star_polygons"]["roofline"]["traffic_over_line_bytes"] = tr_s / line_bytes
            plan = plan_main
        # (3) what a user of the drop-in API waits for: cutout.pv(...) on a device-resident Dataset
        if want("api"):
            from atlite_amd import Cutout, Dataset

            cut = Cutout(Dataset(dict(inputs), dict(time=synthetic.time_index(T_loc), y=y, x=x), static=True))
            kw = dict(panel="CSi", orientation={"slope": 30.0, "azimuth": 180.0}, aggregate_time=None)

            def call(**k):
                t0 = time.perf_counter()
                r = cut.pv(**kw, **k)
                return (time.perf_counter() - t0) * 1e3, r

            cold, r0 = call(shapes=polys)  # indicator matrix + plan build + kernel + D2H + labelled result
            warm = min(call(shapes=polys)[0] for _ in range(5))  # plan cached
            warm_m = min(call(matrix=M)[0] for _ in range(5))
            same = bool(np.array_equal(np.asarray(r0.values), step(pp_main).cpu().numpy()))
            from atlite_amd.device import default_context

            dctx = default_context()  # the context the public API runs on
            dctx.set_profiling(True)
            call(shapes=polys)
            result["api_e2e_ms"] = {"call": "cutout.pv(panel='CSi', orientation={slope:30,azimuth:180}, shapes=polys, "
                                            "aggregate_time=None) -> host (shapes x time) labelled array; night early-out on",
                                    "cold": cold, "warm": warm, "warm_matrix_given": warm_m, "kernel_ms": dctx.last_kernel_ms(),
                                    "equals_timed_result": same}
            dctx.set_profiling(False)
            del cut, r0

        # (3b) from a cutout FILE: NetCDF-4, fp32, zlib + shuffle - what atlite writes (atlite/data.py:246-248) -
        # opened with the library's own reader; the chunks' zlib streams are inflated on the device (one wavefront per stream),
        # PCIe carries the compressed bytes.  A month and a half of the C2 grid (the file is written here, with h5py under the
        # image's conda interpreter: ~7 s), the timed call is Cutout(path).pv(matrix=M) end to end.
        if want("from_file"):
            try:
                result["from_file"] = from_file_leg(ctx, _lib, gis, M if (Y, X) == (200, 200) else None, Y, X)
            except Exception as e:  # noqa: BLE001 - a side leg must not cost the run its line
                result["from_file"] = {"skipped": repr(e)}
            try:  # the same grid chunked (time = 100, y, x) - atlite's own chunks={"time": 100} written through: 16 MB streams, few
                # of them (a third of a year: 210).  A zlib stream is sequential - until round 6 these stayed on the host threads -
                # but its DEFLATE blocks are not: they are decoded side by side, a wave per block (atl_inflate_dev.h, "SEGMENTS")
                result["from_file_large_chunks"] = from_file_leg(ctx, _lib, gis, M if (Y, X) == (200, 200) else None, Y, X, T=3000,
                                                                 chunks=(100, Y, X), device_calls=2)
            except Exception as e:  # noqa: BLE001
                result["from_file_large_chunks"] = {"skipped": repr(e)}

        # (4) the same cubes in an allocation each (the layout a caller's own device arrays have, and the library's before
        # round 3): same kernel, same bytes, bit-identical result - the memory system alone makes the difference
        if a.layout == "interleaved" and not a.night_skip and want("separate_cubes"):
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
                                        "frac": algo_bytes / (ksep_ms * 1e-3) / 1e9 / PEAK_GBPS,
                                        "bit_identical": bool(np.array_equal(r_sep.numpy(), step(pp_main).cpu().numpy())),
                                        "note": "one allocation per cube instead of the slot-interleaved one; same kernel and bytes"}
            del sep, plan_sep, r_sep

    if rank == 0 and single and not a.no_cpu_baseline and cfg["stored_angles"] and want("cpu"):
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
        del host

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

    # ---- the other BASELINE.json configurations at N = 1, at their own sizes (after the C2 legs: their cubes go first) ----
    cfg_legs = [l for l in ("c3_series", "c3_cf_map", "c3_aggregated", "c5_heat", "c5_runoff", "odd_caller", "c2_sp", "c4_full_sp") if want(l)]
    if extras and cfg_legs:
        del inputs, plan, full, full3, piece, pin, pins
        import gc

        gc.collect()
        try:
            result["configs"] = config_legs(ctx, cfg_legs, max(3, min(a.steps, 6)), check=not a.no_parity)
        except Exception as e:  # noqa: BLE001 - never a reason to lose the headline line
            result["configs"] = {"error": repr(e)[:400]}

    # ---- N > 1: the two configurations BASELINE.json names for 8 GPUs, in the same run and the same JSON line ----
    wl = a.workloads if a.workloads is not None else ("c4,c5" if world > 1 else "none")
    wl = [w.strip() for w in wl.split(",") if w.strip() and w.strip() != "none"]
    if wl:
        # drop the C2 cubes and buffers (the closures above share these cells: rebinding the names releases the memory)
        inputs = plan = full = full3 = piece = pin = pins = gbuf = tables = piece2 = gbuf2 = placed = None  # noqa: F841
        import gc

        gc.collect()
        mode = "none" if (collective_failed or world == 1) else "gloo-debug" if a.debug_gloo_one_gpu else "lib" if use_lib else "torch"
        result["workloads"] = {}
        for name in wl:
            try:  # each workload on its own: one failure must not cost the others' numbers, nor the C2 line
                result["workloads"][name] = sharded_workload(
                    name, ctx, a, rank, world, dev, torch, dist, comm if use_lib else None, mode, barrier, fence, D, _lib, gis, synthetic, solar,
                    parts=parts)
            except Exception as e:  # noqa: BLE001
                import traceback

                result["workloads"][name] = {"error": repr(e)[:300], "where": traceback.format_exc()[-600:]}
            gc.collect()

    if rank == 0:
        print(json.dumps(result))
    if comm is not None:
        comm.close()
    if world > 1:
        barrier()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
