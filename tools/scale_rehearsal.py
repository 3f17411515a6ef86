#!/usr/bin/env python3
"""
Rehearsal of the 8-GPU runs on ONE GPU: for the three workloads `bench.py --gpus N` emits - C2 strong
scaling, BASELINE configs[3] (pv 8760x800x800, in-kernel solar position) and configs[4] (heat demand + runoff 35040x400x400) -

  1. the whole workload on one GPU                      (bench.py --workloads c4,c5: one shard = everything)
  2. rank 0's shard of the 8-way split, no collective   (bench.py --emulate-shard 8 --workloads c4,c5)
  3. the ragged all-gather of each workload's result among 8 virtual ranks on this GPU, over the library's in-process
     transport (atl_comm_init_local: the same atl_allgather_time_v code path as RCCL, peer copies instead of xGMI)
  4. two real processes sharing the GPU, control plane and collective over gloo (bench.py --gpus 2 --debug-gloo-one-gpu)

and the speed-up at 8 GPUs these predict: whole step / (rank 0's shard step + the gather, none of it hidden).  The gather in 3
runs on one GPU's copy engines, i.e. it moves 8x the bytes one rank moves in a real run through one device: an upper bound of a
rank's share; xGMI latency is not in it.  Writes profiles/r06_scale_rehearsal.json.

    python tools/scale_rehearsal.py [out.json] [--scale 0.05]     (--scale: a fraction of the time axes, for a quick check)
"""
import json
import os
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def bench(args, env=None, timeout=3000):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, env=e, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": r.stderr[-1500:], "returncode": r.returncode, "seconds": time.perf_counter() - t0}
    j = json.loads(lines[-1])
    j["_seconds"] = time.perf_counter() - t0
    return j


def local_gathers(n, shapes, reps=10):
    """ms of one blocking ragged all-gather among n virtual ranks on device 0 for every (N rows, per-rank lens) in shapes."""
    from atlite_amd.device import Context
    from atlite_amd.distributed import LocalComm, LocalGroup

    ctxs = [Context(0) for _ in range(n)]
    grp = LocalGroup(n)
    pool = ThreadPoolExecutor(n)
    comms = list(pool.map(lambda r: LocalComm(ctxs[r], grp, r), range(n)))
    out = {}
    try:
        for name, (N, lens) in shapes.items():
            def rank(r):
                ctx, comm = ctxs[r], comms[r]
                local = ctx.zeros((N, max(lens[r], 1)))
                full = ctx.empty((N, sum(lens))).no_recycle()
                comm.gather_time_v(local, N, lens, out=full)  # warm
                ctx.sync()
                t0 = time.perf_counter()
                for _ in range(reps):
                    comm.gather_time_v(local, N, lens, out=full)
                ctx.sync()
                return (time.perf_counter() - t0) / reps * 1e3

            out[name] = {"rows": N, "slots_per_rank": lens, "result_bytes": int(N * sum(lens) * 8), "ms": max(pool.map(rank, range(n)))}
    finally:
        for c in comms:
            c.close()
        grp.close()
        pool.shutdown()
        for c in ctxs:
            c.close()
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    out_path = Path(args[0]) if args else ROOT / "profiles" / "r06_scale_rehearsal.json"
    scale = None
    if "--scale" in sys.argv:
        scale = sys.argv[sys.argv.index("--scale") + 1]
    env = {"ATL_BENCH_WORKLOAD_SCALE": scale} if scale else {}
    c2T = ["--T", str(max(960, int(8760 * float(scale)) // 24 * 24))] if scale else []
    common = ["--no-cpu-baseline", "--no-extras", "--steps", "10", "--warmup", "3", "--workload-steps", "5", *c2T]
    from atlite_amd import distributed as D

    rec = {"what": __doc__.strip().split("\n\n")[0], "scale_of_the_time_axes": float(scale) if scale else 1.0, "ranks": 8}
    whole = bench(["--workloads", "c4,c5", *common], env)
    shard = bench(["--emulate-shard", "8", "--workloads", "c4,c5", *common], env)
    two = bench(["--gpus", "2", "--debug-gloo-one-gpu", "--workloads", "c4,c5", *common], env)
    rec["two_processes_one_gpu_gloo"] = {k: two.get(k) for k in ("n_gpus", "ms_per_step", "value", "multi_gpu", "workloads", "error", "_seconds")}
    # the results' shapes -> the gathers among 8 virtual ranks
    T2 = int(c2T[1]) if c2T else 8760
    f = float(scale) if scale else 1.0
    T4 = max(48 * 8, int(8760 * f) // 24 * 24)
    T5 = max(48 * 8, int(35040 * f) // 24 * 24)
    l2 = np.diff(D.time_partition(T2, 8)).tolist()
    l4 = np.diff(D.time_partition(T4, 8)).tolist()
    l5 = np.diff(D.time_partition(T5, 8, align=24)).tolist()
    g = local_gathers(8, {"c2": (100, l2), "c4": (500, l4), "c5_heat_demand": (50, [-(-v // 24) for v in l5]), "c5_runoff": (50, l5)})
    rec["gather_8_virtual_ranks_local_transport"] = g
    rows = {}
    for name in ("c2", "c4", "c5"):
        if name == "c2":
            w_ms, s_ms = whole.get("ms_per_step"), shard.get("ms_per_step")
            gm = g["c2"]["ms"]
        else:
            w = (whole.get("workloads") or {}).get(name, {})
            s = (shard.get("workloads") or {}).get(name, {})
            w_ms, s_ms = w.get("ms_per_step"), s.get("ms_per_step")
            gm = g["c4"]["ms"] if name == "c4" else g["c5_heat_demand"]["ms"] + g["c5_runoff"]["ms"]
            rows[name + "_parity"] = {"whole": w.get("parity"), "shard": s.get("parity")}
        rows[name] = {"whole_step_ms_1_gpu": w_ms, "rank0_shard_step_ms": s_ms, "gather_ms_unhidden": gm,
                      "predicted_speedup_at_8_gpus": (w_ms / (s_ms + gm)) if w_ms and s_ms else None,
                      "predicted_speedup_gather_hidden": (w_ms / s_ms) if w_ms and s_ms else None}
    rec["prediction"] = rows
    rec["whole_line_errors"] = {k: v.get("error") for k, v in (("whole", whole), ("shard", shard), ("two", two)) if v.get("error")}
    out_path.write_text(json.dumps(rec, indent=1))
    print(json.dumps(rec["prediction"], indent=1))
    print("written", out_path)


if __name__ == "__main__":
    main()
