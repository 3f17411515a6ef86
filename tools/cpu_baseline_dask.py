#!/usr/bin/env python3
"""
BASELINE.md section 3, variant C: the pv convert + aggregate chain expressed on ``dask.array`` - chunks {"time": 100},
threaded scheduler - timed on the host cores.  Runs under /opt/conda/bin/python3.9 (dask 2021.10.0 = the reference's
minimum pin, numpy 1.26, scipy 1.7; the default interpreter has no dask):

    /opt/conda/bin/python3.9 tools/cpu_baseline_dask.py [T' = 800] [threads = all]

The arithmetic is the oracle's (oracle/atlite_oracle.py: the reference's operation sequence on NumPy functions), which
numpy's __array_function__ protocol routes to dask.array when it is handed dask arrays - the same lazy, chunked graph
the reference builds through xarray (atlite/convert.py:840-854, pv/irradiation.py:196-255, pv/solar_panel_model.py:22-41;
the per-chunk sparse product of atlite/aggregate.py:21-32 as a map_blocks).  Variants A (eager NumPy, 1 thread) and B
(the oracle over time chunks of 100 on a thread pool) are timed beside it.  Test infrastructure: nothing in atlite_amd
imports this.
"""
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import dask  # noqa: E402
import dask.array as da  # noqa: E402

from oracle import atlite_oracle as orc  # noqa: E402
from tests import helpers as H  # noqa: E402

CSI = dict(c_temp_amb=1, c_temp_irrad=0.035, r_tmod=298, r_irradiance=1000, k_1=-0.017162, k_2=-0.040289,
           k_3=-0.004681, k_4=0.000148, k_5=0.000169, k_6=0.000005, inverter_efficiency=0.9)
ORI = dict(slope=np.radians(30.0), azimuth=np.radians(180.0))


def usable_cpus():
    n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(p))))
    except Exception:
        pass
    return n


def main():
    Tn = int(sys.argv[1]) if len(sys.argv) > 1 else 800
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else usable_cpus()
    Y = X = 200
    S = Y * X
    ds = H.pv_dataset(Tn, Y, X, seed=1)  # (T', S) fp64, seven cubes, physically consistent (night, twilight, noon)
    M = H.blob_matrix(100, Y, X, seed=2).tocsr()
    MT = M.T.tocsr()
    res = {"workload": f"pv CSi slope 30 az 180, {Tn} x {Y} x {X} fp64 (a slab of C2's grid), 100 shapes", "cell_steps": Tn * S,
           "host_threads_used": threads, "os_cpu_count": os.cpu_count(), "versions": dict(dask=dask.__version__, numpy=np.__version__)}

    def chunk_job(c):
        sub = {k: v[c[0]:c[1]] for k, v in ds.items()}
        return orc.aggregate_matrix(orc.convert_pv(sub, CSI, ORI), M, dask_branch=True)

    # A: eager NumPy, one thread, whole slab at once in chunks of 100 (the same arithmetic, no scheduler)
    T1 = min(Tn, 400)
    t0 = time.perf_counter()
    ref = np.concatenate([chunk_job((a, min(a + 100, T1))) for a in range(0, T1, 100)], axis=0)
    dt = time.perf_counter() - t0
    res["A_numpy_1_thread"] = dict(seconds=dt, steps=T1, cell_steps_per_s=T1 * S / dt)
    # B: the oracle over time chunks of 100 on a thread pool
    chunks = [(a, min(a + 100, Tn)) for a in range(0, Tn, 100)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        outB = np.concatenate(list(ex.map(chunk_job, chunks)), axis=0)
    dt = time.perf_counter() - t0
    res["B_numpy_thread_pool"] = dict(seconds=dt, steps=Tn, cell_steps_per_s=Tn * S / dt)
    # C: the chain on dask.array, chunks {"time": 100}, threaded scheduler
    dds = {k: da.from_array(v, chunks=(100, S)) for k, v in ds.items()}
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        conv = orc.convert_pv(dds, CSI, ORI)  # lazy: a dask graph of the reference's elementwise chain
        assert isinstance(conv, da.Array), type(conv)
        agg = conv.map_blocks(lambda b: np.asarray(b @ MT), chunks=(conv.chunks[0], (M.shape[0],)), dtype=np.float64)
        outC = agg.compute(scheduler="threads", num_workers=threads)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    res["C_dask_array_threads"] = dict(seconds=best, steps=Tn, cell_steps_per_s=Tn * S / best, graph_tasks=len(agg.__dask_graph__()))
    scale = np.abs(outB).max()
    res["C_equals_B_max_rel"] = float(np.max(np.abs(outC - outB) / np.maximum(np.abs(outB), 1e-12 * scale)))
    res["B_equals_A_max_rel"] = float(np.max(np.abs(outB[:T1] - ref) / np.maximum(np.abs(ref), 1e-12 * scale)))
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
