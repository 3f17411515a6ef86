#!/usr/bin/env python3
"""
rocprofv3 target of round 3: every dominant kernel of the library on its workload, a few launches each, so that ONE
set of passes (--kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE / SQ counters, tools/profile_configs.sh) covers a
whole group.  usage: profile_all.py <group>
  pvfam  C2 shape: pv() defaults with and without the early-out, in-kernel solar position (both), the influx / outflux
         head, the general kernel (influx + Hay-Davies), a tracker, per-cell series and capacity-factor maps (k_cells_*)
  cfg    C3 wind (series, map, aggregated), C5 shard heat demand + runoff, C4 shard pv
  dense  matrix x layout shaped plans (16 / 32 partial rows per tile): runoff, wind, pv on the matrix cores
"""
import os
import runpy
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
group = sys.argv[1] if len(sys.argv) > 1 else "pvfam"
os.environ.setdefault("ATL_VARIANT_REPS", "4")
os.environ.setdefault("ATL_VARIANT_WARMUP", "2")
os.environ.setdefault("ATL_CFG_WARMUP", "2")
os.environ.setdefault("ATL_CFG_REPS", "4")
if group == "pvfam":
    os.environ.setdefault("ATL_VARIANTS", "|".join([
        "getter, scalar orientation", "getter + night early-out", "in-kernel solar position", "influx / outflux dataset",
        "pv(tracking='horizontal') - fast", "pv(tracking='horizontal') + night", "per-cell series out", "per-cell time-mean"]))
    sys.argv = [sys.argv[0]]
    runpy.run_path(str(ROOT / "tools" / "bench_pv_variants.py"), run_name="__main__")
elif group == "cfg":
    sys.argv = [sys.argv[0], "C3", "C3m", "C3a", "C5h", "C5r", "C4s"]
    runpy.run_path(str(ROOT / "tools" / "bench_configs.py"), run_name="__main__")
elif group == "dense":
    os.environ.setdefault("ATL_DENSE_R", "16,32")
    sys.argv = [sys.argv[0], "runoff", "wind", "pv"]
    runpy.run_path(str(ROOT / "tools" / "bench_dense.py"), run_name="__main__")
else:
    sys.exit(f"unknown group {group!r}")
