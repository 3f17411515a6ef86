#!/usr/bin/env python3
"""rocprofv3 target for the kernels the bench line does not show: the three C3 wind kernels (series, capacity-factor
map, aggregated) and a selection of the pv family (general kernel on an influx-only dataset, trackers, bofinger, with
and without the night early-out) on the C2 shape.  Driven by tools/r02_job_extra_prof.sh."""
import os
import runpy
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("ATL_VARIANT_REPS", "2")
os.environ.setdefault("ATL_VARIANTS", "influx / outflux dataset|pv(tracking='horizontal')|KANENA|bofinger + tracking='horizontal'|irradiation(tracking='dual')")
sys.argv = [sys.argv[0], "C3", "C3m", "C3a"]
runpy.run_path(str(ROOT / "tools" / "bench_configs.py"), run_name="__main__")
runpy.run_path(str(ROOT / "tools" / "bench_pv_variants.py"), run_name="__main__")
