#!/usr/bin/env python3
"""Copy the summaries of a tools/r03_job7.sh sweep (gpurun_out/r03_job7) into profiles/ and rebuild profiles/pmc_latest.json
(the counter traffic bench.py quotes as roofline.traffic) from the per-workload json files of that sweep."""
import json
import shutil
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
J = ROOT / "gpurun_out" / "r03_job7"
P = ROOT / "profiles"

for f in sorted((J / "summ").glob("r03_*")):
    if f.suffix == ".err":
        continue
    shutil.copy(f, P / f.name)
for src, dst in (("bench_c2.json", "r03_bench_c2.json"), ("bench_c4_n1.json", "r03_bench_c4_n1.json"), ("gloo8.json", "r03_gloo8_one_gpu.json"),
                 ("rccl_self.json", "r03_rccl_self.json"), ("shard8.json", "r03_strong_shard8_overhead.json"), ("configs.log", "r03_configs.log"),
                 ("dense.log", "r03_dense.log"), ("pv_variants.log", "r03_pv_variants.log"), ("spread.log", "r03_launch_spread.log")):
    if (J / src).exists():
        text = "\n".join(l for l in (J / src).read_text().splitlines() if "amdgpu.ids" not in l)
        if src.endswith(".json"):
            text = [l for l in text.splitlines() if l.startswith("{")][-1]
        (P / dst).write_text(text + "\n")
old = json.loads((P / "pmc_latest.json").read_text())
work = {}
for f in sorted(P.glob("r03_pv_c*.json")):
    j = json.loads(f.read_text())
    if "workload" in j and "hbm_bytes_per_launch" in j:
        work[j["workload"]] = j
old["workloads"].update(work)
(P / "pmc_latest.json").write_text(json.dumps(old, indent=1) + "\n")
for name in ("r03_bench_c2.json", "r03_bench_c4_n1.json", "r03_strong_shard8_overhead.json"):
    j = json.loads((P / name).read_text())
    r = j["roofline"]
    print(name, f"value {j['value']:.4e} ms/step {j['ms_per_step']:.3f} kernel {r['kernel_ms']:.3f} frac {r['frac']:.3f} traffic {r['traffic']}",
          {k: (round(v['kernel_ms'], 3) if isinstance(v, dict) and 'kernel_ms' in v else None) for k, v in j.items()
           if k in ('night_skip', 'star_polygons', 'separate_cubes')}, j.get("api_e2e_ms", {}).get("warm"), j.get("cpu_baseline", {}).get("value"))
