#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_p
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_p/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "kernel_ms", d["roofline"]["kernel_ms"])
print("night", {k: d["night_skip"].get(k) for k in ("kernel_ms","ms_per_step")})
print("api", d.get("api_e2e_ms"))
for k,v in d["configs"].items(): print(k, {a:b for a,b in v.items() if a in ("kernel_ms","frac","parity_rel","frac_from_profile")})
PY
