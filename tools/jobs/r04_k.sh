#!/bin/bash
# full GPU suite, the configs profile group (new series kernel) and a bench line
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_k
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $OUT/tests.log
cp profiles/bench_profile_latest.json $OUT/
timeout 900 python tools/profile_bench.py $OUT configs > $OUT/profile.log 2>&1
cp $OUT/bench_profile_latest.json profiles/bench_profile_latest.json
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/tests.log; tail -8 $OUT/profile.log; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_k/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"])
for k,v in d["configs"].items(): print(k, {a:b for a,b in v.items() if a in ("kernel_ms","frac","traffic","frac_from_profile","parity_rel","roofline")})
PY
