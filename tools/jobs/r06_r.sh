#!/bin/bash
# round 6: dark batches of the day-map kernel written as zeros without a reduction: tests, A/B against the commit before (variants/lib_prev.so)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_r
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
V=$REPO/atlite_amd/lib/variants
timeout 600 python -X faulthandler -m pytest tests/test_gpu_day_map.py tests/test_gpu_parity.py tests/test_gpu_api_golden.py -x -q -m gpu -p no:cacheprovider > $OUT/tests.log 2>&1
echo "tests rc=$? $(grep -E 'passed|failed' $OUT/tests.log | tail -1)"; grep -E "^(FAILED|ERROR)" $OUT/tests.log | head
show() { python - <<PY
import json
j = json.loads(open("$1").read().strip().splitlines()[-1])
n = j["night_skip"]
print("$2 headline %.4f ms | night_skip %.4f ms kernel %.4f bit_identical %s voting %.4f | api warm %.3f" % (j["ms_per_step"], n["ms_per_step"], n["roofline"].get("kernel_ms"), n["bit_identical"], n["voting_kernel"]["ms_per_step"], j.get("api_e2e_ms", {}).get("warm", 0)))
PY
}
for i in 1 2 3; do
  ATLITE_HIP_LIB=$V/lib_prev.so timeout 300 python bench.py --legs headline,night_skip,api --no-cpu-baseline --steps 20 > $OUT/prev$i.json 2> $OUT/prev$i.err; show $OUT/prev$i.json prev$i
  timeout 300 python bench.py --legs headline,night_skip,api --no-cpu-baseline --steps 20 > $OUT/new$i.json 2> $OUT/new$i.err; show $OUT/new$i.json new$i
done
