#!/bin/bash
# round 6: line-granular day map: its tests, then the night_skip leg with counters (A/B against the library of the commit before: variants/lib_prev.so)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_l
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
timeout 600 python -X faulthandler -m pytest tests/test_gpu_day_map.py tests/test_gpu_aligned_plans.py tests/test_gpu_parity.py -x -q -m gpu -p no:cacheprovider > $OUT/tests.log 2>&1
echo "tests rc=$? $(tail -1 $OUT/tests.log)"; grep -E "^(FAILED|ERROR)|Error" $OUT/tests.log | head
for i in 1 2; do
timeout 300 python bench.py --legs headline,night_skip --no-cpu-baseline --steps 20 > $OUT/new$i.json 2> $OUT/new$i.err
python - <<PY
import json
j = json.loads(open("$OUT/new$i.json").read().strip().splitlines()[-1])
n = j["night_skip"]
print("new$i headline %.4f ms | night_skip %.4f ms kernel %s bit_identical %s map build %.3f ms voting %.4f" % (j["ms_per_step"], n["ms_per_step"], n["roofline"].get("kernel_ms"), n["bit_identical"], n["day_map_build_ms"], n["voting_kernel"]["ms_per_step"]))
PY
done
