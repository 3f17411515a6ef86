#!/bin/bash
# round 5, job c - the night early-out's day map (tests, A/B against the voting kernel and against the unpipelined
# variant), the persistent flat series kernel (A/B), the fall-through ladder of bench.py --gpus N on one GPU
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05_c
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
echo "== 1. tests"; date +%T
timeout 900 python -X faulthandler -m pytest tests/test_gpu_day_map.py tests/test_gpu_aligned_plans.py tests/test_gpu_post.py \
  tests/test_gpu_multidevice.py::test_bench_prints_its_line_when_no_collective_works tests/test_gpu_parity.py -x -q > $OUT/tests.log 2>&1
echo "rc=$? $(tail -1 $OUT/tests.log)"; grep -E "^(FAILED|ERROR)|Error" $OUT/tests.log | head -10
echo "== 2. night early-out: day map vs vote (bench leg), pipelined vs not"; date +%T
timeout 600 python bench.py --legs night_skip,api --no-cpu-baseline > $OUT/night.json 2> $OUT/night.err
python - <<'PY'
import json
j=json.loads([l for l in open("gpurun_out/r05_c/night.json") if l.startswith("{")][-1])
n=j["night_skip"]; print("headline ms", j["ms_per_step"], "kernel", j["roofline"]["kernel_ms"])
print("night day-map ms/step", n["ms_per_step"], "kernel_ms", n["roofline"]["kernel_ms"], "frac", n["roofline"]["frac"], "build_ms", n["day_map_build_ms"], "bit_identical", n["bit_identical"])
print("night voting  ms/step", n["voting_kernel"]["ms_per_step"], "kernel_ms", n["voting_kernel"]["kernel_ms"])
print("api", j.get("api_e2e_ms"))
PY
ATLITE_HIP_LIB=$REPO/atlite_amd/lib/variants/lib_nopipe.so timeout 600 python bench.py --legs night_skip --no-cpu-baseline > $OUT/night_nopipe.json 2> $OUT/night_nopipe.err
python - <<'PY'
import json
j=json.loads([l for l in open("gpurun_out/r05_c/night_nopipe.json") if l.startswith("{")][-1])
n=j["night_skip"]; print("UNPIPELINED day-map ms/step", n["ms_per_step"], "kernel_ms", n["roofline"]["kernel_ms"])
PY
echo "== 3. C3 per-cell series: flat vs persistent"; date +%T
for p in 0 4 8 16; do
  ATLITE_HIP_SERIES_PERSIST=$p timeout 600 python bench.py --legs c3_series --no-cpu-baseline > $OUT/c3_p$p.json 2> $OUT/c3_p$p.err
  python - $p <<'PY'
import json,sys
j=json.loads([l for l in open(f"gpurun_out/r05_c/c3_p{sys.argv[1]}.json") if l.startswith("{")][-1])
c=j["configs"]["c3_series"]; print("persist", sys.argv[1], {k: c[k] for k in c if k in ("ms","kernel_ms","frac","parity","ms_per_step")} or c)
PY
done
date +%T
