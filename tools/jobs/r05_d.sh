#!/bin/bash
# round 5, job d - device inflate after the fixes (descriptor count, slot streams no longer chained, scalar decode loop,
# smaller tables, word-wise flush), then job c's content: day map, persistent series kernel, the bench ladder
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05_d
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
echo "== 1. ingest tests"; date +%T
timeout 420 python -X faulthandler -m pytest tests/test_gpu_ingest.py -x -q > $OUT/ingest_tests.log 2>&1
echo "rc=$? $(tail -1 $OUT/ingest_tests.log)"; grep -E "^(FAILED|ERROR)|Error|assert " $OUT/ingest_tests.log | head -10
echo "== 2. from-file bench, T=1440 and T=4380"; date +%T
timeout 300 python tools/bench_ingest.py --T 1440 --quick --keep /tmp/c1440.nc > $OUT/ingest_1440.log 2>&1
cut -c1-250 $OUT/ingest_1440.log
date +%T
timeout 480 python tools/bench_ingest.py --T 4380 --quick --keep /tmp/c4380.nc > $OUT/ingest_4380.log 2>&1
cut -c1-250 $OUT/ingest_4380.log
rm -f /tmp/c4380.nc
echo "== 3. tests: day map, aligned refusals, post, parity, bench ladder"; date +%T
timeout 600 python -X faulthandler -m pytest tests/test_gpu_day_map.py tests/test_gpu_aligned_plans.py tests/test_gpu_post.py \
  tests/test_gpu_multidevice.py::test_bench_prints_its_line_when_no_collective_works tests/test_gpu_parity.py -x -q > $OUT/tests.log 2>&1
echo "rc=$? $(tail -1 $OUT/tests.log)"; grep -E "^(FAILED|ERROR)|Error" $OUT/tests.log | head -10
echo "== 4. night early-out: day map vs vote, pipelined vs not"; date +%T
timeout 300 python bench.py --legs night_skip,api --no-cpu-baseline > $OUT/night.json 2> $OUT/night.err
python - <<'PY'
import json
try:
    j=json.loads([l for l in open("gpurun_out/r05_d/night.json") if l.startswith("{")][-1])
    n=j["night_skip"]; print("headline ms", j["ms_per_step"], "kernel", j["roofline"]["kernel_ms"])
    print("night day-map ms/step", n["ms_per_step"], "kernel_ms", n["roofline"]["kernel_ms"], "frac", n["roofline"]["frac"], "build_ms", n["day_map_build_ms"], "bit_identical", n["bit_identical"])
    print("night voting  ms/step", n["voting_kernel"]["ms_per_step"], "kernel_ms", n["voting_kernel"]["kernel_ms"])
    print("api", j.get("api_e2e_ms"))
except Exception as e:
    print("night leg failed", repr(e)); print(open("gpurun_out/r05_d/night.err").read()[-1500:])
PY
ATLITE_HIP_LIB=$REPO/atlite_amd/lib/variants/lib_nopipe.so timeout 300 python bench.py --legs night_skip --no-cpu-baseline > $OUT/night_nopipe.json 2> $OUT/night_nopipe.err
python - <<'PY'
import json
try:
    j=json.loads([l for l in open("gpurun_out/r05_d/night_nopipe.json") if l.startswith("{")][-1])
    n=j["night_skip"]; print("UNPIPELINED day-map ms/step", n["ms_per_step"], "kernel_ms", n["roofline"]["kernel_ms"])
except Exception as e:
    print("nopipe leg failed", repr(e))
PY
echo "== 5. C3 per-cell series: flat vs persistent"; date +%T
for p in 0 8 16; do
  ATLITE_HIP_SERIES_PERSIST=$p timeout 300 python bench.py --legs c3_series --no-cpu-baseline > $OUT/c3_p$p.json 2> $OUT/c3_p$p.err
  python - $p <<'PY'
import json,sys
try:
    j=json.loads([l for l in open(f"gpurun_out/r05_d/c3_p{sys.argv[1]}.json") if l.startswith("{")][-1])
    c=j["configs"]["c3_series"]; print("persist", sys.argv[1], "ms", c["ms"], "frac", c["frac"], "parity", c.get("parity",{}).get("ok"))
except Exception as e:
    print("c3 leg failed", sys.argv[1], repr(e))
PY
done
date +%T
