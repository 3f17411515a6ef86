#!/bin/bash
# the whole GPU suite, unbuffered, then the bench line (default arguments)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-r06_suite}
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
hostname > $OUT/box.txt
timeout 1200 stdbuf -o0 -e0 python -X faulthandler -m pytest tests -m gpu -x -q --capture=no -p no:cacheprovider > $OUT/tests.log 2>&1
echo "suite rc=$? $(grep -E 'passed|failed' $OUT/tests.log | tail -1)"; grep -E "^(FAILED|ERROR)" $OUT/tests.log | head
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; tail -c 600 $OUT/bench.err
python - <<PY
import json
j = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("value %.4g ms_per_step %.4f frac %.3f" % (j["value"], j["ms_per_step"], j["roofline"]["frac"]))
for k in ("night_skip", "star_polygons", "api_e2e_ms", "from_file", "from_file_large_chunks"):
    v = j.get(k)
    if not v: continue
    if k.startswith("from_file"):
        print(k, {m: (v.get(m) or {}).get("fp64_equivalent_GBps") for m in ("device_inflate", "host_inflate")}, v.get("bit_identical"), v.get("skipped"), (v.get("device_inflate") or {}).get("best_s"))
    else:
        print(k, {m: v.get(m) for m in ("ms_per_step", "cold", "warm", "kernel_ms", "bit_identical")})
for k, v in (j.get("configs") or {}).items():
    if isinstance(v, dict): print(k, v.get("ms"), v.get("frac"), (v.get("parity") or {}).get("ok"))
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
