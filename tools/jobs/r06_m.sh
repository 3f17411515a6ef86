#!/bin/bash
# round 6: A/B on one box: line-granular day map vs the tile-granular one (variants/lib_prev.so = the commit before), with FETCH_SIZE;
# per-cell series with 1 / 2 / 4 / 8 pairs per thread (variants/lib_fp*.so; the product has 4)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_m
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
V=$REPO/atlite_amd/lib/variants
timeout 300 python -X faulthandler -m pytest tests/test_gpu_day_map.py -x -q -m gpu -p no:cacheprovider > $OUT/tests.log 2>&1
echo "tests rc=$? $(tail -1 $OUT/tests.log)"
show() { python - <<PY
import json
j = json.loads(open("$1").read().strip().splitlines()[-1])
n = j.get("night_skip")
if n: print("$2 headline %.4f ms | night_skip %.4f ms kernel %.4f bit_identical %s voting %.4f" % (j["ms_per_step"], n["ms_per_step"], n["roofline"].get("kernel_ms"), n["bit_identical"], n["voting_kernel"]["ms_per_step"]))
c = (j.get("configs") or {}).get("c3_series")
if c: print("$2 c3_series %.4f ms median %.4f frac %.3f parity %s" % (c["ms"], c["ms_median"], c["frac"], (c.get("parity") or {}).get("ok")))
PY
}
for i in 1 2; do
  ATLITE_HIP_LIB=$V/lib_prev.so timeout 300 python bench.py --legs headline,night_skip --no-cpu-baseline --steps 20 > $OUT/prev$i.json 2> $OUT/prev$i.err; show $OUT/prev$i.json prev$i
  timeout 300 python bench.py --legs headline,night_skip --no-cpu-baseline --steps 20 > $OUT/new$i.json 2> $OUT/new$i.err; show $OUT/new$i.json new$i
done
# FETCH_SIZE of the mapped kernel, both libraries
for L in prev new; do
  if [ $L = prev ]; then export ATLITE_HIP_LIB=$V/lib_prev.so; else unset ATLITE_HIP_LIB; fi
  (cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_$L -o run -- python $REPO/bench.py --legs headline,night_skip --no-cpu-baseline --no-parity --steps 5 > $OUT/pmc_$L.log 2>&1)
done
unset ATLITE_HIP_LIB
python - <<PY
import sqlite3, glob
for L in ("prev", "new"):
    fs = glob.glob("$OUT/pmc_%s/**/*.db" % L, recursive=True)
    if not fs: print(L, "no db"); continue
    con = sqlite3.connect(fs[0])
    for k, c, n, a in con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%night%' group by kernel_name, counter_name"):
        print(L, k[:110], c, "n=%d" % n, "FETCH_SIZE x 2 = %.4g GB" % (a * 1024 * 2 / 1e9))
PY
# per-cell series
for k in 1 2 8; do ATLITE_HIP_LIB=$V/lib_fp$k.so timeout 300 python bench.py --legs c3_series --no-cpu-baseline --steps 6 > $OUT/fp$k.json 2> $OUT/fp$k.err; show $OUT/fp$k.json fp$k; done
timeout 300 python bench.py --legs c3_series --no-cpu-baseline --steps 6 > $OUT/fp4.json 2> $OUT/fp4.err; show $OUT/fp4.json fp4
ATLITE_HIP_LIB=$V/lib_fp1.so timeout 300 python bench.py --legs c3_series --no-cpu-baseline --steps 6 > $OUT/fp1b.json 2> $OUT/fp1b.err; show $OUT/fp1b.json fp1b
timeout 300 python bench.py --legs c3_series --no-cpu-baseline --steps 6 > $OUT/fp4b.json 2> $OUT/fp4b.err; show $OUT/fp4b.json fp4b
