#!/bin/bash
# Round 3, GPU call 1 (kept as the record of how profiles/r03_clock_per_launch.txt, r03_partial_line_probe.txt, r03_occupancy.txt and
# r03_ab_night_rowcache_waves.txt were produced; the variant libraries it A/B-ed were built with tools/build_variant.sh n0 / n1 / w2):
# (gloo, all on GPU 0), launch-to-launch spread with per-dispatch shader clocks, the partial-line probe, the star
# workload's request sizes, and a first A/B of night-kernel variants.   bash tools/r03_job1.sh
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
O=$REPO/gpurun_out/r03_job1
mkdir -p $O
export ATLITE_HIP_DEBUG_OCCUPANCY=1
( timeout 420 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ) 
tail -n 4 $O/pytest.log
unset ATLITE_HIP_DEBUG_OCCUPANCY
# bench.py starts its own ranks: 8 ranks on GPU 0, gloo on host copies (the whole N-rank flow incl. placement check)
timeout 300 python bench.py --gpus 8 --debug-gloo-one-gpu --steps 5 --warmup 2 > $O/gloo8.json 2> $O/gloo8.err
echo "gloo8 rc=$?"; tail -c 600 $O/gloo8.json
# occupancy the runtime reports for the headline / night kernels
ATLITE_HIP_DEBUG_OCCUPANCY=1 timeout 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity 2>&1 >/dev/null | grep "atlite-hip" | sort | uniq -c > $O/occupancy.txt
cat $O/occupancy.txt
# launch-to-launch spread, plain
timeout 200 python tools/launch_spread.py 30 night,base,wind > $O/spread.log 2>&1
grep -E "min .* median" $O/spread.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
grep -iE "RDREQ|FETCH_SIZE|TCC_EA" $O/counters_list.txt | head -40 > $O/counters_tcc.txt
# per-dispatch shader clock
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $O/clk -o clk -- python $REPO/tools/launch_spread.py 10 night,base,wind > $O/clk.log 2>&1
python $REPO/tools/rocpd_clock_per_launch.py $O/clk k_fused > $O/clock_per_launch.txt 2>&1
tail -n 25 $O/clock_per_launch.txt
# partial-line probe: plain, then FETCH_SIZE, then request counters
$REPO/tools/probes/partial_line_probe 8 > $O/probe_plain.txt 2>&1; cat $O/probe_plain.txt
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/probe_fetch -o p -- $REPO/tools/probes/partial_line_probe 8 > $O/probe_fetch.log 2>&1
timeout 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace -d $O/probe_rdreq -o p -- $REPO/tools/probes/partial_line_probe 8 > $O/probe_rdreq.log 2>&1
timeout 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace -d $O/star_rdreq -o p -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-extras --shape-kind star > $O/star_rdreq.log 2>&1
python - <<PY > $O/probe_counters.txt 2>&1
import sqlite3, glob, collections
for d in ("probe_fetch", "probe_rdreq", "star_rdreq"):
    fs = glob.glob("$O/%s/**/*.db" % d, recursive=True)
    if not fs:
        print(d, "no db"); continue
    con = sqlite3.connect(fs[0])
    rows = con.execute("select kernel_name, counter_name, count(*), avg(value) from (select dispatch_id, kernel_name, counter_name, sum(value) as value from counters_collection group by dispatch_id, kernel_name, counter_name) group by kernel_name, counter_name").fetchall()
    for k, c, n, a in rows:
        k = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if k.startswith("k_partial") or k.startswith("k_fused"):
            print(d, k[:70], c, "n=%d" % n, "avg=%.6g" % a)
PY
cat $O/probe_counters.txt
cd $REPO
# A/B: product kernels vs night-kernel variants (row cache 1 / 0, two-wave blocks), interleaved rounds
bash tools/ab_bench.sh 2 --no-extras --night-skip > $O/ab_night.txt 2>&1
bash tools/ab_bench.sh 2 --no-extras > $O/ab_base.txt 2>&1
echo "== night"; cat $O/ab_night.txt; echo "== base"; cat $O/ab_base.txt
rm -rf $O/clk $O/probe_fetch $O/probe_rdreq $O/star_rdreq
