#!/bin/bash
# round-4 GPU job B: GPU tests (device post-processing, crs, async collective, graph), the bench line, the pv family's
# influx heads after the plain Hay-Davies tail, dense-plan baseline, rocprofv3 passes of bench.py's legs.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_b
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
echo "start $(date +%s)" > $OUT/status
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/gputests.log 2>&1; echo "gputests rc=$? $(date +%s)" >> $OUT/status
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? $(date +%s)" >> $OUT/status
ATL_VARIANTS="influx" timeout 600 python tools/bench_pv_variants.py > $OUT/pv_variants.log 2>&1; echo "pv_variants rc=$? $(date +%s)" >> $OUT/status
timeout 600 python tools/bench_dense.py runoff wind pv > $OUT/dense.log 2>&1; echo "dense rc=$? $(date +%s)" >> $OUT/status
timeout 300 python bench.py --emulate-shard 8 --steps 200 --warmup 20 --no-cpu-baseline --no-parity > $OUT/shard8.json 2>> $OUT/bench.err
timeout 1500 python tools/profile_bench.py $OUT/prof headline night_skip star configs c4 > $OUT/profile.log 2>&1; echo "profile rc=$? $(date +%s)" >> $OUT/status
echo "end $(date +%s)" >> $OUT/status
tail -3 $OUT/gputests.log; cat $OUT/status; tail -c 800 $OUT/bench.err
