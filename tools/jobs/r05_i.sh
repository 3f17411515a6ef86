#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05_i
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
timeout 300 python tools/bench_ingest.py --T 1440 --quick --keep /tmp/c1440.nc > $OUT/ingest_1440.log 2>&1
grep "pv from FILE\|stage split\|identical" $OUT/ingest_1440.log | cut -c1-250
timeout 400 python tools/bench_ingest.py --T 4380 --quick --keep /tmp/c4380.nc > $OUT/ingest_4380.log 2>&1
grep "wrote\|pv from FILE\|stage split\|identical" $OUT/ingest_4380.log | cut -c1-250
rm -f /tmp/c4380.nc
