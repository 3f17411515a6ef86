#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05_k
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
ATLITE_HIP_STREAM_TIMING=1 timeout 400 python tools/bench_ingest.py --T 4380 --quick --keep /tmp/c4380.nc > $OUT/ingest_4380.log 2>&1
grep "wrote\|pv from FILE\|stage split\|identical" $OUT/ingest_4380.log | cut -c1-250
grep "streaming" $OUT/ingest_4380.log | head -24
rm -f /tmp/c4380.nc
