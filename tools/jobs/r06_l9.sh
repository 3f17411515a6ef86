#!/bin/bash
# round 6: atlite's own compression level (zlib 9, data.py:139) on its own chunking: T = 2000 of the C2 grid in (100, y, x) chunks
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_l9
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
ATL_FIXTURE_GZIP=9 ATLITE_HIP_INGEST_DEBUG=1 timeout 1500 python tools/bench_ingest.py --T 2000 --quick --chunks 100,200,200 --default-policy > $OUT/l9.log 2>&1
grep "^wrote\|DEVICE\|launch\|host threads\|identical\|rror\|split: streams\|pool" $OUT/l9.log | cut -c1-330 | tail -8
