#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05_g
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
ATLITE_HIP_LIB=$REPO/atlite_amd/lib/variants/lib_infprof.so timeout 300 python tools/bench_ingest.py --T 1440 --quick --keep /tmp/c1440.nc > $OUT/prof.log 2>&1
grep "pv from FILE" $OUT/prof.log | cut -c1-200
grep "k_inflate 0/\|decode 0" $OUT/prof.log | tail -8 | cut -c1-300
