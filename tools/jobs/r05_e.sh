#!/bin/bash
# round 5, job e - where a stream's cycles go in k_inflate (instrumented variant, printf from two streams per launch)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05_e
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
ATLITE_HIP_LIB=$REPO/atlite_amd/lib/variants/lib_infprof.so timeout 300 python tools/bench_ingest.py --T 1440 --quick --keep /tmp/c1440.nc > $OUT/prof.log 2>&1
grep -v "k_inflate" $OUT/prof.log | cut -c1-220
grep "k_inflate" $OUT/prof.log | tail -14 | cut -c1-300
