#!/bin/bash
# round 6: where a stream's cycles go now (instrumented build, stream 0 and the middle stream print their counters), lone-ish streams: T = 720
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_t
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp ATLITE_HIP_INFLATE=device
ATLITE_HIP_LIB=$REPO/atlite_amd/lib/variants/lib_infprof.so timeout 300 python tools/bench_ingest.py --T 720 --quick --no-host > $OUT/prof.log 2>&1
grep "k_inflate \|decode 0" $OUT/prof.log | tail -6 | cut -c1-400
grep "DEVICE\|launch" $OUT/prof.log | cut -c1-300
