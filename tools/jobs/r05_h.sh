#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05_h
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
timeout 300 python -X faulthandler -m pytest tests/test_gpu_ingest.py -x -q > $OUT/ingest_tests.log 2>&1
echo "rc=$? $(tail -1 $OUT/ingest_tests.log)"; grep -E "^(FAILED|ERROR)|Error|assert " $OUT/ingest_tests.log | head -10
timeout 300 python tools/bench_ingest.py --T 1440 --quick --keep /tmp/c1440.nc > $OUT/ingest_1440.log 2>&1
grep "pv from FILE\|stage split\|identical" $OUT/ingest_1440.log | cut -c1-250
ATLITE_HIP_LIB=$REPO/atlite_amd/lib/variants/lib_infprof.so timeout 300 python tools/bench_ingest.py --T 1440 --quick --keep /tmp/c1440.nc > $OUT/prof.log 2>&1
grep "k_inflate 0/\|decode 0" $OUT/prof.log | tail -6 | cut -c1-300
