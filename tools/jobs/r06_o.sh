#!/bin/bash
# round 6: the fed launch under its knobs (small batches, several jobs, unfed, a 1 ms time-out) - tests/test_gpu_ingest.py
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_o
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp ATLITE_HIP_INGEST_DEBUG=1
timeout 600 python -X faulthandler -m pytest tests/test_gpu_ingest.py -x -q -m gpu -p no:cacheprovider > $OUT/tests.log 2>&1
echo "rc=$? $(tail -1 $OUT/tests.log)"; grep -E "^(FAILED|ERROR)|Error|ingest\]" $OUT/tests.log | cut -c1-250 | sort | uniq -c | head
