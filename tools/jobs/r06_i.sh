#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_i
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp ATLITE_HIP_INGEST_TIMEOUT_MS=3000 ATLITE_HIP_INGEST_DEBUG=1
timeout 300 python -X faulthandler -m pytest tests/test_gpu_ingest.py -x -q -m gpu -p no:cacheprovider > $OUT/tests.log 2>&1
echo "rc=$?"; grep -E "ingest\]|passed|failed" $OUT/tests.log | sort | uniq -c | head -30
timeout 600 python -X faulthandler -m pytest tests/test_gpu_streaming.py tests/test_gpu_api_golden.py tests/test_gpu_post.py tests/test_gpu_multidevice.py -x -q -m gpu -p no:cacheprovider > $OUT/tests2.log 2>&1
echo "rc=$?"; grep -E "ingest\]|passed|failed|^FAILED|^ERROR" $OUT/tests2.log | sort | uniq -c | head -30
