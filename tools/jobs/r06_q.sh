#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_q
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
timeout 1200 python -X faulthandler -m pytest tests/test_gpu_multidevice.py -x -q -m gpu -p no:cacheprovider -k "bench" > $OUT/tests.log 2>&1
echo "rc=$? $(grep -E 'passed|failed' $OUT/tests.log | tail -1)"; grep -E "^(FAILED|ERROR)|Error" $OUT/tests.log | head
