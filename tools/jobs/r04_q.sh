#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_q
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_aligned_plans.py -x -q -k "dense or aligned" > $OUT/tests.log 2>&1
grep -E "passed|failed|rror" $OUT/tests.log | tail -5
echo "== pair kernel" > $OUT/dense.log
ATL_DENSE_R=16,32,64 timeout 600 python tools/bench_dense.py runoff wind 2>/dev/null | grep "dense R" >> $OUT/dense.log
echo "== ATLITE_HIP_NO_PAIR=1" >> $OUT/dense.log
ATLITE_HIP_NO_PAIR=1 ATL_DENSE_R=16,32,64 timeout 600 python tools/bench_dense.py runoff wind 2>/dev/null | grep "dense R" >> $OUT/dense.log
cat $OUT/dense.log
