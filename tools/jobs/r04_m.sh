#!/bin/bash
# line-aligned plans: parity tests, then contiguous / aligned / padded on real-world grids
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_m
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_aligned_plans.py tests/test_gpu_odd_grids.py -x -q > $OUT/tests.log 2>&1
grep -E "passed|failed|rror" $OUT/tests.log | tail -5
timeout 900 python tools/bench_pitch.py 201 201 189 157 241 321 201 200 > $OUT/pitch.log 2>&1
cat $OUT/pitch.log
