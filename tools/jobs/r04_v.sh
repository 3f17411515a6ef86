#!/bin/bash
# per-cell series on a caller's contiguous cubes of odd grids: blocks on each slot's own line grid (shift) vs not
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_v
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_odd_grids.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_api_golden.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
echo "== shift (default)" > $OUT/series.log
timeout 300 python tools/experiments/odd_series.py 2>/dev/null >> $OUT/series.log
echo "== ATLITE_HIP_SERIES_NO_SHIFT=1" >> $OUT/series.log
ATLITE_HIP_SERIES_NO_SHIFT=1 timeout 300 python tools/experiments/odd_series.py 2>/dev/null >> $OUT/series.log
cat $OUT/series.log
