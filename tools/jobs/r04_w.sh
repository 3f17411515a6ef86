#!/bin/bash
# capacity-factor maps on a caller's contiguous cubes of odd grids: chunks by alignment class (class_walk) vs not
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_w
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_odd_grids.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_api_golden.py tests/test_gpu_interleave.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
echo "== classes (default)" > $OUT/map.log
timeout 300 python tools/bench_pitch.py 201 201 189 157 2>/dev/null | grep -E "capacity-factor" >> $OUT/map.log
echo "== ATLITE_HIP_SERIES_NO_SHIFT=1" >> $OUT/map.log
ATLITE_HIP_SERIES_NO_SHIFT=1 timeout 300 python tools/bench_pitch.py 201 201 189 157 2>/dev/null | grep -E "capacity-factor" >> $OUT/map.log
cat $OUT/map.log
