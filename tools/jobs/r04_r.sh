#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_r
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k dense 2>&1 | grep -E "passed|failed" 
for rep in 1 2; do
for c in 16 8 32 16 8; do
  echo "== ATLITE_HIP_CHUNK=$c" >> $OUT/chunk.log
  ATLITE_HIP_CHUNK=$c timeout 300 python bench.py --legs night_skip --no-cpu-baseline --no-parity --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('headline kernel_ms', round(d['roofline']['kernel_ms'],4), 'night', round(d['night_skip'].get('kernel_ms') or d['night_skip']['ms_per_step'],4))" >> $OUT/chunk.log
done
done
cat $OUT/chunk.log
