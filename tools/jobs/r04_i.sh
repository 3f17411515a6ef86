#!/bin/bash
# round-4 GPU job I: per-cell series in flat order (k_cells_series_flat) against the slot-walking kernel
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_i
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api_golden.py tests/test_gpu_wind_speed.py tests/test_gpu_odd_grids.py tests/test_gpu_fuzz.py -m gpu -x -q > $OUT/gputests.log 2>&1; echo "gputests rc=$?" > $OUT/status
for ch in 0 8 4 16 32 0 8; do
  echo "== ATLITE_HIP_SERIES_CHUNKS=$ch" >> $OUT/ab.log
  ATLITE_HIP_SERIES_CHUNKS=$ch timeout 300 python tools/bench_configs.py C3 2>/dev/null | grep -E "^C[0-9]" >> $OUT/ab.log
  ATLITE_HIP_SERIES_CHUNKS=$ch ATL_VARIANTS="per-cell series out (no matrix), no early-out" timeout 300 python tools/bench_pv_variants.py 2>/dev/null | grep -E "per-cell" >> $OUT/ab.log
done
./tools/probes/mix_probe >> $OUT/mix_probe.log 2>&1
cat $OUT/status; cat $OUT/ab.log
