#!/bin/bash
# A/B of the flat per-cell series kernel (k_cells_series_flat) against the slot-walking one, alternating, same box
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_j5
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
for v in 0 1 0 1 0 1; do
  echo "== ATLITE_HIP_SERIES_FLAT=$v" >> $OUT/ab.log
  ATLITE_HIP_SERIES_FLAT=$v timeout 300 python tools/bench_configs.py C3 2>/dev/null | grep -E "^C3 " >> $OUT/ab.log
  ATLITE_HIP_SERIES_FLAT=$v ATL_VARIANTS="per-cell series out (no matrix), no early-out" timeout 300 python tools/bench_pv_variants.py 2>/dev/null | grep -E "per-cell" >> $OUT/ab.log
done
cat $OUT/ab.log
timeout 900 python -m pytest tests -m gpu -x -q -k "series or cell or wind or odd or runoff or temperature" 2>&1 | tail -5
