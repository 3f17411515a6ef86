#!/bin/bash
# A/B: flat early-out per-cell series (k_cells_series_flat_night) vs k_cells_night; dense plans with / without the MFMA path
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_l
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
for v in 0 1 s 0 1 s; do
  echo "== flat=$v" >> $OUT/ab.log
  if [ "$v" = "s" ]; then export ATLITE_HIP_SERIES_FLAT=1 ATLITE_HIP_SERIES_FLAT_STRIPS=1; else export ATLITE_HIP_SERIES_FLAT=$v; unset ATLITE_HIP_SERIES_FLAT_STRIPS; fi
  ATL_VARIANTS="per-cell series out (no matrix) + night early-out" timeout 300 python tools/bench_pv_variants.py 2>/dev/null | grep -E "per-cell" >> $OUT/ab.log
done
unset ATLITE_HIP_SERIES_FLAT ATLITE_HIP_SERIES_FLAT_STRIPS
cat $OUT/ab.log
timeout 900 python -m pytest tests -m gpu -x -q -k "series or cell or pv or odd or night" 2>&1 | grep -E "passed|failed|rror" | tail -5
echo "== dense, MFMA path" > $OUT/dense.log
ATL_DENSE_R=16,32 timeout 600 python tools/bench_dense.py runoff wind pv 2>/dev/null | grep "dense R" >> $OUT/dense.log
echo "== dense, ATLITE_HIP_NO_MFMA=1 (butterfly)" >> $OUT/dense.log
ATLITE_HIP_NO_MFMA=1 ATL_DENSE_R=16,32 timeout 600 python tools/bench_dense.py runoff wind pv 2>/dev/null | grep "dense R" >> $OUT/dense.log
cat $OUT/dense.log
